"""Phase breakdown of the three scan kernels from the diagnostic build (make -C parallel-gps_amd/csrc stamps [ST=f32 SD=6]).
Run on the GPU box:  PGPS_LIB=parallel-gps_amd/pssgp/libpgps_stamps.so python tools/stamps.py [chunk] [stage]
STAMP_KERNEL / STAMP_DTYPE (bench.py's --kernel / --dtype names; default matern32 / f64) pick the model -- the library must have
been built with the matching ST / SD."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("PGPS_LIB", os.path.join(ROOT, "parallel-gps_amd", "pssgp", "libpgps_stamps.so"))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "parallel-gps_amd"))
import numpy as np
from pssgp import _backend as B
from pssgp.kernels import Matern32

chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 16
stage = int(sys.argv[2]) if len(sys.argv) > 2 else 4
single = int(sys.argv[3]) if len(sys.argv) > 3 else 0      # 1: single-pass filter kernel (k_filter_single)
block = int(sys.argv[4]) if len(sys.argv) > 4 else 0       # lanes per workgroup: 0 auto, 128, 256
dma = int(sys.argv[5]) if len(sys.argv) > 5 else -1        # LDS-DMA ring in the Kalman pass: -1 auto, 0 off, 1 on
sys.path.insert(0, ROOT)
import bench as _bench
kname, dt = os.environ.get("STAMP_KERNEL", "matern32"), os.environ.get("STAMP_DTYPE", "f64")
npdt = np.float64 if dt == "f64" else np.float32
n = 1 << 20
ctx = B.Context(0)
ctx.set_resident(0)
if dt == "f32":
    ctx.set_f32_policy(1)
ctx.set_chunk(chunk); ctx.set_stage(stage); ctx.set_block(block); ctx.set_dma(dma)
if single:
    ctx.set_single_pass(1, 256)
sde = _bench.make_kernel(kname).get_sde()
d = sde.F.shape[0]
rng = np.random.default_rng(0)
ts = np.cumsum(0.05 * rng.uniform(0.5, 1.5, n))
Fs, Qs = B.discretise(sde.F, sde.P0, ts, 0.0)
ys = rng.standard_normal(n)
def dev(a):
    a = np.ascontiguousarray(a, dtype=npdt); p = ctx.malloc(a.nbytes); ctx.h2d(p, a); return p
P0, F_, Q_, H, Y = dev(sde.P0), dev(Fs), dev(Qs), dev(sde.H.reshape(-1)), dev(ys)
w = np.dtype(npdt).itemsize
fms, fPs, sms, sPs, ll = (ctx.malloc(n * d * w), ctx.malloc(n * d * d * w), ctx.malloc(n * d * w),
                          ctx.malloc(n * d * d * w), ctx.malloc(16))
P = ctypes.c_void_p
for it in range(5):
    ctx.call("pgps_pkfs_dev_" + dt, ctypes.c_long(n), ctypes.c_int(d), P(P0), P(F_), P(Q_), P(H), (ctypes.c_double if dt == "f64" else ctypes.c_float)(0.1),
             P(Y), P(fms), P(fPs), P(sms), P(sPs), P(ll))
ctx.synchronize()
lanes, lc, nb = ctx.get_geometry(n, d)
buf = np.zeros((3, nb, 8), dtype=np.int64)
nstamps = {0: 4, 1: 6, 2: 4}
ctx.lib.pgps_debug_read_stamps.argtypes = [P, P, ctypes.c_long]
B.check(ctx, ctx.lib.pgps_debug_read_stamps(ctx.handle, buf.ctypes.data_as(P), buf.size), "read_stamps")
names = {0: ["lane-serial reduce", "block scan", "store lpre/spine"],
         1: ["fold spine (prologue)", "lpre load+apply", "lane-serial KF+smooth-agg", "ll reduce", "suffix scan+store"],
         2: ["fold sspine (prologue)", "lsuf load+apply", "lane-serial RTS"]}
print(f"{lanes} lanes per workgroup, chunk {lc}, {nb} workgroups, stage {stage}; s_memtime ticks are 100 MHz-domain? printing raw ticks and share")
if single:
    names[1] = ["stream to registers + reduce", "block scan", "publish + wait for left totals", "fold + apply",
                "Kalman pass from registers + stores", "ll reduce + suffix scan (+ publish)"]

for k, extra in ((1, 6), (2, 4)):
    arrive = (buf[k][:, extra] - buf[k][:, 0]).astype(np.float64)
    arrive = arrive[buf[k][:, extra] > 0]
    if arrive.size:
        print(f"kernel {k}: spine records in registers {np.median(arrive):.0f} ticks after the workgroup's first stamp (max {arrive.max():.0f})")
    t0 = buf[k][:, 0].astype(np.float64)
    print(f"kernel {k}: workgroup start skew: median {np.median(t0 - t0.min()):.0f} ticks, max {(t0 - t0.min()).max():.0f}")
for k, kn in enumerate(["k_filter_reduce", "k_filter_apply", "k_smoother_apply"]):
    st = buf[k]
    nph = len(names[k])
    dur = np.diff(st[:, :nph + 1], axis=1).astype(np.float64)
    tot = dur.sum(axis=1)
    span = st[:, nph].max() - st[:, 0].min()
    if single and k == 0:
        continue
    print(f"{kn}: median workgroup total {np.median(tot):.0f} ticks; first-start to last-end {span} ticks")
    for i, nm in enumerate(names[k]):
        print(f"    {nm:32s} median {np.median(dur[:, i]):9.0f}  max {dur[:, i].max():9.0f}  share {np.median(dur[:, i]) / np.median(tot):5.1%}")
