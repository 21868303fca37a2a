"""The multi-GPU (segment-sharded) path.

CPU (`not gpu`):  the protocol stitched from numpy stand-ins reproduces the unsegmented oracle;
                  the same `run_protocol` driver under torch.distributed / gloo with world_size 2.
GPU (`gpu`):      R logical ranks on ONE MI355X -- one libpgps context (own scratch) per rank,
                  the all-gather done by hand -- against the oracle, records checked field by field.
"""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import np_oracle as O
from oracle.segments import OracleSegment, unpack_filter_record, unpack_smoother_record
from tests.conftest import make_times, relerr, sample_series

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _problem(n=700, seed=3, kernel=None):
    from pssgp.kernels import Matern52
    k = kernel or Matern52(1., 0.8)
    t = make_times(n, seed=seed)
    ssm = O.get_ssm(k.get_sde(), t, 0.1)
    y = sample_series(ssm, seed=seed, nan_frac=0.15)
    return ssm, y


def _slice(ssm, lo, hi):
    P0, Fs, Qs, H, R = ssm
    return (P0, Fs[lo:hi], Qs[lo:hi], H, R)


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_protocol_with_cpu_standins_equals_unsegmented(world):
    from pssgp.distributed import run_protocol, split_segments
    ssm, y = _problem()
    fms, fPs, ll = O.pkf(ssm, y, True)
    sms, sPs = O.pks(ssm, fms, fPs)
    bounds = split_segments(y.size, world)
    assert bounds[0][0] == 0 and bounds[-1][1] == y.size and all(b[1] - b[0] >= y.size // world for b in bounds)
    segs = [OracleSegment(r, world, _slice(ssm, lo, hi), y[lo:hi]) for r, (lo, hi) in enumerate(bounds)]
    # lock-step emulation of the collectives
    gf = np.stack([s.phase_reduce() for s in segs])
    gs = np.stack([s.phase_filter(gf) for s in segs])
    for s in segs:
        s.phase_smoother(gs)
    assert relerr(np.concatenate([s.fms for s in segs]), fms) < 1e-11
    assert relerr(np.concatenate([s.fPs for s in segs]), fPs) < 1e-11
    assert relerr(np.concatenate([s.sms for s in segs]), sms) < 1e-11
    assert relerr(np.concatenate([s.sPs for s in segs]), sPs) < 1e-11
    assert all(abs(s.ll - ll) < 1e-11 * abs(ll) for s in segs)
    # and through the shared driver, one rank at a time with a canned gather
    for r, s in enumerate(segs):
        seq = iter((gf, gs))
        out = run_protocol(r, world, s.phase_reduce, s.phase_filter, s.phase_smoother, lambda rec: next(seq))
        assert abs(out - ll) < 1e-11 * abs(ll)


_GLOO_WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "parallel-gps_amd"))
from oracle import np_oracle as O
from oracle.segments import OracleSegment
from pssgp.distributed import run_protocol, split_segments
from tests.test_segments import _problem, _slice

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
ssm, y = _problem()
lo, hi = split_segments(y.size, world)[rank]
seg = OracleSegment(rank, world, _slice(ssm, lo, hi), y[lo:hi])

def all_gather(rec):
    rec = torch.from_numpy(np.ascontiguousarray(rec))
    out = [torch.empty_like(rec) for _ in range(world)]
    dist.all_gather(out, rec)
    return torch.stack(out).numpy()

ll = run_protocol(rank, world, seg.phase_reduce, seg.phase_filter, seg.phase_smoother, all_gather)
fms, fPs, ll_ref = O.pkf(ssm, y, True)
sms, sPs = O.pks(ssm, fms, fPs)
err = max(np.max(np.abs(seg.fms - fms[lo:hi])), np.max(np.abs(seg.sms - sms[lo:hi])),
          np.max(np.abs(seg.sPs - sPs[lo:hi])), abs(ll - ll_ref) / abs(ll_ref))
dist.barrier()
dist.destroy_process_group()
print("RANK", rank, "ERR", err)
sys.exit(0 if err < 1e-10 else 3)
'''


def test_protocol_under_gloo_world_size_2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_GLOO_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29617", WORLD_SIZE="2",
               PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{out}"
        assert "ERR" in out


def test_record_lengths_match_layout():
    from pssgp.distributed import record_lengths
    for d in range(1, 7):
        rf, rs, pad = record_lengths(d)
        sym = d * (d + 1) // 2
        assert rf == (d * d + 2 * d + 2 * sym) + 2 * d * d
        assert pad % 2 == 0 and pad >= d * d + d + sym and rs == pad + 2


# ------------------------------------------------------------------------------------------------
# GPU: several logical ranks on one device
# ------------------------------------------------------------------------------------------------
class _Dev:
    """Tiny device-array helper over pgps_malloc / memcpy (no torch in the GPU parity tests)."""

    def __init__(self, ctx, arr=None, shape=None, dtype=np.float64):
        self.ctx = ctx
        if arr is not None:
            arr = np.ascontiguousarray(arr, dtype=dtype)
            shape = arr.shape
        self.shape, self.dtype = tuple(shape), np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        self.ptr = ctx.malloc(max(self.nbytes, 16))
        if arr is not None:
            ctx.h2d(self.ptr, arr)

    def get(self):
        out = np.empty(self.shape, self.dtype)
        self.ctx.d2h(out, self.ptr)
        return out

    def put(self, arr):
        self.ctx.h2d(self.ptr, np.ascontiguousarray(arr, dtype=self.dtype).reshape(self.shape))

    @property
    def p(self):
        return ctypes.c_void_p(self.ptr)

    def free(self):
        self.ctx.free(self.ptr)


def _seg_kernel(name):
    from pssgp.kernels import Matern32, Matern52, RBF, Periodic, SquaredExponential
    return {
        "m32+m52": lambda: Matern32(1., 1.) + Matern52(1., 0.7),                                     # d = 5: lane-chunk
        "rbf8": lambda: RBF(variance=1., lengthscales=0.7, order=8, balancing_iter=10),              # d = 8: row-cooperative
        "c5": lambda: Periodic(SquaredExponential(1., 1.), period=1., order=1) * Matern32(1., 1.) + Matern52(1., 1.),
        # above d = 16: the wave-cooperative family (the reference's CO2 kernel, d = 18; Periodic order 10, d = 22)
        "co2": lambda: Periodic(SquaredExponential(1., 1.), period=1., order=3) * Matern32(0.5, 5.) + Matern32(1., 2.),
        "periodic10": lambda: Periodic(SquaredExponential(1., 0.8), period=1.5, order=10),
    }[name]()


def run_segments_on_one_gpu(ssm, y, bounds, dtype, family):
    """R logical ranks on one device, one context each, the all-gathers by hand.  Returns the concatenated outputs, every
    rank's log-likelihood and the gathered records."""
    from pssgp import _backend as B
    from pssgp.distributed import record_lengths
    d = ssm[1].shape[1]
    world = len(bounds)
    suf, real = B._suffix(dtype)
    rf, rs, pad = record_lengths(d)
    ranks = []
    for r, (lo, hi) in enumerate(bounds):
        ctx = B.Context(0)
        ctx.set_family(family)
        n = hi - lo
        P0, Fs, Qs, H, R = _slice(ssm, lo, hi)
        ranks.append(dict(ctx=ctx, n=n, lo=lo, hi=hi,
                          P0=_Dev(ctx, P0, dtype=dtype), Fs=_Dev(ctx, Fs, dtype=dtype), Qs=_Dev(ctx, Qs, dtype=dtype),
                          H=_Dev(ctx, H.reshape(-1), dtype=dtype), ys=_Dev(ctx, y[lo:hi], dtype=dtype),
                          fms=_Dev(ctx, shape=(n, d), dtype=dtype), fPs=_Dev(ctx, shape=(n, d, d), dtype=dtype),
                          sms=_Dev(ctx, shape=(n, d), dtype=dtype), sPs=_Dev(ctx, shape=(n, d, d), dtype=dtype),
                          rec_f=_Dev(ctx, shape=(rf,), dtype=dtype), rec_s=_Dev(ctx, shape=(rs,), dtype=dtype),
                          gf=_Dev(ctx, shape=(world, rf), dtype=dtype), gs=_Dev(ctx, shape=(world, rs), dtype=dtype),
                          ll=_Dev(ctx, shape=(2,), dtype=np.float64)))
    Rv = real(float(ssm[4].reshape(())))
    L, I = ctypes.c_long, ctypes.c_int
    try:
        for r, k in enumerate(ranks):
            k["ctx"].call(f"pgps_seg_filter_reduce_dev_{suf}", L(k["n"]), I(d), I(r), I(world), k["P0"].p, k["Fs"].p,
                          k["Qs"].p, k["H"].p, Rv, k["ys"].p, k["rec_f"].p)
        gf = np.stack([k["rec_f"].get() for k in ranks])
        for r, k in enumerate(ranks):
            k["gf"].put(gf)
            k["ctx"].call(f"pgps_seg_filter_apply_dev_{suf}", L(k["n"]), I(d), I(r), I(world), k["P0"].p, k["Fs"].p,
                          k["Qs"].p, k["H"].p, Rv, k["ys"].p, k["gf"].p, k["fms"].p, k["fPs"].p, k["rec_s"].p)
        gs = np.stack([k["rec_s"].get() for k in ranks])
        for r, k in enumerate(ranks):
            k["gs"].put(gs)
            k["ctx"].call(f"pgps_seg_smoother_apply_dev_{suf}", L(k["n"]), I(d), I(r), I(world), k["Fs"].p, k["Qs"].p,
                          k["fms"].p, k["fPs"].p, k["gs"].p, k["sms"].p, k["sPs"].p, k["ll"].p)
        got = {n: np.concatenate([k[n].get() for k in ranks]) for n in ("fms", "fPs", "sms", "sPs")}
        lls = [float(k["ll"].get()[0]) for k in ranks]
    finally:
        for k in ranks:
            k["ctx"].synchronize()
            for v in k.values():
                if isinstance(v, _Dev):
                    v.free()
            k["ctx"].close()
    return got, lls, gf, gs


@pytest.mark.gpu
@pytest.mark.parametrize("world,dtype,kname,family", [
    (1, np.float64, "m32+m52", 0), (2, np.float64, "m32+m52", 0), (3, np.float64, "m32+m52", 0), (8, np.float64, "m32+m52", 0),
    (4, np.float32, "m32+m52", 0),
    # the row-cooperative family's segment protocol: forced at d = 5, automatic above d = 6
    (3, np.float64, "m32+m52", 3), (1, np.float64, "rbf8", 0), (2, np.float64, "rbf8", 0), (8, np.float64, "rbf8", 0),
    (3, np.float64, "c5", 0), (5, np.float64, "c5", 0),
    # the wave-cooperative family's: automatic above d = 16, forced (family 2) below
    (1, np.float64, "co2", 0), (2, np.float64, "co2", 0), (5, np.float64, "co2", 0), (3, np.float64, "periodic10", 0),
    (3, np.float32, "co2", 0), (3, np.float64, "rbf8", 2), (4, np.float64, "m32+m52", 2)])
def test_segments_on_one_gpu(world, dtype, kname, family):
    from pssgp.distributed import split_segments
    ssm, y = _problem(n=5000, seed=9, kernel=_seg_kernel(kname))
    d = ssm[1].shape[1]
    fms, fPs, ll = O.kf(ssm, y, True)
    sms, sPs = O.kfs(ssm, y)
    bounds = split_segments(y.size, world)
    got, lls, gf, gs = run_segments_on_one_gpu(ssm, y, bounds, dtype, family)
    # the records themselves: compare with the numpy stand-ins field by field
    segs = [OracleSegment(r, world, _slice(ssm, lo, hi), y[lo:hi]) for r, (lo, hi) in enumerate(bounds)]
    gf_o = np.stack([s.phase_reduce() for s in segs])
    tol = (1e-9 if kname == "m32+m52" else 1e-7) if dtype == np.float64 else 2e-3
    for r in range(world):
        (A, b, C, J, eta), F0, Q0 = unpack_filter_record(gf[r].astype(np.float64), d)
        (Ao, bo, Co, Jo, etao), F0o, Q0o = unpack_filter_record(gf_o[r], d)
        assert relerr(b, bo) < tol and relerr(C, Co) < tol and relerr(F0, F0o) < tol and relerr(Q0, Q0o) < tol
        if r > 0:       # J, eta of a prefix that holds the first element never reach an output
            # A of a long segment underflows towards 0 (the filter forgets): absolute scale 1
            assert np.max(np.abs(A - Ao)) < tol * max(1.0, np.max(np.abs(Ao)))
            assert relerr(J, Jo) < tol and relerr(eta, etao) < tol
    assert relerr(got["fms"], fms) < tol and relerr(got["fPs"], fPs) < tol
    assert relerr(got["sms"], sms) < tol and relerr(got["sPs"], sPs) < tol
    for v in lls:
        assert abs(v - ll) < (1e-10 if dtype == np.float64 else 1e-4) * abs(ll)


@pytest.mark.gpu
def test_segmentscan_world1_with_torch():
    """The torch.distributed driver at world_size 1 (what bench.py runs per rank), on the GPU."""
    torch = pytest.importorskip("torch")
    from pssgp import _backend as B
    from tools.torch_segment_scan import SegmentScan
    ssm, y = _problem(n=4000, seed=2)
    d = ssm[1].shape[1]
    dev = torch.device("cuda", 0)
    ctx = B.Context(0)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
    P0, Fs, Qs, H, ys = T(ssm[0]), T(ssm[1]), T(ssm[2]), T(ssm[3].reshape(-1)), T(y)
    n = y.size
    fms, sms = torch.empty((n, d), dtype=torch.float64, device=dev), torch.empty((n, d), dtype=torch.float64, device=dev)
    fPs, sPs = torch.empty((n, d, d), dtype=torch.float64, device=dev), torch.empty((n, d, d), dtype=torch.float64, device=dev)
    ll = torch.zeros(2, dtype=torch.float64, device=dev)
    seg = SegmentScan(ctx, 0, 1, d, np.float64, torch_device=dev)
    seg.pkfs(n, P0, Fs, Qs, H, 0.1, ys, fms, fPs, sms, sPs, ll)
    torch.cuda.synchronize(dev)
    of, oP, oll = O.kf(ssm, y, True)
    os_, osP = O.kfs(ssm, y)
    assert relerr(fms.cpu().numpy(), of) < 1e-9 and relerr(sms.cpu().numpy(), os_) < 1e-9
    assert relerr(sPs.cpu().numpy(), osP) < 1e-9 and abs(ll[0].item() - oll) < 1e-10 * abs(oll)
    ctx.close()


_GPU_WORKER = r"""
import os, sys
import numpy as np
import torch                                   # before libpgps: one HIP runtime per process
import torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "parallel-gps_amd"))
from oracle import np_oracle as O
from pssgp import _backend as B
from pssgp.distributed import split_segments
from tools.torch_segment_scan import SegmentScan
from tests.test_segments import _problem, _slice

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
dev = torch.device("cuda", 0)                  # every rank on GPU 0: RCCL would refuse, gloo does not care
torch.cuda.set_device(dev)
ssm, y = _problem(n=40000, seed=4)
d = ssm[1].shape[1]
lo, hi = split_segments(y.size, world)[rank]
P0, Fs, Qs, H, R = _slice(ssm, lo, hi)
ctx = B.Context(0)
ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
tP0, tFs, tQs, tH, tys = T(P0), T(Fs), T(Qs), T(H.reshape(-1)), T(y[lo:hi])
n = hi - lo
E = lambda *s: torch.empty(s, dtype=torch.float64, device=dev)
fms, fPs, sms, sPs, ll = E(n, d), E(n, d, d), E(n, d), E(n, d, d), torch.zeros(2, dtype=torch.float64, device=dev)
seg = SegmentScan(ctx, rank, world, d, np.float64, torch_device=dev)
for _ in range(3):                             # repeated passes reuse the scratch and the records
    seg.pkfs(n, tP0, tFs, tQs, tH, 0.1, tys, fms, fPs, sms, sPs, ll)
torch.cuda.synchronize(dev)
of, oP, oll = O.kf(ssm, y, True)
os_, osP = O.kfs(ssm, y)
rel = lambda a, b: float(np.max(np.abs(a - b)) / np.max(np.abs(b)))
err = max(rel(fms.cpu().numpy(), of[lo:hi]), rel(fPs.cpu().numpy(), oP[lo:hi]), rel(sms.cpu().numpy(), os_[lo:hi]),
          rel(sPs.cpu().numpy(), osP[lo:hi]), abs(ll[0].item() - oll) / abs(oll))
dist.barrier()
dist.destroy_process_group()
print("RANK", rank, "ERR", err)
sys.exit(0 if err < 1e-9 else 3)
"""


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_segmentscan_multiprocess_on_one_gpu(tmp_path, world):
    """The real multi-process driver -- one process, one libpgps context and one torch.distributed rank
    per segment -- with all ranks on GPU 0 and gloo standing in for RCCL (which refuses two ranks on
    one device): everything except the collective's transport is what `bench.py --gpus N` runs."""
    script = tmp_path / "worker.py"
    script.write_text(_GPU_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29700 + world), WORLD_SIZE=str(world),
               PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{out[-3000:]}"


@pytest.mark.gpu
def test_config_c4_scale_eight_segments_on_one_gpu():
    """BASELINE config c4 (Matern-3/2, fp64, 2^24 steps as 8 contiguous segments of 2^21) with the eight ranks played
    one after the other on a single GPU, the two all-gathers done by numpy: every rank's slice of the filtered and
    smoothed moments and the log-likelihood against the sequential C oracle on the whole series."""
    from oracle import c_oracle as C
    from pssgp import _backend as B
    from pssgp.distributed import record_lengths, split_segments
    from pssgp.kernels import Matern32
    from tests.conftest import make_times, sample_series_fast
    world, n_total, d = 8, 1 << 24, 2
    sde = Matern32(1., 1.).get_sde()
    t = make_times(n_total, seed=0)
    Fs, Qs = B.discretise(sde.F, sde.P0, t, 0.0)
    ssm = (np.asarray(sde.P0), Fs, Qs, np.asarray(sde.H).reshape(1, -1), np.array([[0.1]]))
    y = sample_series_fast(ssm, seed=0, nan_frac=0.1)
    cf, cP, cs, csP, cll = C.kfs(ssm, y)
    rf, rs, _ = record_lengths(d)
    bounds = split_segments(n_total, world)
    L, I, Rv = ctypes.c_long, ctypes.c_int, ctypes.c_double(0.1)
    ranks = []
    try:
        for r, (lo, hi) in enumerate(bounds):
            n = hi - lo
            ctx = B.Context(0)              # a rank = a context: the scan workspace lives in it between the phases
            ranks.append(dict(ctx=ctx, n=n, lo=lo, hi=hi, P0=_Dev(ctx, ssm[0]), H=_Dev(ctx, ssm[3].reshape(-1)),
                              Fs=_Dev(ctx, Fs[lo:hi]), Qs=_Dev(ctx, Qs[lo:hi]), ys=_Dev(ctx, y[lo:hi]),
                              fms=_Dev(ctx, shape=(n, d)), fPs=_Dev(ctx, shape=(n, d, d)), sms=_Dev(ctx, shape=(n, d)),
                              sPs=_Dev(ctx, shape=(n, d, d)), rec_f=_Dev(ctx, shape=(rf,)), rec_s=_Dev(ctx, shape=(rs,)),
                              gf=_Dev(ctx, shape=(world, rf)), gs=_Dev(ctx, shape=(world, rs)), ll=_Dev(ctx, shape=(2,))))
        for r, k in enumerate(ranks):
            k["ctx"].call("pgps_seg_filter_reduce_dev_f64", L(k["n"]), I(d), I(r), I(world), k["P0"].p, k["Fs"].p, k["Qs"].p, k["H"].p, Rv,
                     k["ys"].p, k["rec_f"].p)
        gf = np.stack([k["rec_f"].get() for k in ranks])
        for r, k in enumerate(ranks):
            k["gf"].put(gf)
            k["ctx"].call("pgps_seg_filter_apply_dev_f64", L(k["n"]), I(d), I(r), I(world), k["P0"].p, k["Fs"].p, k["Qs"].p, k["H"].p, Rv,
                     k["ys"].p, k["gf"].p, k["fms"].p, k["fPs"].p, k["rec_s"].p)
        gs = np.stack([k["rec_s"].get() for k in ranks])
        for r, k in enumerate(ranks):
            k["gs"].put(gs)
            k["ctx"].call("pgps_seg_smoother_apply_dev_f64", L(k["n"]), I(d), I(r), I(world), k["Fs"].p, k["Qs"].p, k["fms"].p,
                     k["fPs"].p, k["gs"].p, k["sms"].p, k["sPs"].p, k["ll"].p)
        for k in ranks:
            lo, hi = k["lo"], k["hi"]
            assert relerr(k["fms"].get(), cf[lo:hi]) < 1e-8 and relerr(k["fPs"].get(), cP[lo:hi]) < 1e-8
            assert relerr(k["sms"].get(), cs[lo:hi]) < 1e-8 and relerr(k["sPs"].get(), csP[lo:hi]) < 1e-8
            assert abs(k["ll"].get()[0] - cll) < 1e-10 * abs(cll)
    finally:
        for k in ranks:
            k["ctx"].synchronize()
            for v in k.values():
                if isinstance(v, _Dev):
                    v.free()
            k["ctx"].close()


# ------------------------------------------------------------------------------------------------
# GPU: the exchange inside libpgps (the context owns the RCCL communicator)
# ------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("dtype,kname", [(np.float64, None), (np.float64, "m32+m52"), (np.float32, "m32+m52"),
                                         (np.float64, "rbf8"), (np.float32, "rbf8"), (np.float64, "c5"),
                                         (np.float64, "co2")])
def test_rccl_communicator_in_library_world_1(dtype, kname, tmp_path):
    """pgps_comm_init + pgps_pkfs_seg_dev_* with a REAL RCCL communicator of one rank (all this box has): the five
    launches and the two ncclAllGathers of a pass go out on the context's stream with no host step in between, no
    torch anywhere.  Against the oracle; twice, to see that the scratch and records are reusable."""
    from pssgp import _backend as B
    from pssgp.distributed import ShardedScan, share_unique_id
    ssm, y = _problem(n=6000, seed=11, kernel=_seg_kernel(kname) if kname else None)
    d = ssm[1].shape[1]
    of, oP, oll = O.kf(ssm, y, True)
    os_, osP = O.kfs(ssm, y)
    ctx = B.Context(0)
    assert ctx.comm_info() == (0, 0)
    uid = share_unique_id(0, path=str(tmp_path / "uid"), run_id="world-1-test")
    assert share_unique_id(1, path=str(tmp_path / "uid"), run_id="world-1-test") == uid        # what another rank would read
    seg = ShardedScan(ctx, uid, 0, 1, d, dtype)
    assert ctx.comm_info() == (0, 1)
    n = y.size
    P0, Fs, Qs, H = (_Dev(ctx, a, dtype=dtype) for a in (ssm[0], ssm[1], ssm[2], ssm[3].reshape(-1)))
    ys = _Dev(ctx, y, dtype=dtype)
    fms, fPs, sms, sPs = (_Dev(ctx, shape=s, dtype=dtype) for s in ((n, d), (n, d, d), (n, d), (n, d, d)))
    ll = _Dev(ctx, shape=(2,), dtype=np.float64)
    for _ in range(2):
        seg.pkfs(n, P0.ptr, Fs.ptr, Qs.ptr, H.ptr, float(ssm[4].reshape(())), ys.ptr, fms.ptr, fPs.ptr, sms.ptr, sPs.ptr,
                 ll.ptr)
    ctx.synchronize()
    tol = (1e-9 if d <= 5 else 1e-7) if dtype == np.float64 else 2e-3
    assert relerr(fms.get(), of) < tol and relerr(fPs.get(), oP) < tol
    assert relerr(sms.get(), os_) < tol and relerr(sPs.get(), osP) < tol
    assert abs(ll.get()[0] - oll) < (1e-10 if dtype == np.float64 else 1e-4) * abs(oll)
    # a second communicator on the same context is refused; after destroy the pass is refused
    with pytest.raises(B.PgpsError):
        ctx.comm_init(uid, 0, 1)
    seg.close()
    with pytest.raises(B.PgpsError):
        seg.pkfs(n, P0.ptr, Fs.ptr, Qs.ptr, H.ptr, 0.1, ys.ptr, fms.ptr, fPs.ptr, sms.ptr, sPs.ptr, ll.ptr)
    for v in (P0, Fs, Qs, H, ys, fms, fPs, sms, sPs, ll):
        v.free()
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kname", [None, "rbf8", "co2"])
def test_segment_phases_out_of_order_are_refused(kname):
    """The three phases keep state in the context's scratch (for d > 6 the smoothing elements themselves): a phase
    that does not directly follow its predecessor on the context gets PGPS_E_INVALID, not stale scratch."""
    from pssgp import _backend as B
    from pssgp.distributed import record_lengths
    ssm, y = _problem(n=3000, seed=5, kernel=_seg_kernel(kname) if kname else None)
    d = ssm[1].shape[1]
    n = y.size
    rf, rs, _ = record_lengths(d)
    ctx = B.Context(0)
    P0, Fs, Qs, H = (_Dev(ctx, a) for a in (ssm[0], ssm[1], ssm[2], ssm[3].reshape(-1)))
    ys = _Dev(ctx, y)
    fms, fPs, sms, sPs = (_Dev(ctx, shape=s) for s in ((n, d), (n, d, d), (n, d), (n, d, d)))
    rec_f, rec_s, ll = _Dev(ctx, shape=(rf,)), _Dev(ctx, shape=(rs,)), _Dev(ctx, shape=(2,))
    L, I, Rv = ctypes.c_long, ctypes.c_int, ctypes.c_double(0.1)

    def reduce_():
        ctx.call("pgps_seg_filter_reduce_dev_f64", L(n), I(d), I(0), I(1), P0.p, Fs.p, Qs.p, H.p, Rv, ys.p, rec_f.p)

    def filter_(nn=n):
        ctx.call("pgps_seg_filter_apply_dev_f64", L(nn), I(d), I(0), I(1), P0.p, Fs.p, Qs.p, H.p, Rv, ys.p, rec_f.p, fms.p,
                 fPs.p, rec_s.p)

    def smoother_():
        ctx.call("pgps_seg_smoother_apply_dev_f64", L(n), I(d), I(0), I(1), Fs.p, Qs.p, fms.p, fPs.p, rec_s.p, sms.p, sPs.p,
                 ll.p)

    def refused(fn, *a):
        with pytest.raises(B.PgpsError) as e:
            fn(*a)
        assert e.value.code == -1

    refused(filter_)                    # no phase 1 at all
    refused(smoother_)
    reduce_()
    refused(smoother_)                  # phase 2 skipped
    reduce_()
    refused(filter_, n - 1)             # another N
    reduce_()
    ctx.call("pgps_pkf_dev_f64", L(n), I(d), P0.p, Fs.p, Qs.p, H.p, Rv, ys.p, fms.p, fPs.p, ll.p)   # scratch re-laid out
    refused(filter_)
    reduce_(); filter_()
    ctx.set_chunk(8)
    refused(smoother_)                  # geometry changed between the phases
    ctx.set_chunk(0)
    reduce_(); filter_(); smoother_()   # and the regular order still works
    ctx.synchronize()
    os_, osP = O.kfs(ssm, y)
    assert relerr(sms.get(), os_) < 1e-7 and relerr(sPs.get(), osP) < 1e-7
    for v in (P0, Fs, Qs, H, ys, fms, fPs, sms, sPs, rec_f, rec_s, ll):
        v.free()
    ctx.close()


# ------------------------------------------------------------------------------------------------
# CPU: the communicator id reaches every rank of THIS launch only (no clocks, no file ages)
# ------------------------------------------------------------------------------------------------
def _fake_id(byte):
    from pssgp import _backend as B
    return lambda: bytes([byte]) * B.COMM_ID_BYTES


def test_unique_id_by_run_tag_ignores_another_runs_file(tmp_path):
    from pssgp.distributed import share_unique_id
    path = str(tmp_path / "uid")
    old = share_unique_id(0, path=path, run_id="run-A", make_id=_fake_id(1))
    with pytest.raises(TimeoutError):                     # run B's rank 1 does not join run A's id, however fresh the file is
        share_unique_id(1, path=path, run_id="run-B", timeout=0.3)
    new = share_unique_id(0, path=path, run_id="run-B", make_id=_fake_id(2))
    assert share_unique_id(1, path=path, run_id="run-B", timeout=5.0) == new != old


def test_unique_id_without_a_launch_id_needs_the_handshake(tmp_path, monkeypatch):
    from pssgp.distributed import share_unique_id
    for k in ("TORCHELASTIC_RUN_ID", "SLURM_JOB_ID"):
        monkeypatch.delenv(k, raising=False)
    with pytest.raises(ValueError):
        share_unique_id(1, path=str(tmp_path / "uid"), timeout=0.2)
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "none")     # plain torchrun
    with pytest.raises(ValueError):
        share_unique_id(1, path=str(tmp_path / "uid"), timeout=0.2)


@pytest.mark.parametrize("rank0_late", [False, True])
def test_unique_id_handshake_survives_a_dead_runs_files(tmp_path, monkeypatch, rank0_late):
    """A relaunch at the same path right after a crash (plain torchrun: no launch id): the dead run's id file -- seconds old --
    and its hello / ack files are all there.  Every rank must end up with the NEW id, whichever side starts first."""
    import threading
    import time as _time
    from pssgp import _backend as B
    from pssgp.distributed import share_unique_id, remove_unique_id_file
    for k in ("TORCHELASTIC_RUN_ID", "SLURM_JOB_ID"):
        monkeypatch.delenv(k, raising=False)
    path = str(tmp_path / "uid")
    world = 4
    # what the dead run left: a well-formed id file with ITS tokens, and the matching hello / ack files
    dead_tokens = [bytes([0x40 + r]) * 16 for r in range(1, world)]
    with open(path, "wb") as fh:
        fh.write(b"PGPSUID2HANDSHAK" + b"\x07" * B.COMM_ID_BYTES + b"".join(dead_tokens))
    for r in range(1, world):
        for kind in ("hello", "ack"):
            with open(f"{path}.{kind}.{r}", "wb") as fh:
                fh.write(dead_tokens[r - 1])
    got, errs = {}, []

    def run(rank, delay):
        try:
            _time.sleep(delay)
            got[rank] = share_unique_id(rank, path=path, world=world, timeout=20.0, make_id=_fake_id(9))
        except Exception as e:              # noqa: BLE001
            errs.append((rank, repr(e)))

    ths = [threading.Thread(target=run, args=(r, (0.3 if rank0_late else 0.0) if r == 0 else (0.0 if rank0_late else 0.3) + 0.05 * r))
           for r in range(world)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(30.0)
    assert not errs, errs
    assert len(got) == world and all(v == bytes([9]) * B.COMM_ID_BYTES for v in got.values())
    remove_unique_id_file(path, world)
    assert not list(tmp_path.iterdir())
