import sys, os, time, ctypes, gc
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/parallel-gps_amd")
import numpy as np, torch
from pssgp import _backend
from pssgp.kernels import Matern32
dev = torch.device("cuda", 0)
ctx = _backend.Context(0); ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
n = 1 << 20; d = 2
sde = Matern32(1., 1.).get_sde()
rng = np.random.default_rng(0)
ts = np.cumsum(0.05 * rng.uniform(0.5, 1.5, n))
T = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
F_d, P0_d, H_d, ts_d = T(sde.F), T(sde.P0), T(sde.H.reshape(-1)), T(ts)
Fs = torch.empty((n, d, d), dtype=torch.float64, device=dev); Qs = torch.empty_like(Fs)
P = lambda t: ctypes.c_void_p(t.data_ptr())
ctx.call("pgps_discretise_dev_f64", ctypes.c_long(n), ctypes.c_int(d), P(F_d), P(P0_d), P(ts_d), ctypes.c_double(0.0), P(Fs), P(Qs))
ys = T(rng.standard_normal(n))
fms = torch.empty((n, d), dtype=torch.float64, device=dev); sms = torch.empty_like(fms)
fPs = torch.empty_like(Fs); sPs = torch.empty_like(Fs); ll = torch.zeros(2, dtype=torch.float64, device=dev)
def step():
    ctx.call("pgps_pkfs_dev_f64", ctypes.c_long(n), ctypes.c_int(d), P(P0_d), P(Fs), P(Qs), P(H_d), ctypes.c_double(0.1), P(ys), P(fms), P(fPs), P(sms), P(sPs), P(ll))
for cfg in [(16, 4, 0), (16, 4, 8), (8, 0, 0), (8, 0, 8)]:
    ctx.set_chunk(cfg[0]); ctx.set_stage(cfg[1]); ctx.profile_enable(cfg[2])
    for _ in range(20): step()
    torch.cuda.synchronize()
    for rep in range(3):
        hs = []
        t0 = time.perf_counter()
        for _ in range(200):
            a = time.perf_counter(); step(); hs.append(time.perf_counter() - a)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        hs = np.array(hs) * 1e6
        print(cfg, "rep", rep, "enqueue total %.1f ms, +sync %.1f ms => %.1f us/step | host per-step median %.1f us, max %.1f us, >1ms: %d" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t2 - t0) / 200 * 1e6, np.median(hs), hs.max(), (hs > 1000).sum()))
    ctx.profile_read(True)
