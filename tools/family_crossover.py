"""Times pkfs (fp64, 2^18 steps, device-resident arrays) with the lane-chunk and the row-cooperative kernels at state
dimensions 3..6 -- the measurement behind the automatic switch at d = 5 (csrc/pgps_core.hip dispatch_scan)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "parallel-gps_amd"))
from pssgp import _backend as B  # noqa: E402
from pssgp.kernels import Matern32, Matern52, RBF  # noqa: E402

dev = torch.device("cuda:0")
ctx = B.get_context(0)
ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())
n = 1 << 18
rng = np.random.default_rng(0)
t = np.cumsum(0.05 * rng.uniform(0.5, 1.5, n))
kernels = {3: Matern52(1., 1.), 4: Matern32(1., 1.) * Matern32(1., 0.7), 5: Matern32(1., 1.) + Matern52(1., 0.7),
           6: RBF(1., 1., order=6, balancing_iter=10)}
for d, k in kernels.items():
    sde = k.get_sde()
    Fs, Qs = B.discretise(sde.F, sde.P0, t, 0.0)
    T = lambda a: torch.tensor(np.ascontiguousarray(a, np.float64), device=dev)
    P0, Fd, Qd, H, ys = T(sde.P0), T(Fs), T(Qs), T(np.asarray(sde.H).reshape(-1)), T(np.sin(t))
    outs = [torch.empty((n, d), dtype=torch.float64, device=dev), torch.empty((n, d, d), dtype=torch.float64, device=dev),
            torch.empty((n, d), dtype=torch.float64, device=dev), torch.empty((n, d, d), dtype=torch.float64, device=dev)]
    ll = torch.zeros(2, dtype=torch.float64, device=dev)
    res = {}
    for fam in (1, 3):
        ctx.set_family(fam)

        def step():
            ctx.call("pgps_pkfs_dev_f64", ctypes.c_long(n), ctypes.c_int(d), P(P0), P(Fd), P(Qd), P(H), ctypes.c_double(0.1), P(ys),
                     P(outs[0]), P(outs[1]), P(outs[2]), P(outs[3]), P(ll))
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            step()
        e1.record()
        torch.cuda.synchronize()
        res[fam] = e0.elapsed_time(e1) / 10
    ctx.set_family(0)
    print(f"d = {d}: lane-chunk {res[1]:.3f} ms, row-cooperative {res[3]:.3f} ms  (2^18 steps, fp64)")
