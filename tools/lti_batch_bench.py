"""Throughput of the batched general-LTI log-likelihood (pgps_lti_ll_batch_dev_f64: discretisation + parallel
filter of B models over one series) at the reference's realistic series lengths, for a d = 6 and the d = 11
(config c5) kernel: one JSON line per (kernel, N, B).  Usage: python tools/lti_batch_bench.py"""
import ctypes
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "parallel-gps_amd"))
from pssgp import _backend as B  # noqa: E402
from pssgp.kernels import Matern32, Matern52, RBF, Periodic, SquaredExponential  # noqa: E402

dev = torch.device("cuda:0")
ctx = B.get_context(0)
ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())
HP = lambda a: a.ctypes.data_as(ctypes.c_void_p)

kernels = {
    "rbf6 (d=6)": RBF(variance=1., lengthscales=1., order=6, balancing_iter=10),
    "c5 (d=11)": Periodic(SquaredExponential(1., 1.), period=1., order=1) * Matern32(1., 1.) + Matern52(1., 1.),
}
for kname, kern in kernels.items():
    sde = kern.get_sde()
    F, P0, H = (np.asarray(a, np.float64) for a in (sde.F, sde.P0, sde.H))
    d = F.shape[0]
    for n, nb in [(1000, 1), (1000, 64), (1000, 1024), (10000, 1), (10000, 64), (10000, 256), (100000, 1), (100000, 16),
                  (100000, 64)]:
        rng = np.random.RandomState(0)
        t = np.cumsum(0.05 * (0.5 + rng.rand(n)))
        y = np.sin(t) + 0.3 * rng.randn(n)
        rows = []
        for _ in range(nb):
            a, v, r = np.exp(rng.uniform(-0.5, 0.5, 3))
            rows.append(np.concatenate([(a * F).ravel(), (v * P0).ravel(), H.ravel(), [0.1 * r]]))
        packed = np.ascontiguousarray(np.stack(rows))
        ts_d, ys_d = torch.tensor(t, device=dev), torch.tensor(y, device=dev)
        ll_d = torch.zeros(nb, dtype=torch.float64, device=dev)

        def step():
            ctx.call("pgps_lti_ll_batch_dev_f64", ctypes.c_int(nb), ctypes.c_long(n), ctypes.c_int(d), HP(packed), P(ts_d),
                     P(ys_d), ctypes.c_double(0.0), P(ll_d))

        for _ in range(3):
            step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for _ in range(reps):
            step()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(json.dumps({"kernel": kname, "N": n, "B": nb, "ms_per_call": round(ms, 4),
                          "model_steps_per_s": n * nb / ms * 1e3, "us_per_model": round(ms * 1e3 / nb, 2),
                          "finite": bool(torch.isfinite(ll_d).all().item())}))
