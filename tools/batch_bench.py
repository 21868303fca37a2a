"""Throughput of the batched log-likelihood (pgps_gp_ll_batch_dev_f64) and of the gradient call at the
reference's realistic series lengths: prints one JSON line per (N, B).  Usage: python tools/batch_bench.py"""
import ctypes
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "parallel-gps_amd"))
from pssgp import _backend as B  # noqa: E402
from pssgp.kernels import Matern32  # noqa: E402

dev = torch.device("cuda:0")
ctx = B.get_context(0)
ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())
HP = lambda a: a.ctypes.data_as(ctypes.c_void_p)

for n, nb in [(1000, 1), (1000, 256), (1000, 4096), (10000, 1), (10000, 256), (10000, 2048), (100000, 1), (100000, 64),
              (100000, 512)]:
    rng = np.random.RandomState(0)
    t = np.cumsum(0.05 * (0.5 + rng.rand(n)))
    y = np.sin(t) + 0.3 * rng.randn(n)
    th = np.exp(rng.uniform(-0.5, 0.5, (nb, 3)))
    rows = []
    for v, l, r in th:
        sde = Matern32(v, l).get_sde()
        lam, N1, N2 = B.nilpotent_form(sde.F)
        rows.append(np.concatenate([[lam], N1.ravel(), N2.ravel(), np.asarray(sde.P0).ravel(), np.asarray(sde.H).ravel(), [r]]))
    packed = np.ascontiguousarray(np.stack(rows))
    ts_d, ys_d = torch.tensor(t, device=dev), torch.tensor(y, device=dev)
    ll_d = torch.zeros(nb, dtype=torch.float64, device=dev)

    def step():
        ctx.call("pgps_gp_ll_batch_dev_f64", ctypes.c_int(nb), ctypes.c_long(n), ctypes.c_int(2), HP(packed), P(ts_d),
                 ctypes.c_double(0.0), P(ys_d), P(ll_d))

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 30
    e0.record()
    for _ in range(reps):
        step()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(json.dumps({"N": n, "B": nb, "ms_per_call": round(ms, 4), "model_steps_per_s": n * nb / ms * 1e3,
                      "us_per_model": round(ms * 1e3 / nb, 3), "chunk": ctx.get_chunk(n)}))
