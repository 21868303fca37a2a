#!/usr/bin/env python3
"""Wall-clock latency of the StateSpaceGP calls the reference's experiments make at their sizes (N = 200 .. 10^4):
objective, objective + gradient, predict_f -- per call, host overheads included.  Usage: python tools/small_n_latency.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "parallel-gps_amd"))
from pssgp.kernels import Matern32, Matern52, RBF
from pssgp.model import StateSpaceGP

rng = np.random.default_rng(0)
for name, k in (("Matern32", lambda: Matern32(1., 0.5)), ("Matern52", lambda: Matern52(1., 0.5)), ("RBF6", lambda: RBF(1., 0.5, order=6, balancing_iter=5))):
    for n in (200, 1000, 3000, 10000):
        t = np.sort(rng.uniform(0, 10, n)); y = np.sin(t) + 0.3 * rng.standard_normal(n)
        gp = StateSpaceGP((t[:, None], y[:, None]), k(), noise_variance=0.1, parallel=True)
        tq = np.sort(rng.uniform(0, 10, max(50, n // 4)))[:, None]
        out = []
        for label, fn in (("ll", gp.maximum_log_likelihood_objective), ("ll+grad", gp.log_likelihood_and_grad), ("predict_f", lambda: gp.predict_f(tq))):
            for _ in range(3): fn()
            reps = 30
            t0 = time.perf_counter()
            for _ in range(reps): fn()
            out.append(f"{label} {1e6 * (time.perf_counter() - t0) / reps:8.1f} us")
        print(f"{name:9s} N={n:6d}  " + "   ".join(out), flush=True)
