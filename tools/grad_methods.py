#!/usr/bin/env python3
"""ll + gradient of the Matern family and its small composites by forward-mode duals and by the adjoint
pass (section 4l): wall-clock microseconds per call, to place the automatic choice of `log_likelihood_and_grad`."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "parallel-gps_amd"))
from pssgp.kernels import Matern12, Matern32, Matern52
from pssgp.model import StateSpaceGP


def bench(fn, calls=60):
    for _ in range(5):
        fn()
    best = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(calls):
            fn()
        best.append((time.perf_counter() - t0) / calls * 1e6)
    return float(np.median(best))


kernels = {"m12": lambda: Matern12(1., 0.5), "m32": lambda: Matern32(1., 0.5), "m52": lambda: Matern52(1., 0.5),
           "m32+m52": lambda: Matern32(1., 0.5) + Matern52(0.5, 2.0), "m32*m52": lambda: Matern32(1., 0.5) * Matern52(0.5, 2.0)}
for name, mk in kernels.items():
    for n in ((200, 1000, 2048, 4096, 32768, 1 << 17, 1 << 20) if '+' not in name and '*' not in name else (200, 1000, 4096, 32768)):
        rng = np.random.default_rng(n)
        t = np.sort(rng.uniform(0, 10, n)); y = np.sin(t) + 0.3 * rng.standard_normal(n)
        gp = StateSpaceGP((t[:, None], y[:, None]), mk(), noise_variance=0.1, parallel=True)
        row = [f"{name:8s} N={n:6d}"]
        ref = None
        for method in ("dual", "adjoint"):
            try:
                ll, g = gp.log_likelihood_and_grad(method=method)
                us = bench(lambda: gp.log_likelihood_and_grad(method=method))
                gv = np.array([float(v) for v in (g.values() if isinstance(g, dict) else g)])
                if ref is None:
                    ref = gv
                row.append(f"{method} {us:7.1f} us (max rel diff to dual {np.abs(gv - ref).max() / np.abs(ref).max():.1e})")
            except Exception as e:      # noqa: BLE001
                row.append(f"{method}: {type(e).__name__} {e}")
        print("   ".join(row), flush=True)
