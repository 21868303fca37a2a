#!/usr/bin/env python3
"""One evaluation scenario at the reference's series lengths, for `rocprofv3 --kernel-trace --stats -- python3
tools/trace_small.py <scenario> [calls]`: which launches an evaluation is made of and what each one costs.
Scenarios: rbf6_ll, rbf6_grad, rbf6_predict (RBF order 6, N = 1000), rbf15_grad, per2_grad, c5_grad (N = 4096),
co2_ll, co2_grad (the reference's CO2 kernel, d = 18, N = 3192), m32_grad (Matern-3/2, N = 1000), c1_ll / c1_grad / c1_predict
(Matern-3/2, N = 4096; 1024 query points).
Prints the wall-clock microseconds per call as well."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "parallel-gps_amd"))
from pssgp.kernels import Matern32, Matern52, RBF, Periodic, SquaredExponential
from pssgp.model import StateSpaceGP


def make(name):
    rng = np.random.default_rng(0)
    if name.startswith("co2"):
        from pssgp.experiments.real_data import co2_covariance
        n = 3192
        t = np.cumsum(rng.uniform(0.5, 1.5, n)) * (1.0 / 52.0)
        y = 0.3 * np.sin(2 * np.pi * t) + 0.01 * t + 0.05 * rng.standard_normal(n)
        return StateSpaceGP((t[:, None], y[:, None]), co2_covariance(3), 0.05, parallel=True), t
    n = 4096 if name.startswith(("c5", "c1")) else 1000
    t = np.sort(rng.uniform(0, 10, n)); y = np.sin(t) + 0.3 * rng.standard_normal(n)
    if name.startswith("rbf6"):
        k = RBF(1., 0.5, order=6, balancing_iter=5)
    elif name.startswith("rbf15"):
        k = RBF(1., 0.5, order=15, balancing_iter=10)
    elif name.startswith("per2"):
        k = Periodic(SquaredExponential(1., 1.), period=1., order=2)
    elif name.startswith("c5"):
        k = Periodic(SquaredExponential(1., 1.), period=1., order=1) * Matern32(1., 1.) + Matern52(1., 1.)
    else:
        k = Matern32(1., 0.5)
    return StateSpaceGP((t[:, None], y[:, None]), k, noise_variance=0.1, parallel=True), t


def main():
    name = sys.argv[1]
    calls = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    gp, t = make(name)
    if name.endswith("_ll"):
        fn = gp.maximum_log_likelihood_objective
    elif name.endswith("_grad"):
        fn = gp.log_likelihood_and_grad
    else:
        tq = np.sort(np.random.default_rng(1).uniform(t[0], t[-1], max(50, t.size // 4)))[:, None]
        fn = lambda: gp.predict_f(tq)
    for _ in range(3):
        out = fn()
    t0 = time.perf_counter()
    for _ in range(calls):
        out = fn()
    us = (time.perf_counter() - t0) / calls * 1e6
    print(f"{name}: {us:.1f} us per call; result {np.asarray(out[0] if isinstance(out, tuple) else out).ravel()[:3]}", flush=True)


if __name__ == "__main__":
    main()
