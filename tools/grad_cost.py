#!/usr/bin/env python3
"""What the adjoint gradient costs next to one log-likelihood (both through the resident series, host overheads
included): RBF order 6 / 15, Periodic order 2, BASELINE c5's kernel (d = 11), the reference's CO2 kernel (d = 18), by series
length.  GPU box: python tools/grad_cost.py [--quick]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "parallel-gps_amd"))
from pssgp.kernels import Matern32, Matern52, RBF, Periodic, SquaredExponential
from pssgp.model import StateSpaceGP

KERNELS = {
    "rbf6": lambda: RBF(1., 0.5, order=6, balancing_iter=5),
    "per2": lambda: Periodic(SquaredExponential(1., 1.), period=1., order=2),
    "c5": lambda: Periodic(SquaredExponential(1., 1.), period=1., order=1) * Matern32(1., 1.) + Matern52(1., 1.),
    "rbf15": lambda: RBF(1., 0.5, order=15, balancing_iter=10),
    "co2": lambda: Periodic(SquaredExponential(1.2, 0.8), period=1., order=3) * Matern32(1., 30.) + Matern32(2., 1.5),
}


def bench(fn, reps):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        ts.append((time.perf_counter() - t0) / reps * 1e6)
    return sorted(ts)[1]


def main():
    quick = "--quick" in sys.argv
    rng = np.random.default_rng(0)
    sizes = (1000, 3192, 32768) if quick else (1000, 3192, 32768, 1 << 17, 1 << 20)
    print("median wall-clock microseconds per call (3 rounds); same hyper-parameter setting every call (memoised SDE)")
    for name, mk in KERNELS.items():
        for n in sizes:
            if name == "co2" and n > (1 << 17):
                continue
            t = np.cumsum(0.05 * rng.uniform(0.5, 1.5, n)); y = np.sin(t) + 0.3 * rng.standard_normal(n)
            gp = StateSpaceGP((t[:, None], y[:, None]), mk(), noise_variance=0.1, parallel=True)
            reps = 20 if n <= (1 << 17) else 5
            ll = bench(gp.maximum_log_likelihood_objective, reps)
            gr = bench(gp.log_likelihood_and_grad, reps)
            npar = len(gp.trainable_parameters())
            print(f"{name:6s} d={gp.kernel.get_sde().F.shape[0]:2d} P={npar}  N={n:8d}  ll {ll:10.1f}   ll+grad (adjoint) {gr:10.1f}   ratio {gr / ll:5.2f}", flush=True)


if __name__ == "__main__":
    main()
