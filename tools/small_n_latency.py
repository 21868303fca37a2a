#!/usr/bin/env python3
"""Wall-clock latency of the StateSpaceGP calls the reference's experiments make at their sizes (N = 200 .. 32768; the
reference's toy mesh, toy_models/speed_and_stability.py:73, runs 2^12 .. 2^15 training x prediction points, its MCMC
drivers N ~ 10^3): objective, objective + gradient, predict_f -- per call, host overheads included -- and BASELINE
config c1 (Matern-3/2, N = 4096 training + 1024 query points).  Usage: python tools/small_n_latency.py [--quick]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "parallel-gps_amd"))
from pssgp.kernels import Matern32, Matern52, RBF
from pssgp.model import StateSpaceGP


def bench(fn, reps=50, warm=5):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        ts.append((time.perf_counter() - t0) / reps * 1e6)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def main():
    quick = "--quick" in sys.argv
    rng = np.random.default_rng(0)
    print("median (min) wall-clock microseconds per call over 5 rounds of 50 calls", flush=True)
    # BASELINE config c1
    n, k = 4096, 1024
    t = np.sort(rng.uniform(0, 40, n)); y = np.sin(t) + 0.3 * rng.standard_normal(n)
    tq = np.sort(rng.uniform(0, 40, k))[:, None]
    gp = StateSpaceGP((t[:, None], y[:, None]), Matern32(1., 0.5), noise_variance=0.1, parallel=True)
    out = []
    for label, fn in (("ll", gp.maximum_log_likelihood_objective), ("predict_f", lambda: gp.predict_f(tq)),
                      ("ll + predict_f", lambda: (gp.maximum_log_likelihood_objective(), gp.predict_f(tq))),
                      ("ll+grad", gp.log_likelihood_and_grad)):
        med, mn = bench(fn)
        out.append(f"{label} {med:7.1f} ({mn:7.1f})")
    print(f"c1 Matern32 N={n} K={k}:  " + "   ".join(out), flush=True)
    # the same with a NEW hyper-parameter setting at every call, as an optimiser or sampler makes them: the SDE is rebuilt
    # (get_sde, closed-form discretisation form, packing) and nothing is reused from the call before
    state = {"i": 0}

    def fresh(fn):
        def run():
            state["i"] += 1
            gp.kernel.lengthscales = 0.5 * (1.0 + 1e-9 * state["i"])
            return fn()
        return run
    out = []
    for label, fn in (("ll", gp.maximum_log_likelihood_objective), ("predict_f", lambda: gp.predict_f(tq)),
                      ("predict_f then ll", lambda: (gp.predict_f(tq), gp.maximum_log_likelihood_objective())),
                      ("ll+grad", gp.log_likelihood_and_grad)):
        med, mn = bench(fresh(fn))
        out.append(f"{label} {med:7.1f} ({mn:7.1f})")
    print(f"   new setting every call:   " + "   ".join(out), flush=True)
    kernels = (("Matern32", lambda: Matern32(1., 0.5)), ("Matern52", lambda: Matern52(1., 0.5)),
               ("RBF6", lambda: RBF(1., 0.5, order=6, balancing_iter=5)))
    for name, mk in kernels[:1] if quick else kernels:
        for n in (200, 1000, 4096, 10000, 32768):
            t = np.sort(rng.uniform(0, 10, n)); y = np.sin(t) + 0.3 * rng.standard_normal(n)
            gp = StateSpaceGP((t[:, None], y[:, None]), mk(), noise_variance=0.1, parallel=True)
            tq = np.sort(rng.uniform(0, 10, max(50, n // 4)))[:, None]
            out = []
            for label, fn in (("ll", gp.maximum_log_likelihood_objective), ("ll+grad", gp.log_likelihood_and_grad),
                              ("predict_f", lambda: gp.predict_f(tq))):
                med, mn = bench(fn, reps=30 if name != "Matern32" else 50)
                out.append(f"{label} {med:8.1f} ({mn:8.1f})")
            print(f"{name:9s} N={n:6d}  " + "   ".join(out), flush=True)


if __name__ == "__main__":
    main()
