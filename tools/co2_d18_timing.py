#!/usr/bin/env python3
"""Timing of the reference's CO2 kernel (Periodic(order 3) * Matern32 + Matern32, d = 18; co2/mcmc.py:42-65) on the
wave-cooperative family, next to order 2 (d = 14) on the row-cooperative family: log-likelihood and predict_f at the
experiment's series length and at 2^17 steps.  Run on the GPU box: python tools/co2_d18_timing.py"""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "parallel-gps_amd"))
from pssgp.model import StateSpaceGP
from pssgp.experiments.real_data import co2_covariance


def timeit(f, n):
    f()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    return (time.perf_counter() - t0) / n * 1e3


for order in (2, 3):
    for N in (3192, 1 << 17):
        rng = np.random.default_rng(0)
        t = np.cumsum(rng.uniform(0.5, 1.5, N)) * (1.0 / 52.0)
        y = 0.3 * np.sin(2 * np.pi * t) + 0.01 * t + 0.05 * rng.standard_normal(N)
        gp = StateSpaceGP((t[:, None], y[:, None]), co2_covariance(order), 0.05, parallel=True)
        d = int(gp.kernel.get_sde().F.shape[0])
        tq = np.sort(rng.uniform(t[0], t[-1], N // 4))
        ll = float(gp.maximum_log_likelihood_objective())
        state = {"i": 0}

        def fresh_ll():                 # a new hyper-parameter setting per call, as a sampler makes them: get_sde() included
            state["i"] += 1
            gp.noise_variance = 0.05 * (1.0 + 1e-7 * state["i"])
            leaf = gp.trainable_parameters()[0]
            setattr(leaf[0], leaf[1], getattr(leaf[0], leaf[1]) * (1.0 + 1e-9))
            return gp.maximum_log_likelihood_objective()
        out = dict(qp_order=order, state_dim=d, N=N, ll=ll,
                   ll_ms=round(timeit(gp.maximum_log_likelihood_objective, 5), 3),
                   ll_new_setting_ms=round(timeit(fresh_ll, 20), 3),
                   predict_ms=round(timeit(lambda: gp.predict_f(tq[:, None]), 3), 3))
        print(json.dumps(out), flush=True)
