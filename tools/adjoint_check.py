#!/usr/bin/env python3
"""Device adjoint statistics (pgps_lti_ll_grad_f64) against the numpy reverse sweep of oracle/np_grad.py, and the model's
gradient three ways (adjoint / batched differences / where it exists the dual-number pass).  GPU box: python tools/adjoint_check.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "parallel-gps_amd"))
from oracle import np_grad as G
from pssgp import _backend
from pssgp.kernels import Matern32, Matern52, RBF, Periodic, SquaredExponential
from pssgp.kernels.sde_grads import sde_with_grads
from pssgp.model import StateSpaceGP


def kernels():
    yield "rbf6", RBF(1.3, 0.7, order=6, balancing_iter=5)
    yield "per2", Periodic(SquaredExponential(1.3, 0.9), period=1.7, order=2)
    yield "m32+m52", Matern32(1.3, 0.7) + Matern52(0.6, 1.1)
    yield "m32*m52", Matern32(1.3, 0.7) * Matern52(0.6, 1.1)
    yield "c5", Periodic(SquaredExponential(1., 1.), period=1., order=1) * Matern32(1., 1.) + Matern52(1., 1.)
    yield "rbf15", RBF(1.3, 0.7, order=15, balancing_iter=10)
    yield "co2", Periodic(SquaredExponential(1.2, 0.8), period=1., order=3) * Matern32(1., 30.) + Matern32(2., 1.5)


def main():
    rng = np.random.default_rng(1)
    for n in (37, 300, 2500):
        t = np.sort(rng.uniform(0, 3 * n / 100, n)); y = np.sin(3 * t) + 0.3 * rng.standard_normal(n)
        y[rng.uniform(size=n) < 0.15] = np.nan
        for name, k in kernels():
            sde, grads = sde_with_grads(k)
            d = sde.F.shape[0]
            try:
                dev = _backend.lti_ll_grad(sde.F, sde.P0, sde.H, 0.1, t, y)
            except _backend.PgpsError as e:
                print(f"{name:8s} d={d:2d} N={n}: {e}")
                continue
            ref = G.ll_grad_stats(sde.F, sde.P0, sde.H, 0.1, t, y)
            errs = [abs(dev[0] - ref[0]) / abs(ref[0])] + [float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / (1e-300 + np.max(np.abs(b))))
                                                            for a, b in zip(dev[1:], ref[1:])]
            gd, gr = _backend.contract_grad_stats(dev, sde.H, grads), G.contract(ref, sde.H, grads)
            print(f"{name:8s} d={d:2d} N={n:5d}: rel err ll {errs[0]:.1e} Abar {errs[1]:.1e} Ubar {errs[2]:.1e} Hbar {errs[3]:.1e} "
                  f"Rbar {errs[4]:.1e}  grad {np.max(np.abs(gd - gr) / (1e-12 + np.abs(gr))):.1e}", flush=True)
    # the wave-cooperative kernels (the road of d = 17..32) forced for every kernel: cross-check of the two families
    ctx = _backend.get_context()
    ctx.set_family(2)
    try:
        n = 700
        t = np.sort(rng.uniform(0, 21, n)); y = np.sin(3 * t) + 0.3 * rng.standard_normal(n)
        y[rng.uniform(size=n) < 0.15] = np.nan
        for name, k in kernels():
            sde, grads = sde_with_grads(k)
            dev = _backend.lti_ll_grad(sde.F, sde.P0, sde.H, 0.1, t, y)
            ref = G.ll_grad_stats(sde.F, sde.P0, sde.H, 0.1, t, y)
            errs = [abs(dev[0] - ref[0]) / abs(ref[0])] + [float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / (1e-300 + np.max(np.abs(b))))
                                                            for a, b in zip(dev[1:], ref[1:])]
            print(f"family 2 {name:8s} d={sde.F.shape[0]:2d} N={n}: rel err ll {errs[0]:.1e} Abar {errs[1]:.1e} Ubar {errs[2]:.1e} "
                  f"Hbar {errs[3]:.1e} Rbar {errs[4]:.1e}", flush=True)
    finally:
        ctx.set_family(0)
    # the model's call, three ways, with timings
    n = 1000
    t = np.sort(rng.uniform(0, 10, n)); y = np.sin(t) + 0.3 * rng.standard_normal(n)
    for name, k in kernels():
        gp = StateSpaceGP((t[:, None], y[:, None]), k, noise_variance=0.1, parallel=True)
        out = {}
        for method in ("adjoint", "differences", "dual"):
            try:
                res = gp.log_likelihood_and_grad(method=method)
                for _ in range(3):
                    gp.log_likelihood_and_grad(method=method)
                t0 = time.perf_counter()
                for _ in range(10):
                    gp.log_likelihood_and_grad(method=method)
                out[method] = (res, (time.perf_counter() - t0) / 10 * 1e6)
            except Exception as e:          # noqa: BLE001
                out[method] = (None, repr(e)[:60])
        ll = float(gp.maximum_log_likelihood_objective())
        t0 = time.perf_counter()
        for _ in range(10):
            gp.maximum_log_likelihood_objective()
        ll_us = (time.perf_counter() - t0) / 10 * 1e6
        line = f"{name:8s} N={n}: ll {ll_us:7.1f} us"
        base = out["adjoint"][0]
        for method in ("adjoint", "differences", "dual"):
            res, us = out[method]
            if res is None:
                line += f" | {method}: {us}"
            else:
                rel = np.max(np.abs(res[1] - base[1]) / (1e-9 + np.abs(base[1]))) if base is not None else float("nan")
                line += f" | {method} {us:8.1f} us (ll diff {abs(float(res[0]) - ll):.1e}, grad vs adjoint {rel:.1e})"
        print(line, flush=True)


if __name__ == "__main__":
    main()
