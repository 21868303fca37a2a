#!/usr/bin/env python3
"""The reference's CO2 experiment (pssgp/experiments/co2/mcmc.py:42-92: Periodic(order 3) * Matern32 + Matern32, d = 18, the
merged Mauna Loa records) on the HIP backend: what one evaluation of the posterior and its gradient -- one leapfrog step of
its HMC / NUTS chains -- costs at a NEW hyper-parameter setting every time (get_sde and its derivatives included), and a short
HMC chain.  Data: tests/golden/real/ (the reference's own files).  GPU box: python tools/co2_hmc_timing.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "parallel-gps_amd"))
from pssgp.experiments import real_data as RD
from pssgp.model import StateSpaceGP


def main():
    data_dir = os.path.join(ROOT, "tests", "golden", "real")
    t, y = RD.load_co2(data_dir, 3192)[:2]
    t, y = np.asarray(t, np.float64).reshape(-1), np.asarray(y, np.float64).reshape(-1)
    gp = StateSpaceGP((t[:, None], y[:, None]), RD.co2_covariance(3), noise_variance=0.05, parallel=True)
    priors, fixed = RD.co2_setup(gp)
    post = RD.Posterior(gp, priors, fixed)
    u = post.u0()
    rng = np.random.default_rng(0)
    lp, g = post(u)
    print(f"CO2 kernel: d = {gp.kernel.get_sde().F.shape[0]}, N = {t.size}, {len(post.free)} free parameters; log-posterior {lp:.4f}")
    for method in ("adjoint", "differences"):
        orig = gp.log_likelihood_and_grad
        gp.log_likelihood_and_grad = lambda wrt=None, m=method, f=orig: f(wrt=wrt, method=m)
        post(u)
        reps = 30 if method == "adjoint" else 5
        t0 = time.perf_counter()
        for _ in range(reps):
            post(u + 1e-4 * rng.standard_normal(u.size))          # a new setting every call, as a leapfrog step makes them
        ms = (time.perf_counter() - t0) / reps * 1e3
        print(f"log-posterior + gradient at a new setting, {method:11s}: {ms:8.2f} ms per evaluation (= per leapfrog step)", flush=True)
        gp.log_likelihood_and_grad = orig
    t0 = time.perf_counter()
    samples, acc = RD.hmc(gp, priors, fixed, n_samples=20, n_burnin=5, step_size=0.002, n_leapfrogs=10, seed=1)
    dt = time.perf_counter() - t0
    print(f"HMC: 25 iterations x 10 leapfrogs in {dt:.2f} s = {dt / 250 * 1e3:.2f} ms per leapfrog; acceptance {acc:.2f}")


if __name__ == "__main__":
    main()
