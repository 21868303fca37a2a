"""Real-data experiments on the HIP backend (reference: pssgp/experiments/sunspot/{common,map}.py and
pssgp/experiments/co2/{common,mcmc}.py).

* loaders for the two data sets the reference ships under experiments/: `sunspots.csv` (id, date, monthly mean
  sunspot number) and NOAA's `co2_weekly_mlo.txt` / `co2_mm_mlo.txt`.  The files are NOT part of this repository
  (and do not exist on the GPU box): pass `--data-dir`; `tests/test_experiments.py` writes tiny synthetic files in
  the same formats;
* the reference's covariance functions and priors: Matern-3/2 (variance 5500, lengthscale 5) for the sunspots
  (sunspot/map.py:44-56), `Periodic(SE(5, 1), period 1, order q) * Matern32(0.1, 50) + Matern32(1, 100)` for CO2
  (co2/mcmc.py:42-65; q = 3 gives state dimension 18, q <= 2 stays within the row-cooperative kernels' 16);
* `map_fit`: maximum a posteriori hyper-parameters by scipy's BFGS on the unconstrained (softplus) scale with the
  gradient from `StateSpaceGP.log_likelihood_and_grad` -- what sunspot/map.py:77-86 does with gpflow's Scipy
  optimiser and TensorFlow autodiff; `hmc`: the sampler of toy.py with per-parameter priors and fixed parameters.

Command line:  python -m pssgp.experiments.real_data sunspot-map --data-dir DIR --n-training 3200
               python -m pssgp.experiments.real_data co2-hmc --data-dir DIR --n-samples 100    (--qp-order 3, the reference's, by default)
"""
import argparse
import csv
import json
import math
import os
import time

import numpy as np

from ..kernels import Matern32, Periodic, SquaredExponential
from ..model import StateSpaceGP


# ---- data ----------------------------------------------------------------------------------------------
def load_sunspots(data_dir, n_training):
    """(t (n, 1) in years since the first record, y (n, 1)): the last `n_training` rows of sunspots.csv
    (sunspot/common.py:29-33; a numpy 'Y' is 365.2425 days)."""
    dates, vals = [], []
    with open(os.path.join(data_dir, "sunspots.csv"), newline="") as f:
        for row in csv.DictReader(f):
            dates.append(np.datetime64(row["date"]))
            vals.append(float(row["sunspots"]))
    d = np.array(dates, dtype="datetime64[D]")
    t = (d - d[0]).astype(np.float64) / 365.2425
    y = np.array(vals, np.float64)
    return t[-n_training:, None], y[-n_training:, None]


def load_co2(data_dir, n_training):
    """(t (n, 1) decimal years, y (n, 1) ppm): weekly and monthly Mauna Loa records merged, invalid (negative)
    entries dropped, sorted in time, the last `n_training` kept (co2/common.py:31-51)."""
    weekly = np.loadtxt(os.path.join(data_dir, "co2_weekly_mlo.txt"))[:, 3:5]
    monthly = np.loadtxt(os.path.join(data_dir, "co2_mm_mlo.txt"))[:, 2:4]
    data = np.concatenate([weekly, monthly], axis=0).astype(np.float64)
    data = data[~np.any(data < 0, axis=1)]
    data = data[np.argsort(data[:, 0], kind="stable")]
    data = data[-n_training:]
    return data[:, 0, None], data[:, 1, None]


# ---- models --------------------------------------------------------------------------------------------
def sunspot_covariance():
    return Matern32(variance=5500., lengthscales=5.)


def sunspot_priors(noise_variance):
    """Normal(mean, std) on the constrained value, in trainable_parameters() order (sunspot/map.py:32-56)."""
    return {"variance": (5500., 5500.), "lengthscales": (5., 5.), "noise_variance": (noise_variance, noise_variance)}


def co2_covariance(qp_order=3):
    base = SquaredExponential(variance=5., lengthscales=1.)
    return Periodic(base, period=1., order=qp_order) * Matern32(variance=1e-1, lengthscales=50.) + \
        Matern32(variance=1., lengthscales=100.)


def co2_setup(gp):
    """(priors by parameter index, fixed parameter indices) of co2/mcmc.py:35-65: the periodic base kernel's
    variance, the period and the noise variance are not trained; Normal priors on the rest."""
    params = gp.trainable_parameters()
    names = [(type(o).__name__, n) for o, n in params]
    priors, fixed = {}, set()
    seen_m32 = 0
    for i, (cls, n) in enumerate(names):
        if cls == "Periodic" and n == "period":
            fixed.add(i)
        elif cls == "SquaredExponential":
            if n == "variance":
                fixed.add(i)
            else:
                priors[i] = (5., 1.)
        elif cls == "Matern32":
            # leaves in kernel order: the damping Matern-3/2 of the product first, the trend Matern-3/2 second
            which = seen_m32 // 2
            seen_m32 += 1
            priors[i] = {("variance", 0): (1e-1, 1e-3), ("lengthscales", 0): (50., 10.),
                         ("variance", 1): (1., 0.1), ("lengthscales", 1): (100., 50.)}[(n, which)]
        elif n == "noise_variance":
            fixed.add(i)
    return priors, fixed


# ---- objective ------------------------------------------------------------------------------------------
def _softplus(u):
    return np.logaddexp(0.0, u)


def _softplus_inv(x):
    return x + np.log(-np.expm1(-x))


class Posterior:
    """log p(y | theta) + sum_i [log Normal(theta_i; mu_i, sd_i) + log |d theta_i / d u_i|] over the free
    parameters, theta = softplus(u) -- gpflow's log_posterior_density with priors on the constrained values
    (training_loss is its negative)."""

    def __init__(self, gp, priors=None, fixed=()):
        self.gp = gp
        self.params = gp.trainable_parameters()
        names = [n for _, n in self.params]
        self.priors = {}
        for key, val in (priors or {}).items():
            idx = key if isinstance(key, int) else names.index(key)
            self.priors[idx] = val
        self.free = [i for i in range(len(self.params)) if i not in set(fixed)]
        self.theta0 = np.array([getattr(o, n) for o, n in self.params], np.float64)

    def u0(self):
        return _softplus_inv(self.theta0[self.free])

    def set(self, u):
        theta = self.theta0.copy()
        theta[self.free] = _softplus(np.asarray(u, np.float64))
        for (o, n), v in zip(self.params, theta):
            setattr(o, n, float(v))
        return theta

    def __call__(self, u):
        u = np.asarray(u, np.float64)
        theta = self.set(u)
        ll, g = self.gp.log_likelihood_and_grad(wrt=self.free)      # (the dual-number path returns all of them anyway)
        sig = 1.0 / (1.0 + np.exp(-u))                      # d softplus / du
        lp, glp = float(ll), np.asarray(g, np.float64)[self.free] * sig
        for j, i in enumerate(self.free):
            lp += math.log(sig[j])
            glp[j] += 1.0 - sig[j]                          # d log sigmoid(u) / du
            if i in self.priors:
                mu, sd = self.priors[i]
                z = (theta[i] - mu) / sd
                lp += -0.5 * z * z - math.log(sd) - 0.5 * math.log(2.0 * math.pi)
                glp[j] += -z / sd * sig[j]
        return lp, glp


def map_fit(gp, priors=None, fixed=(), maxiter=100):
    """Maximum a posteriori fit; returns (constrained parameters, scipy result, seconds)."""
    from scipy.optimize import minimize
    post = Posterior(gp, priors, fixed)

    def loss(u):
        lp, g = post(u)
        return -lp, -g

    tic = time.perf_counter()
    res = minimize(loss, post.u0(), jac=True, method="BFGS", options=dict(maxiter=maxiter))
    seconds = time.perf_counter() - tic
    theta = post.set(res.x)
    return theta, res, seconds


def hmc(gp, priors=None, fixed=(), n_samples=1000, n_burnin=100, step_size=0.01, n_leapfrogs=10, seed=31415, adapt=True):
    """HMC over the free unconstrained parameters (the sampler of experiments/toy.py with this module's
    posterior); returns constrained samples of ALL parameters (n_samples, P) and the acceptance rate."""
    rng = np.random.RandomState(seed)
    post = Posterior(gp, priors, fixed)
    u = post.u0()
    lp, g = post(u)
    out, accepted = [], 0
    for it in range(n_samples + n_burnin):
        p0 = rng.standard_normal(u.shape)
        un, pn = u.copy(), p0 + 0.5 * step_size * g
        ok, lpn, gn = True, lp, g
        for l in range(int(n_leapfrogs)):
            un = un + step_size * pn
            try:
                lpn, gn = post(un)
            except Exception:
                ok = False
                break
            if not (np.isfinite(lpn) and np.all(np.isfinite(gn))):
                ok = False
                break
            pn = pn + (step_size if l + 1 < int(n_leapfrogs) else 0.5 * step_size) * gn
        took = False
        if ok and math.log(rng.uniform()) < (lpn - 0.5 * pn @ pn) - (lp - 0.5 * p0 @ p0):
            u, lp, g, took = un, lpn, gn, True
        if it < n_burnin and adapt:
            step_size *= 1.05 if took else 0.7
        elif took:
            accepted += 1
        if it >= n_burnin:
            out.append(post.set(u))
    post.set(u)
    return np.array(out), accepted / max(1, n_samples)


# ---- drivers ---------------------------------------------------------------------------------------------
def run_chain(gp, mcmc="HMC", priors=None, fixed=(), n_samples=1000, n_burnin=100, step_size=0.01, n_leapfrogs=10,
              seed=31415):
    """The sampler choice of the reference's mcmc scripts (experiments/common.py:95-117: HMC, MALA or NUTS) over this
    module's posterior; returns constrained samples of ALL parameters (n_samples, P) and a diagnostic (acceptance
    rate; mean tree depth for NUTS)."""
    from .toy import mala_chain, nuts_chain
    mcmc = getattr(mcmc, "value", mcmc).upper()
    if mcmc == "HMC":
        return hmc(gp, priors, fixed, n_samples, n_burnin, step_size, n_leapfrogs, seed)
    post = Posterior(gp, priors, fixed)
    rng = np.random.RandomState(seed)
    if mcmc == "MALA":
        us, diag = mala_chain(post, post.u0(), n_samples, n_burnin, step_size, rng)
    elif mcmc == "NUTS":
        us, diag = nuts_chain(post, post.u0(), n_samples, n_burnin, step_size, rng)
    else:
        raise ValueError(f"sampler {mcmc!r}: HMC, MALA or NUTS")
    return np.array([post.set(u) for u in us]), diag


def sunspot_map(data_dir, n_training=3200, noise_variance=10., n_interp_factor=30, maxiter=100):
    """sunspot/map.py: MAP fit of the Matern-3/2 model on the last n_training months, then predict_f on
    n_training * 30 interpolation times."""
    t, y = load_sunspots(data_dir, n_training)
    gp = StateSpaceGP((t, y), sunspot_covariance(), noise_variance, parallel=True)
    theta, res, seconds = map_fit(gp, sunspot_priors(noise_variance), maxiter=maxiter)
    tq = np.linspace(t[0, 0], t[-1, 0], t.shape[0] * n_interp_factor)[:, None]
    gp.predict_f(tq)
    tic = time.perf_counter()
    mean, var = gp.predict_f(tq)
    return dict(n_training=int(t.shape[0]), map_seconds=round(seconds, 3), iterations=int(res.nit),
                neg_log_posterior=float(res.fun), variance=theta[0], lengthscales=theta[1], noise_variance=theta[2],
                predict_points=int(tq.shape[0]), predict_seconds=round(time.perf_counter() - tic, 4),
                mean_range=[float(mean.min()), float(mean.max())], max_std=float(np.sqrt(var.max())))


def co2_hmc(data_dir, n_training=3192, qp_order=3, noise_variance=0.05, n_samples=1000, n_burnin=100, step_size=0.01,
            n_leapfrogs=10, mcmc="HMC"):
    t, y = load_co2(data_dir, n_training)
    gp = StateSpaceGP((t, y), co2_covariance(qp_order), noise_variance, parallel=True)
    priors, fixed = co2_setup(gp)
    tic = time.perf_counter()
    samples, acc = run_chain(gp, mcmc, priors, fixed, n_samples, n_burnin, step_size, n_leapfrogs)
    names = [f"{type(o).__name__}.{n}" for o, n in gp.trainable_parameters()]
    return dict(n_training=int(t.shape[0]), qp_order=qp_order, state_dim=int(gp.kernel.get_sde().F.shape[0]),
                seconds=round(time.perf_counter() - tic, 3), acceptance=acc, parameters=names,
                posterior_mean=samples.mean(axis=0).round(5).tolist(), posterior_std=samples.std(axis=0).round(5).tolist())


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    sub = ap.add_subparsers(dest="cmd", required=True)
    s = sub.add_parser("sunspot-map")
    s.add_argument("--data-dir", required=True)
    s.add_argument("--n-training", type=int, default=3200)
    s.add_argument("--noise-variance", type=float, default=10.)
    c = sub.add_parser("co2-hmc")
    c.add_argument("--data-dir", required=True)
    c.add_argument("--n-training", type=int, default=3192)
    c.add_argument("--qp-order", type=int, default=3)      # 3 = the reference (d = 18); up to 2 the batched kernels apply (d <= 14)
    c.add_argument("--noise-variance", type=float, default=0.05)
    c.add_argument("--n-samples", type=int, default=1000)
    c.add_argument("--n-burnin", type=int, default=100)
    c.add_argument("--step-size", type=float, default=0.01)
    c.add_argument("--mcmc", default="HMC", choices=["HMC", "MALA", "NUTS"])
    args = ap.parse_args(argv)
    if args.cmd == "sunspot-map":
        print(json.dumps(sunspot_map(args.data_dir, args.n_training, args.noise_variance)))
    else:
        print(json.dumps(co2_hmc(args.data_dir, args.n_training, args.qp_order, args.noise_variance, args.n_samples,
                                 args.n_burnin, args.step_size, mcmc=args.mcmc)))


if __name__ == "__main__":
    main()
