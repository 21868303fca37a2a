"""Toy-signal experiments (reference: pssgp/toymodels/data_funcs.py, pssgp/experiments/toy_models/*).

* test signals `sinu`, `comp_sinu`, `rect` and the observation model `obs_noise`
  (data_funcs.py:11-95 -- including its unusual noise draw `x + sqrt(r) N(x, r)`, kept so that seeds
  give the same kind of data);
* `get_data(seed, n_training, n_pred)`: both grids are `linspace(0, 4, n)` (toy_models/common.py:28-46);
* `speed_and_stability(...)`: the (n_training x n_pred) mesh of predict_f wall times and RMSEs against
  the noise-free signal (speed_and_stability.py:26-40,63-95), for the sequential host model ("SSGP")
  and the HIP model ("PSSGP");
* `hmc(...)`: Hamiltonian Monte Carlo over the unconstrained hyper-parameters with the reference's
  priors (toy_models/mcmc.py:30-45: Normal(1, 3) on variance and lengthscale, Normal(0.1, 1) on the
  noise variance, all on the unconstrained = softplus-inverse scale, GPflow's default positive
  transform), driven by `StateSpaceGP.log_likelihood_and_grad` (one dual-number pass per leapfrog
  step) instead of TensorFlow autodiff (experiments/common.py:95-133).

Command line:  python -m pssgp.experiments.toy mesh --cov Matern32 --model PSSGP --mesh-size 4
               python -m pssgp.experiments.toy hmc --n-training 4096 --n-samples 200
"""
import argparse
import json
import math
import time

import numpy as np

from .. import config
from ..kernels import Matern12, Matern32, Matern52, RBF
from ..model import StateSpaceGP


# ---- signals -------------------------------------------------------------------------------------
def sinu(t):
    return np.sin(np.pi * t) + np.sin(2.0 * np.pi * t) + np.cos(3.0 * np.pi * t)


def comp_sinu(t):
    return np.sin(7.0 * np.pi * np.cos(2.0 * np.pi * t ** 2)) ** 2 / (np.cos(5.0 * np.pi * t) + 2.0)


def rect(t):
    tau = (t - np.min(t)) / (np.max(t) - np.min(t))
    edges = np.linspace(1.0 / 6.0, 5.0 / 6.0, 5)
    levels = np.array([0.0, 1.0, 0.0, 0.6, 0.0, 0.4])
    return levels[np.searchsorted(edges, tau, side="right")].astype(t.dtype)


def obs_noise(x, r, seed=None):
    rng = np.random.RandomState(seed)
    return x + np.sqrt(r) * rng.normal(x, math.sqrt(r), (x.shape[0],)).astype(x.dtype)


SIGNALS = {"SINE": sinu, "COMPOSITE_SINE": comp_sinu, "RECT": rect}


def get_data(seed, n_training, n_pred, data_model="SINE", noise_variance=0.5):
    dtype = config.default_float()
    t = np.linspace(0, 4, n_training, dtype=dtype)
    t_pred = np.linspace(0, 4, n_pred, dtype=dtype)
    fun = SIGNALS[data_model]
    ft, ft_pred = fun(t), fun(t_pred)
    y = obs_noise(ft, noise_variance, seed)
    col = lambda a: a.reshape(-1, 1)
    return col(t), col(ft), col(t_pred), col(ft_pred), col(y)


def rmse(a, b):
    a, b = np.asarray(a, np.float64).reshape(-1), np.asarray(b, np.float64).reshape(-1)
    return float(np.sqrt(np.mean((a - b) ** 2)))


# ---- models ----------------------------------------------------------------------------------------
def get_covariance(name, rbf_order=6, rbf_balance_iter=10, **kw):
    name = name.lower()
    if name == "matern12":
        return Matern12(**kw)
    if name == "matern32":
        return Matern32(**kw)
    if name == "matern52":
        return Matern52(**kw)
    if name == "rbf":
        return RBF(order=rbf_order, balancing_iter=rbf_balance_iter, **kw)
    raise ValueError(f"covariance {name!r} not supported here (Matern12/32/52, RBF)")


def get_model(model, data, noise_variance, covariance, max_parallel=10000):
    model = model.upper()
    if model == "SSGP":
        return StateSpaceGP(data, covariance, noise_variance, parallel=False)
    if model == "PSSGP":
        return StateSpaceGP(data, covariance, noise_variance, parallel=True, max_parallel=max_parallel)
    raise ValueError("model must be SSGP (sequential, host) or PSSGP (parallel, HIP)")


# ---- speed and stability mesh ------------------------------------------------------------------------
def speed_and_stability(model="PSSGP", cov="Matern32", mesh_size=10, n_seeds=21, data_model="SINE",
                        noise_variance=0.5, log2_lo=12, log2_hi=15):
    """errors[i, j, seed], times[i, j, seed] over n_training, n_pred in logspace(2^lo, 2^hi, mesh_size)."""
    sizes = np.logspace(log2_lo, log2_hi, mesh_size, base=2).astype(int)
    errors = np.full((mesh_size, mesh_size, n_seeds), np.nan)
    times = np.full((mesh_size, mesh_size, n_seeds), np.nan)
    kern = get_covariance(cov)
    for i, n_training in enumerate(sizes):
        for j, n_pred in enumerate(sizes):
            for seed in range(n_seeds):
                t, _, t_pred, ft_pred, y = get_data(seed, int(n_training), int(n_pred), data_model, noise_variance)
                tic = time.perf_counter()
                gp = get_model(model, (t, y), noise_variance, kern, t.shape[0] + t_pred.shape[0])
                mean, _ = gp.predict_f(t_pred)
                times[i, j, seed] = time.perf_counter() - tic
                errors[i, j, seed] = rmse(mean, ft_pred)
    return sizes, errors, times


# ---- HMC over the unconstrained hyper-parameters -----------------------------------------------------
def _softplus(u):
    return np.logaddexp(0.0, u)


def _softplus_inv(x):
    return x + np.log(-np.expm1(-x))


PRIORS = ((1.0, 3.0), (1.0, 3.0), (0.1, 1.0))      # (mean, std) of variance, lengthscale, noise: unconstrained scale


def log_posterior_and_grad(gp, u):
    """log p(y | theta(u)) + log prior(u) and its gradient in u; theta = softplus(u)."""
    theta = _softplus(u)
    params = gp.trainable_parameters()
    for (o, n), v in zip(params, theta):
        setattr(o, n, float(v))
    ll, g = gp.log_likelihood_and_grad()
    dtheta = 1.0 / (1.0 + np.exp(-u))                # d softplus / du
    lp, glp = float(ll), np.asarray(g) * dtheta
    for k, (mu, sd) in enumerate(PRIORS[:len(u)]):
        lp += -0.5 * ((u[k] - mu) / sd) ** 2 - math.log(sd) - 0.5 * math.log(2.0 * math.pi)
        glp[k] += -(u[k] - mu) / sd ** 2
    return lp, glp


def hmc(gp, n_samples=1000, n_burnin=100, step_size=0.05, n_leapfrogs=10, seed=31415, adapt=True):
    """Plain HMC (identity mass matrix); returns constrained samples (n_samples, P) and the acceptance rate.

    The posterior of N observations has curvature ~N, so a fixed step (the reference's 0.05) is only
    stable for short series; with `adapt` the step is shrunk by 0.7 on every rejected burn-in proposal
    and grown by 1.05 on every accepted one, then frozen for the sampling phase."""
    rng = np.random.RandomState(seed)
    params = gp.trainable_parameters()
    u = _softplus_inv(np.array([getattr(o, n) for o, n in params], np.float64))
    lp, g = log_posterior_and_grad(gp, u)
    out, accepted = [], 0
    for it in range(n_samples + n_burnin):
        p0 = rng.standard_normal(u.shape)
        un, pn, gn, lpn = u.copy(), p0.copy(), g.copy(), lp
        ok = True
        pn = pn + 0.5 * step_size * gn
        for l in range(int(n_leapfrogs)):
            un = un + step_size * pn
            try:
                lpn, gn = log_posterior_and_grad(gp, un)
            except Exception:                       # a divergent proposal: reject
                ok = False
                break
            if l + 1 < int(n_leapfrogs):
                pn = pn + step_size * gn
        if ok and np.all(np.isfinite(gn)) and np.isfinite(lpn):
            pn = pn + 0.5 * step_size * gn
            log_acc = (lpn - 0.5 * pn @ pn) - (lp - 0.5 * p0 @ p0)
            if math.log(rng.uniform()) < log_acc:
                u, lp, g = un, lpn, gn
                if it >= n_burnin:
                    accepted += 1
                elif adapt:
                    step_size *= 1.05
            elif adapt and it < n_burnin:
                step_size *= 0.7
        elif adapt and it < n_burnin:
            step_size *= 0.7
        if it >= n_burnin:
            out.append(_softplus(u))
    for (o, n), v in zip(params, _softplus(u)):
        setattr(o, n, float(v))
    return np.array(out), accepted / max(1, n_samples)


# ---- the reference's other two samplers (experiments/common.py:95-117 picks HMC, MALA or NUTS from tfp.mcmc) ---------
def mala_chain(logp_and_grad, u0, n_samples, n_burnin, step_size, rng):
    """Metropolis-adjusted Langevin with tfp.mcmc's meaning of `step_size` (unit volatility): proposal
    u + step_size / 2 grad + sqrt(step_size) xi, Metropolis-Hastings correction with the asymmetric proposal densities
    N(u + step_size / 2 grad, step_size I).  Returns (samples, acceptance)."""
    u = np.asarray(u0, np.float64).copy()
    lp, g = logp_and_grad(u)
    out, accepted = [], 0
    for it in range(n_samples + n_burnin):
        prop = u + 0.5 * step_size * g + math.sqrt(step_size) * rng.standard_normal(u.shape)
        try:
            lpn, gn = logp_and_grad(prop)
            ok = np.isfinite(lpn) and np.all(np.isfinite(gn))
        except Exception:
            ok = False
        if ok:
            fwd = -np.sum((prop - u - 0.5 * step_size * g) ** 2) / (2.0 * step_size)
            bwd = -np.sum((u - prop - 0.5 * step_size * gn) ** 2) / (2.0 * step_size)
            if math.log(rng.uniform()) < lpn + bwd - lp - fwd:
                u, lp, g = prop, lpn, gn
                accepted += it >= n_burnin
        if it >= n_burnin:
            out.append(u.copy())
    return np.array(out), accepted / max(1, n_samples)


def nuts_chain(logp_and_grad, u0, n_samples, n_burnin, step_size, rng, max_depth=8):
    """No-U-Turn sampler (Hoffman & Gelman 2014, algorithm 3: slice variable, doubling, uniform sampling from the
    admissible set), identity mass matrix, fixed step.  Returns (samples, mean tree depth)."""
    def leap(u, r, g, eps):
        r = r + 0.5 * eps * g
        u = u + eps * r
        lp, g = logp_and_grad(u)
        return u, r + 0.5 * eps * g, g, lp

    def build(u, r, g, log_slice, v, j, eps):
        if j == 0:
            try:
                u1, r1, g1, lp1 = leap(u, r, g, v * eps)
                joint = lp1 - 0.5 * r1 @ r1
                if not (np.isfinite(joint) and np.all(np.isfinite(g1))):
                    joint = -np.inf
            except Exception:
                u1, r1, g1, lp1, joint = u, r, g, -np.inf, -np.inf
            n1 = int(log_slice <= joint)
            s1 = log_slice < joint + 1000.0
            return u1, r1, g1, u1, r1, g1, u1, g1, lp1, n1, s1
        um, rm, gm, up, rp, gp_, u1, g1, lp1, n1, s1 = build(u, r, g, log_slice, v, j - 1, eps)
        if s1:
            if v == -1:
                um, rm, gm, _, _, _, u2, g2, lp2, n2, s2 = build(um, rm, gm, log_slice, v, j - 1, eps)
            else:
                _, _, _, up, rp, gp_, u2, g2, lp2, n2, s2 = build(up, rp, gp_, log_slice, v, j - 1, eps)
            if n2 > 0 and rng.uniform() < n2 / max(n1 + n2, 1):
                u1, g1, lp1 = u2, g2, lp2
            du = up - um
            s1 = s2 and (du @ rm >= 0) and (du @ rp >= 0)
            n1 += n2
        return um, rm, gm, up, rp, gp_, u1, g1, lp1, n1, s1

    u = np.asarray(u0, np.float64).copy()
    lp, g = logp_and_grad(u)
    out, depths = [], []
    for it in range(n_samples + n_burnin):
        r0 = rng.standard_normal(u.shape)
        log_slice = lp - 0.5 * r0 @ r0 + math.log(rng.uniform())
        um, up, rm, rp, gm, gp_ = u, u, r0, r0, g, g
        j, n, s = 0, 1, True
        while s and j < max_depth:
            v = -1 if rng.uniform() < 0.5 else 1
            if v == -1:
                um, rm, gm, _, _, _, u1, g1, lp1, n1, s1 = build(um, rm, gm, log_slice, v, j, step_size)
            else:
                _, _, _, up, rp, gp_, u1, g1, lp1, n1, s1 = build(up, rp, gp_, log_slice, v, j, step_size)
            if s1 and n1 > 0 and rng.uniform() < min(1.0, n1 / n):
                u, g, lp = u1, g1, lp1
            n += n1
            du = up - um
            s = s1 and (du @ rm >= 0) and (du @ rp >= 0)
            j += 1
        if it >= n_burnin:
            out.append(u.copy())
            depths.append(j)
    return np.array(out), float(np.mean(depths)) if depths else 0.0


def run_chain(gp, mcmc="HMC", n_samples=1000, n_burnin=100, step_size=0.05, n_leapfrogs=10, seed=31415):
    """The sampler choice of the reference's scripts (experiments/common.py:95-117) over this module's posterior:
    returns constrained samples (n_samples, P) and a diagnostic (acceptance rate; mean tree depth for NUTS)."""
    mcmc = getattr(mcmc, "value", mcmc).upper()
    if mcmc == "HMC":
        return hmc(gp, n_samples, n_burnin, step_size, n_leapfrogs, seed)
    rng = np.random.RandomState(seed)
    params = gp.trainable_parameters()
    u0 = _softplus_inv(np.array([getattr(o, n) for o, n in params], np.float64))
    f = lambda u: log_posterior_and_grad(gp, u)
    if mcmc == "MALA":
        us, diag = mala_chain(f, u0, n_samples, n_burnin, step_size, rng)
    elif mcmc == "NUTS":
        us, diag = nuts_chain(f, u0, n_samples, n_burnin, step_size, rng)
    else:
        raise ValueError(f"sampler {mcmc!r}: HMC, MALA or NUTS")
    theta = _softplus(us)
    for (o, n), v in zip(params, theta[-1] if len(theta) else _softplus(u0)):
        setattr(o, n, float(v))
    return theta, diag


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    sub = ap.add_subparsers(dest="cmd", required=True)
    m = sub.add_parser("mesh")
    m.add_argument("--model", default="PSSGP")
    m.add_argument("--cov", default="Matern32")
    m.add_argument("--mesh-size", type=int, default=10)
    m.add_argument("--n-seeds", type=int, default=21)
    m.add_argument("--data-model", default="SINE")
    m.add_argument("--noise-variance", type=float, default=0.5)
    m.add_argument("--dtype", default="float64")
    m.add_argument("--out", default=None, help="npz file for (sizes, errors, times)")
    h = sub.add_parser("hmc")
    h.add_argument("--cov", default="Matern32")
    h.add_argument("--n-training", type=int, default=4096)
    h.add_argument("--n-samples", type=int, default=1000)
    h.add_argument("--n-burnin", type=int, default=100)
    h.add_argument("--step-size", type=float, default=0.05)
    h.add_argument("--n-leapfrogs", type=int, default=10)
    h.add_argument("--noise-variance", type=float, default=0.5)
    h.add_argument("--np-seed", type=int, default=42)
    h.add_argument("--no-adapt", action="store_true", help="keep the step size fixed during burn-in (as the reference does)")
    args = ap.parse_args(argv)
    if args.cmd == "mesh":
        config.set_default_float(getattr(np, args.dtype))
        sizes, errors, times = speed_and_stability(args.model, args.cov, args.mesh_size, args.n_seeds, args.data_model,
                                                   args.noise_variance)
        if args.out:
            np.savez(args.out, sizes=sizes, errors=errors, times=times)
        print(json.dumps({"model": args.model, "cov": args.cov, "sizes": sizes.tolist(),
                          "median_time_s": np.median(times, axis=2).round(6).tolist(),
                          "median_rmse": np.median(errors, axis=2).round(6).tolist()}))
    else:
        t, _, _, _, y = get_data(args.np_seed, args.n_training, 1, "SINE", args.noise_variance)
        gp = get_model("PSSGP", (t, y), args.noise_variance, get_covariance(args.cov), t.shape[0])
        tic = time.perf_counter()
        samples, acc = hmc(gp, args.n_samples, args.n_burnin, args.step_size, args.n_leapfrogs, adapt=not args.no_adapt)
        toc = time.perf_counter() - tic
        print(json.dumps({"cov": args.cov, "n_training": args.n_training, "seconds": round(toc, 3),
                          "acceptance": acc, "posterior_mean": samples.mean(axis=0).round(4).tolist(),
                          "posterior_std": samples.std(axis=0).round(4).tolist(),
                          "parameters": ["variance", "lengthscales", "noise_variance"]}))


if __name__ == "__main__":
    main()
