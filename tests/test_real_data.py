"""The reference's own experiment data (tests/golden/real/: sunspots.csv, the two Mauna Loa CO2 files) through the
reference's models: sunspot MAP (pssgp/experiments/sunspot/map.py: Matern-3/2, variance 5500, lengthscale 5, noise 350 as
in experiments/sunspots/map.sh, the last 3200 months, interpolation grid) and the CO2 quasi-periodic kernel of
pssgp/experiments/co2/mcmc.py:42-65 at the reference's order 3 (state dimension 18) on the last 3192 merged weekly +
monthly records.  The loaders are exercised on the real formats on the CPU; on the GPU the HIP path's log-likelihood,
gradient and predictions on these irregular time stamps are compared with the oracle."""
import os

import numpy as np
import pytest

from oracle import np_oracle as O

REAL = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "real")


def test_loaders_read_the_references_files():
    from pssgp.experiments import real_data as rd
    t, y = rd.load_sunspots(REAL, 3200)
    assert t.shape == y.shape == (3200, 1)
    assert np.all(np.diff(t[:, 0]) > 0) and abs(np.median(np.diff(t[:, 0])) - 1.0 / 12.0) < 2e-3     # months, in years
    assert 0.0 <= y.min() and 200.0 < y.max() < 400.0
    tall, _ = rd.load_sunspots(REAL, 10 ** 6)
    assert tall.shape[0] == 3235                                    # every month from 1749-01 on
    t, y = rd.load_co2(REAL, 3192)
    assert t.shape == y.shape == (3175, 1)                          # all valid weekly + monthly records (fewer than asked for)
    # merged weekly and monthly series: sorted, with near-coincident stamps (1e-4 years apart) -- tiny time steps
    assert np.all(np.diff(t[:, 0]) >= 0) and np.diff(t[:, 0]).min() < 1e-3 and 1958.0 < t[0, 0] and t[-1, 0] > 2015.0
    assert 300.0 < y.min() and y.max() < 430.0                      # ppm; the -999.99 markers are gone


@pytest.mark.gpu
def test_sunspot_map_model_on_the_real_months():
    from pssgp.experiments import real_data as rd
    from pssgp.model import StateSpaceGP
    t, y = rd.load_sunspots(REAL, 3200)
    kern = rd.sunspot_covariance()
    gp = StateSpaceGP((t, y), kern, noise_variance=350.0, parallel=True)
    sde = kern.get_sde()
    ll = float(gp.maximum_log_likelihood_objective())
    ll_o = O.ssgp_log_likelihood(sde, t[:, 0], y[:, 0], 350.0, parallel=False)
    assert abs(ll - ll_o) < 1e-9 * abs(ll_o)
    # the interpolation grid of sunspot/map.py:91-110 (30 points per observation interval, here thinned to 3)
    tq = np.linspace(t[0, 0], t[-1, 0], 3 * t.shape[0])
    mean, var = gp.predict_f(tq[:, None])
    mean_o, var_o = O.ssgp_predict_f(sde, t[:, 0], y[:, 0], 350.0, tq, parallel=False)
    assert np.max(np.abs(mean[:, 0] - mean_o)) < 1e-8 * np.max(np.abs(mean_o))
    assert np.max(np.abs(var[:, 0] - var_o)) < 1e-8 * np.max(np.abs(var_o))
    # MAP: a few BFGS iterations on the device gradient raise the posterior density (sunspot/map.py:74-86)
    post = rd.Posterior(gp, rd.sunspot_priors(350.0))
    loss_before = -post(post.u0())[0]
    theta, res, _ = rd.map_fit(gp, rd.sunspot_priors(350.0), maxiter=8)
    assert res.fun < loss_before - 1.0 and np.all(np.isfinite(theta))
    ll_fit = float(gp.maximum_log_likelihood_objective())
    assert abs(ll_fit - O.ssgp_log_likelihood(kern.get_sde(), t[:, 0], y[:, 0], gp.noise_variance, parallel=False)) < 1e-8 * abs(ll_fit)


@pytest.mark.gpu
@pytest.mark.parametrize("qp_order", [1, 3])
def test_co2_quasi_periodic_model_on_the_real_records(qp_order):
    from pssgp.experiments import real_data as rd
    from pssgp.model import StateSpaceGP
    t, y = rd.load_co2(REAL, 3192)
    y = y - np.mean(y)                      # co2/mcmc.py centres the series
    kern = rd.co2_covariance(qp_order)
    sde = kern.get_sde()
    assert sde.F.shape[0] == (18 if qp_order == 3 else 10)
    gp = StateSpaceGP((t - t[0], y), kern, noise_variance=0.05, parallel=True)
    tt = (t - t[0])[:, 0]
    ll = float(gp.maximum_log_likelihood_objective())
    ll_o = O.ssgp_log_likelihood(sde, tt, y[:, 0], 0.05, parallel=False)
    assert abs(ll - ll_o) < 1e-7 * abs(ll_o)
    tq = np.linspace(tt[0], tt[-1] + 2.0, 500)              # interpolation and two years of forecast
    mean, var = gp.predict_f(tq[:, None])
    mean_o, var_o = O.ssgp_predict_f(sde, tt, y[:, 0], 0.05, tq, parallel=False)
    assert np.max(np.abs(mean[:, 0] - mean_o)) < 1e-6 * max(1.0, np.max(np.abs(mean_o)))
    assert np.max(np.abs(var[:, 0] - var_o)) < 1e-6 * max(1.0, np.max(np.abs(var_o)))
