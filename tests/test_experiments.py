"""The toy-signal experiment drivers (pssgp/experiments/toy.py; reference protocols
pssgp/experiments/toy_models/{common,speed_and_stability,mcmc}.py, pssgp/toymodels/data_funcs.py)."""
import numpy as np
import pytest

from oracle import np_oracle as O


def test_toy_signals_and_data():
    from pssgp.experiments import toy
    t = np.linspace(0.0, 4.0, 1001)
    assert np.allclose(toy.sinu(t), np.sin(np.pi * t) + np.sin(2 * np.pi * t) + np.cos(3 * np.pi * t))
    r = toy.rect(t)
    assert set(np.unique(r)) == {0.0, 0.4, 0.6, 1.0}
    assert r[0] == 0.0 and r[-1] == 0.4 and r[np.searchsorted(t, 4 * 0.25)] == 1.0 and r[np.searchsorted(t, 4 * 0.58)] == 0.6
    c = toy.comp_sinu(t)
    assert np.all(np.isfinite(c)) and c.min() >= 0.0 and c.max() <= 1.0
    # obs_noise keeps the reference's draw: x + sqrt(r) * N(x, r)  ==  x (1 + sqrt(r)) + r z
    x = toy.sinu(t)
    y = toy.obs_noise(x, 0.5, seed=3)
    z = np.random.RandomState(3).normal(0.0, 1.0, x.shape[0])
    assert np.allclose(y, x * (1 + np.sqrt(0.5)) + 0.5 * z)
    tt, ft, tp, ftp, yy = toy.get_data(0, 128, 64)
    assert tt.shape == (128, 1) and tp.shape == (64, 1) and yy.shape == (128, 1) and ftp.shape == (64, 1)
    assert tt[0, 0] == 0.0 and tt[-1, 0] == 4.0


def test_reference_import_paths_of_the_helpers():
    """`from pssgp.toymodels import sinu, obs_noise` and `from pssgp.misc_utils import rmse` are what the reference's
    experiment scripts and notebook write (experiments/toy_models/common.py:8, speed_and_stability.py:18)."""
    from pssgp.toymodels import sinu, comp_sinu, rect, obs_noise
    from pssgp.misc_utils import rmse
    from pssgp.experiments import toy
    assert sinu is toy.sinu and comp_sinu is toy.comp_sinu and rect is toy.rect and obs_noise is toy.obs_noise
    assert rmse(np.array([[1.0], [2.0], [3.0]]), np.array([1.0, 2.0, 5.0])) == pytest.approx(np.sqrt(4.0 / 3.0))


@pytest.mark.gpu
def test_sequential_mesh_matches_oracle_posterior():
    """SSGP (sequential host recursion; the discretisation still runs on the GPU) cell of the mesh:
    predictions equal the oracle's, RMSE computed the reference's way."""
    from pssgp.experiments import toy
    from pssgp.kernels import Matern32
    t, _, tp, ftp, y = toy.get_data(1, 300, 100)
    gp = toy.get_model("SSGP", (t, y), 0.5, Matern32())
    mean, var = gp.predict_f(tp)
    mean_o, var_o = O.ssgp_predict_f(Matern32().get_sde(), t.ravel(), y.ravel(), 0.5, tp.ravel(), parallel=False)
    assert np.max(np.abs(mean[:, 0] - mean_o)) < 1e-9 and np.max(np.abs(var[:, 0] - var_o)) < 1e-9
    assert abs(toy.rmse(mean, ftp) - np.sqrt(np.mean((mean_o - ftp.ravel()) ** 2))) < 1e-9


@pytest.mark.gpu
def test_mesh_parallel_equals_sequential():
    from pssgp.experiments import toy
    kw = dict(cov="Matern32", mesh_size=2, n_seeds=2, log2_lo=9, log2_hi=11)
    s1, e_par, t_par = toy.speed_and_stability(model="PSSGP", **kw)
    s2, e_seq, _ = toy.speed_and_stability(model="SSGP", **kw)
    assert np.array_equal(s1, s2) and np.all(np.isfinite(t_par))
    np.testing.assert_allclose(e_par, e_seq, rtol=1e-8, atol=1e-10)


@pytest.mark.gpu
def test_log_posterior_gradient_in_unconstrained_space():
    from pssgp.experiments import toy
    from pssgp.kernels import Matern32
    t, _, _, _, y = toy.get_data(42, 2000, 1)
    gp = toy.get_model("PSSGP", (t, y), 0.5, Matern32())
    u = np.array([0.3, -0.2, 0.1])
    lp, g = toy.log_posterior_and_grad(gp, u)
    fd = np.zeros(3)
    for k in range(3):
        h = 1e-5
        up, um = u.copy(), u.copy()
        up[k] += h
        um[k] -= h
        fd[k] = (toy.log_posterior_and_grad(gp, up)[0] - toy.log_posterior_and_grad(gp, um)[0]) / (2 * h)
    np.testing.assert_allclose(g, fd, rtol=1e-6, atol=1e-6)


@pytest.mark.gpu
def test_hmc_samples_the_posterior():
    """A short chain moves, is accepted at a healthy rate, and ends near the posterior mode found by
    a grid of batched log-likelihood evaluations."""
    from pssgp.experiments import toy
    from pssgp.kernels import Matern32
    t, _, _, _, y = toy.get_data(42, 1500, 1)
    gp = toy.get_model("PSSGP", (t, y), 0.5, Matern32())
    samples, acc = toy.hmc(gp, n_samples=150, n_burnin=150, step_size=0.03, n_leapfrogs=8, seed=1)
    assert samples.shape == (150, 3) and 0.4 < acc <= 1.0
    assert np.all(samples > 0) and samples.std(axis=0).min() > 0
    lls = gp.log_likelihood_batch(np.vstack([samples.mean(axis=0), [1.0, 1.0, 0.5], [5.0, 0.05, 2.0]]))
    assert lls[0] > lls[1] - 5.0 and lls[0] > lls[2]


# ---- real-data drivers (pssgp/experiments/real_data.py) on synthetic files in the reference's formats -------------
def _write_sunspots(path, n=420, seed=0):
    rng = np.random.default_rng(seed)
    months = np.arange(np.datetime64("1749-01"), np.datetime64("1749-01") + np.timedelta64(n, "M"), np.timedelta64(1, "M"))
    days = (months + np.timedelta64(1, "M")).astype("datetime64[D]") - np.timedelta64(1, "D")     # month ends
    t = (days - days[0]).astype(float) / 365.2425
    vals = np.maximum(0.0, 80.0 + 70.0 * np.sin(2 * np.pi * t / 11.0) + 25.0 * rng.standard_normal(n))
    with open(path / "sunspots.csv", "w") as f:
        f.write("id,date,sunspots\n")
        for i, (d, v) in enumerate(zip(days, vals)):
            f.write(f"{i},{d},{v:.1f}\n")
    return t, vals


def _write_co2(path, seed=1):
    rng = np.random.default_rng(seed)
    tw = 1990.0 + np.arange(300) / 52.0
    tm = 1975.0 + np.arange(200) / 12.0
    co2 = lambda t: 330.0 + 1.6 * (t - 1975.0) + 3.0 * np.sin(2 * np.pi * t) + 0.3 * rng.standard_normal(t.shape)
    yw, ym = co2(tw), co2(tm)
    yw[[7, 90]] = -999.99                                     # NOAA's missing-value marker
    with open(path / "co2_weekly_mlo.txt", "w") as f:
        f.write("# synthetic file in the column layout of NOAA's weekly record: yr mon day decimal ppm ...\n")
        for t, v in zip(tw, yw):
            f.write(f"{int(t)} 1 1 {t:.4f} {v:.2f} 7 0.0 0.0 0.0\n")
    with open(path / "co2_mm_mlo.txt", "w") as f:
        f.write("# synthetic file in the column layout of NOAA's monthly record: yr mon decimal average ...\n")
        for t, v in zip(tm, ym):
            f.write(f"{int(t)} 1 {t:.4f} {v:.2f} {v:.2f} 30 0.1 0.1\n")
    return tw, yw, tm, ym


def test_real_data_loaders(tmp_path):
    from pssgp.experiments import real_data as RD
    t_all, vals = _write_sunspots(tmp_path)
    t, y = RD.load_sunspots(str(tmp_path), 100)
    assert t.shape == (100, 1) and y.shape == (100, 1)
    assert np.allclose(t[:, 0], t_all[-100:], atol=1e-9) and np.allclose(y[:, 0], np.round(vals[-100:], 1))
    tw, yw, tm, ym = _write_co2(tmp_path)
    t, y = RD.load_co2(str(tmp_path), 10000)
    assert t.shape[0] == 300 + 200 - 2 and np.all(np.diff(t[:, 0]) >= 0) and np.all(y > 0)
    t2, y2 = RD.load_co2(str(tmp_path), 50)
    assert np.array_equal(t2, t[-50:]) and np.array_equal(y2, y[-50:])


def test_co2_model_setup():
    """Kernel structure, state dimension and the trained / fixed split of co2/mcmc.py:35-65."""
    from pssgp.experiments import real_data as RD
    from pssgp.model import StateSpaceGP
    t = np.linspace(2000.0, 2001.0, 20)[:, None]
    gp = StateSpaceGP((t, np.zeros_like(t)), RD.co2_covariance(3), 0.05, parallel=False)
    assert gp.kernel.get_sde().F.shape[0] == 18
    names = [(type(o).__name__, n) for o, n in gp.trainable_parameters()]
    assert names == [("Periodic", "period"), ("SquaredExponential", "variance"), ("SquaredExponential", "lengthscales"),
                     ("Matern32", "variance"), ("Matern32", "lengthscales"), ("Matern32", "variance"),
                     ("Matern32", "lengthscales"), ("StateSpaceGP", "noise_variance")]
    priors, fixed = RD.co2_setup(gp)
    assert fixed == {0, 1, 7}
    assert priors == {2: (5., 1.), 3: (1e-1, 1e-3), 4: (50., 10.), 5: (1., 0.1), 6: (100., 50.)}
    assert RD.co2_covariance(2).get_sde().F.shape[0] == 14


@pytest.mark.gpu
def test_sunspot_map_and_co2_hmc_on_synthetic_files(tmp_path):
    from pssgp.experiments import real_data as RD
    from pssgp.model import StateSpaceGP
    _write_sunspots(tmp_path, n=600, seed=3)
    _write_co2(tmp_path)
    # MAP: the posterior density rises, the optimiser ends at a stationary point, predict_f runs on 30 x N points
    t, y = RD.load_sunspots(str(tmp_path), 500)
    gp = StateSpaceGP((t, y), RD.sunspot_covariance(), 10.0, parallel=True)
    post = RD.Posterior(gp, RD.sunspot_priors(10.0))
    lp0, g0 = post(post.u0())
    # the objective's own gradient against central differences of its value
    u = post.u0()
    for i in range(3):
        e = np.zeros(3); e[i] = 1e-4
        fd = (post(u + e)[0] - post(u - e)[0]) / 2e-4
        assert abs(fd - g0[i]) < 1e-4 * max(1.0, abs(g0[i]))
    out = RD.sunspot_map(str(tmp_path), n_training=500, maxiter=60)
    assert -out["neg_log_posterior"] > lp0 and out["predict_points"] == 500 * 30
    assert 0.0 < out["lengthscales"] < 100.0 and out["noise_variance"] > 0.0 and np.isfinite(out["max_std"])
    # CO2 kernel at order 1 (d = 10) on the general-LTI path: a few HMC iterations move, fixed parameters stay
    res = RD.co2_hmc(str(tmp_path), n_training=300, qp_order=1, n_samples=6, n_burnin=4, step_size=0.002)
    nuts = RD.co2_hmc(str(tmp_path), n_training=300, qp_order=1, n_samples=4, n_burnin=2, step_size=0.002, mcmc="NUTS")
    assert np.all(np.isfinite(nuts["posterior_mean"])) and 0 < nuts["acceptance"] <= 8     # (mean tree depth for NUTS)
    assert res["state_dim"] == 10 and len(res["posterior_mean"]) == 8
    assert res["posterior_mean"][0] == 1.0 and res["posterior_std"][0] == 0.0        # the period is not trained
    assert abs(res["posterior_mean"][7] - 0.05) < 1e-12 and all(np.isfinite(res["posterior_mean"]))


def test_reference_named_enums_and_factories():
    """pssgp/experiments/common.py:21-71 under the same names: enumerations, covariance factory (QP = Periodic over a
    squared-exponential base kernel), model factory (the dense GPflow GPR is not part of this backend)."""
    from pssgp.experiments.common import CovarianceEnum, MCMC, ModelEnum, get_model, get_simple_covariance_function
    from pssgp.kernels import Matern52, Periodic, RBF
    from pssgp.experiments.toy import rmse
    assert [m.value for m in MCMC] == ["HMC", "MALA", "NUTS"] and [m.value for m in ModelEnum] == ["GP", "SSGP", "PSSGP"]
    assert isinstance(get_simple_covariance_function(CovarianceEnum.Matern52, variance=2., lengthscales=.5), Matern52)
    assert isinstance(get_simple_covariance_function("RBF", variance=1., lengthscales=1., order=4), RBF)
    qp = get_simple_covariance_function("QP", variance=1.5, lengthscales=0.7, period=2., order=2)
    assert isinstance(qp, Periodic) and qp.base_kernel.variance == 1.5 and qp.get_sde().F.shape == (6, 6)
    t = np.linspace(0., 1., 8)[:, None]
    for name, par in (("SSGP", False), ("PSSGP", True)):
        m = get_model(name, (t, np.sin(t)), 0.1, qp)
        assert m.parallel is par and m.noise_variance == 0.1
    with pytest.raises(NotImplementedError):
        get_model(ModelEnum.GP, (t, t), 0.1, qp)
    assert abs(rmse([1., 2.], [[1.], [4.]]) - np.sqrt(2.0)) < 1e-15


@pytest.mark.gpu
def test_factory_models_agree():
    from pssgp.experiments.common import get_model, get_simple_covariance_function
    from pssgp.experiments.toy import obs_noise, sinu
    t = np.sort(np.random.default_rng(4).uniform(0, 1, 300))
    y = obs_noise(sinu(t), 0.1, 4)
    lls = [float(get_model(name, (t[:, None], y[:, None]), 0.1,
                           get_simple_covariance_function("Matern32", variance=1., lengthscales=0.5)
                           ).maximum_log_likelihood_objective()) for name in ("SSGP", "PSSGP")]
    assert abs(lls[0] - lls[1]) < 1e-9 * abs(lls[0])
    assert abs(lls[0] - O.dense_gp(("matern32", 1., 0.5), t, y, 0.1)) < 1e-8 * abs(lls[0])


def test_mala_and_nuts_sample_a_gaussian():
    """The two other samplers of the reference's scripts (MALA, NUTS: experiments/common.py:95-117) on a correlated
    3-d Gaussian: means and covariances recovered."""
    from pssgp.experiments.toy import mala_chain, nuts_chain
    mu = np.array([1.0, -2.0, 0.5])
    S = np.array([[1.0, 0.6, 0.0], [0.6, 2.0, -0.3], [0.0, -0.3, 0.5]])
    Si = np.linalg.inv(S)
    f = lambda u: (-0.5 * (u - mu) @ Si @ (u - mu), -Si @ (u - mu))
    rng = np.random.RandomState(1)
    s, acc = mala_chain(f, np.zeros(3), 12000, 1000, 0.3, rng)
    assert 0.6 < acc <= 1.0 and np.max(np.abs(s.mean(0) - mu)) < 0.12 and np.max(np.abs(np.cov(s.T) - S)) < 0.15
    s, depth = nuts_chain(f, np.zeros(3), 2500, 300, 0.35, rng)
    assert 1.0 <= depth <= 8.0 and np.max(np.abs(s.mean(0) - mu)) < 0.12 and np.max(np.abs(np.cov(s.T) - S)) < 0.2


@pytest.mark.gpu
@pytest.mark.parametrize("mcmc", ["MALA", "NUTS"])
def test_run_chain_other_samplers_on_the_gpu_posterior(mcmc):
    from pssgp.experiments import toy
    from pssgp.experiments.common import MCMC
    t, _, _, _, y = toy.get_data(3, 400, 10)
    gp = toy.get_model("PSSGP", (t, y), 0.5, toy.get_covariance("Matern32", variance=1., lengthscales=1.))
    theta, diag = toy.run_chain(gp, MCMC(mcmc), n_samples=40, n_burnin=20, step_size=2e-4 if mcmc == "MALA" else 0.05, seed=2)
    # (MALA's drift is step_size / 2 * gradient and the posterior of 400 points is steep away from its mode: a small step)
    assert theta.shape == (40, 3) and np.all(np.isfinite(theta)) and np.all(theta > 0)
    assert np.std(theta[:, 0]) > 0          # the chain moves
    assert 0.0 < diag <= 8.0
