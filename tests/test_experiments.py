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
