"""CPU tests of the host half of the adjoint gradient: the kernels' SDE derivatives in a frozen state basis
(pssgp/kernels/sde_grads.py), the numpy reverse sweep that checks the device (oracle/np_grad.py) -- both against
difference quotients of the oracle's own likelihood and, for the kernels whose state-space form is exact, of the dense
GP -- and the memo keys / data property of StateSpaceGP (hyper-parameters of any numeric type, reassigned data).
Reference: the gradients the reference takes with tf.GradientTape, tests/test_gp_vs_kfs.py:53-78."""
import numpy as np
import pytest

from oracle import np_grad as G
from oracle import np_oracle as O


def _kernels():
    from pssgp.kernels import Matern12, Matern32, Matern52, RBF, Periodic, SquaredExponential
    return {
        "m12": lambda: Matern12(1.3, 0.7),
        "m32": lambda: Matern32(1.3, 0.7),
        "m52": lambda: Matern52(1.3, 0.7),
        "rbf6": lambda: RBF(1.3, 0.7, order=6, balancing_iter=5),
        "per2": lambda: Periodic(SquaredExponential(1.3, 0.9), period=1.7, order=2),
        "m32+m52": lambda: Matern32(1.3, 0.7) + Matern52(0.6, 1.1),
        "m32*m52": lambda: Matern32(1.3, 0.7) * Matern52(0.6, 1.1),
        # BASELINE config c5's kernel (d = 11) and the reference's CO2 kernel (co2/mcmc.py:42-65, d = 18)
        "c5": lambda: Periodic(SquaredExponential(1., 1.), period=1., order=1) * Matern32(1., 1.) + Matern52(1., 1.),
        "co2": lambda: Periodic(SquaredExponential(1.2, 0.8), period=1., order=3) * Matern32(1., 30.) + Matern32(2., 1.5),
        "nested": lambda: (Matern32(1., 0.8) + Matern52(0.5, 2.0)) * Matern32(0.7, 1.5),
    }


def _series(n, seed=1, nan_frac=0.15):
    rng = np.random.default_rng(seed)
    t = np.sort(rng.uniform(0, 3, n))
    y = np.sin(3 * t) + 0.3 * rng.standard_normal(n)
    y[rng.uniform(size=n) < nan_frac] = np.nan
    return t, y


def _richardson(f, x0, rel=1e-4):
    h = rel * max(abs(x0), 1e-3)
    d1 = (f(x0 + h) - f(x0 - h)) / (2 * h)
    d2 = (f(x0 + 0.5 * h) - f(x0 - 0.5 * h)) / h
    return (4 * d2 - d1) / 3


@pytest.mark.parametrize("name", list(_kernels()))
def test_sde_with_grads_reproduces_get_sde_and_commutes(name):
    from pssgp.kernels.sde_grads import leaf_parameters, sde_with_grads
    k = _kernels()[name]()
    sde, grads = sde_with_grads(k)
    ref = k.get_sde()
    assert np.allclose(sde.F, ref.F, rtol=0, atol=1e-12 * max(1.0, np.max(np.abs(ref.F))))
    assert np.allclose(sde.P0, ref.P0, rtol=0, atol=1e-12 * max(1.0, np.max(np.abs(ref.P0))))
    assert np.allclose(np.atleast_2d(sde.H), np.atleast_2d(ref.H), rtol=0, atol=1e-13)
    assert len(grads) == len(leaf_parameters(k))
    for dF, dP, dH in grads:
        # what the device's contraction of the transition matrices' adjoint relies on
        assert np.max(np.abs(sde.F @ dF - dF @ sde.F)) <= 1e-10 * max(1.0, np.max(np.abs(sde.F)) ** 2)
        assert np.allclose(dP, dP.T, atol=1e-12 * max(1.0, np.max(np.abs(dP))))


@pytest.mark.parametrize("name", list(_kernels()))
def test_adjoint_gradient_equals_difference_quotients(name):
    """d ll / d theta from the reverse sweep + the frozen-basis derivatives == Richardson differences of the likelihood of
    get_sde(theta) (its basis moving with theta: the invariance the construction rests on), and d ll / d R."""
    from pssgp.kernels.sde_grads import leaf_parameters, sde_with_grads
    k = _kernels()[name]()
    t, y = _series(90)
    R = 0.1
    sde, grads = sde_with_grads(k)
    stats = G.ll_grad_stats(sde.F, sde.P0, sde.H, R, t, y)
    # the sweep's own likelihood is the oracle's sequential filter
    assert abs(stats[0] - O.ssgp_log_likelihood(sde, t, y, R, parallel=False)) < 1e-9 * abs(stats[0])
    g = G.contract(stats, sde.H, grads)

    def ll_at():
        s = k.get_sde()
        return G.ll_only(s.F, s.P0, s.H, R, t, y)

    fd = []
    for o, a in leaf_parameters(k):
        x0 = getattr(o, a)

        def f(x, o=o, a=a):
            setattr(o, a, x)
            return ll_at()
        fd.append(_richardson(f, x0))
        setattr(o, a, x0)
    fd.append(_richardson(lambda r: G.ll_only(sde.F, sde.P0, sde.H, r, t, y), R))
    fd = np.array(fd)
    assert np.max(np.abs(g - fd) / (1e-6 + np.abs(fd))) < 2e-6, (g, fd)


@pytest.mark.parametrize("name", ["m12", "m32", "m52", "m32+m52", "m32*m52"])
def test_adjoint_gradient_equals_the_dense_gp_gradient(name):
    """Kernels whose state-space form is exact: the gradient of the dense GP's log marginal likelihood (what
    tests/test_gp_vs_kfs.py:74-78 compares with at 1e-2 / 1e-3 / 1e-1) -- here at 1e-5."""
    from pssgp.kernels.sde_grads import leaf_parameters, sde_with_grads
    k = _kernels()[name]()
    t, y = _series(60, seed=3, nan_frac=0.0)
    R = 0.1
    sde, grads = sde_with_grads(k)
    g = G.contract(G.ll_grad_stats(sde.F, sde.P0, sde.H, R, t, y), sde.H, grads)

    def dense(noise=R):
        K = k.K(t[:, None]) + noise * np.eye(t.size)
        L = np.linalg.cholesky(K)
        alpha = np.linalg.solve(L.T, np.linalg.solve(L, y))
        return float(-0.5 * y @ alpha - np.sum(np.log(np.diag(L))) - 0.5 * t.size * np.log(2 * np.pi))

    fd = []
    for o, a in leaf_parameters(k):
        x0 = getattr(o, a)

        def f(x, o=o, a=a):
            setattr(o, a, x)
            return dense()
        fd.append(_richardson(f, x0))
        setattr(o, a, x0)
    fd.append(_richardson(lambda r: dense(r), R))
    fd = np.array(fd)
    assert np.max(np.abs(g - fd) / (1e-6 + np.abs(fd))) < 1e-5, (g, fd)


def test_statistics_contract_with_any_commuting_direction():
    """The device's Abar = sum dt Fbar F^T serves every dF that commutes with F -- not only the kernels' own: a random
    polynomial in F, and directions of Pinf, H; checked against differences of the likelihood in (F, Pinf, H)."""
    from pssgp.kernels import RBF
    sde = RBF(1.0, 0.6, order=4, balancing_iter=5).get_sde()
    F, P, H = np.asarray(sde.F), np.asarray(sde.P0), np.asarray(sde.H).reshape(-1)
    t, y = _series(70, seed=5)
    st = G.ll_grad_stats(F, P, H, 0.2, t, y)
    rng = np.random.default_rng(0)
    dF = 0.3 * F + 0.1 * F @ F - 0.2 * np.eye(4)
    # a direction of (F, Pinf) that keeps Pinf stationary: scale the process noise, i.e. Pinf, alone
    dP = 0.7 * P
    dH = rng.standard_normal(4)
    eps = 1e-5

    def ll(e):
        # Pinf must stay the stationary covariance of the perturbed drift for the model Q = Pinf - F Pinf F^T to be the
        # same function: dF commutes with F but moves Pinf, so perturb them one at a time
        return G.ll_only(F, P + e * dP, H + e * dH, 0.2, t, y)
    want = (ll(eps) - ll(-eps)) / (2 * eps)
    got = st[2] @ dP @ H + st[3] @ dH
    assert abs(got - want) < 1e-6 * max(1.0, abs(want))
    # time scaling: dF = -F / l with Pinf, H fixed IS a valid direction (the lengthscale of any stationary kernel)
    got_l = np.sum(st[1] * (-F / 0.6))
    want_l = (G.ll_only(F * (0.6 / (0.6 + eps)), P, H, 0.2, t, y) - G.ll_only(F * (0.6 / (0.6 - eps)), P, H, 0.2, t, y)) / (2 * eps)
    assert abs(got_l - want_l) < 1e-5 * max(1.0, abs(want_l))
    assert np.isfinite(np.sum(st[1] * dF))


# ---- hyper-parameters of any numeric type, structure changes, reassigned data (advisor, round 3) -----------------------
def test_parameters_are_floats_whatever_they_are_assigned_as():
    from pssgp.kernels import Matern32, Periodic, SquaredExponential
    from pssgp.model import StateSpaceGP
    k = Matern32(1, 2)
    assert type(k.variance) is float and type(k.lengthscales) is float
    k.lengthscales = 3
    assert type(k.lengthscales) is float and k.lengthscales == 3.0
    k.lengthscales = np.array(0.5)
    assert type(k.lengthscales) is float and k.lengthscales == 0.5
    k.variance = np.float32(2.0)
    assert type(k.variance) is float
    p = Periodic(SquaredExponential(1, 1), period=2)
    assert type(p.period) is float and type(p.base_kernel.variance) is float
    t = np.linspace(0.1, 1, 20)
    gp = StateSpaceGP((t, np.sin(t)), k, noise_variance=1, parallel=False)
    names = [n for _, n in gp.trainable_parameters()]
    assert names == ["variance", "lengthscales", "noise_variance"]


def test_param_key_follows_every_assignment():
    from pssgp.kernels import Matern32, Matern52, RBF
    from pssgp.model import StateSpaceGP
    t = np.linspace(0.1, 1, 20)
    k = Matern32(1.0, 2.0)
    gp = StateSpaceGP((t, np.sin(t)), k, 0.1, parallel=False)
    key0 = gp._param_key()
    assert gp._param_key() == key0
    k.lengthscales = 3                      # an int: used to drop out of the key
    key1 = gp._param_key()
    assert key1 != key0
    k.lengthscales = np.array(0.5)          # a 0-d array
    assert gp._param_key() not in (key0, key1)
    # what else shapes the SDE is in the key as well: the order of an RBF, its balancing sweeps, the kernel's class
    r = RBF(1.0, 0.5, order=4, balancing_iter=5)
    gp2 = StateSpaceGP((t, np.sin(t)), r, 0.1, parallel=False)
    ka = gp2._param_key()
    r._order = 6
    kb = gp2._param_key()
    r._balancing_iter = 7
    kc = gp2._param_key()
    assert len({ka, kb, kc}) == 3
    gp2.kernel = Matern52(1.0, 0.5)
    assert gp2._param_key() not in (ka, kb, kc)
    # two models over two kernel objects with the same numbers: equal keys, but the memos are tied to the object
    assert StateSpaceGP((t, np.sin(t)), Matern32(1.0, 2.0), 0.1)._param_key() == StateSpaceGP((t, np.sin(t)), Matern32(1.0, 2.0), 0.1)._param_key()


def test_data_is_a_property_that_drops_the_memos():
    from pssgp.kernels import Matern32
    from pssgp.model import StateSpaceGP
    t = np.linspace(0.1, 1, 20)
    gp = StateSpaceGP((t, np.sin(t)), Matern32(1.0, 2.0), 0.1, parallel=False)
    gp._ll_memo = ("stale",)
    gp._series = None
    gp.data = (t, np.cos(t))            # (the values after a reassignment are checked on the GPU: tests/test_gpu_adjoint.py)
    assert gp._ll_memo is None and gp.data[1].shape == (20, 1) and np.allclose(gp.data[1][:, 0], np.cos(t))
    with pytest.raises(ValueError):
        gp.data = (t, np.stack([t, t], axis=1))
    # the stamp of the uploaded arrays sees a replaced array and an edited end
    ts, ys = gp.data
    s0 = gp._data_stamp(ts, ys)
    ys2 = ys.copy()
    assert gp._data_stamp(ts, ys2) != s0
    ys[-1, 0] += 1.0
    assert gp._data_stamp(ts, ys) != s0
