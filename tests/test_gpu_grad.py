"""GPU tests of the log-likelihood gradient (pgps_gp_ll_grad_f64: forward-mode dual numbers carried
through the parallel filter) -- the replacement for TensorFlow autodiff through the scan that the
reference checks in tests/test_gp_vs_kfs.py:53-78 (there at 1e-2 against the dense GP's gradient)."""
import numpy as np
import pytest

from oracle import np_oracle as O
from oracle import c_oracle as C
from tests.conftest import relerr
from tests.test_cpu_math import fd_grad, grad_case

pytestmark = pytest.mark.gpu


def _model(cls, theta, t, y):
    from pssgp.model import StateSpaceGP
    return StateSpaceGP((t[:, None], y[:, None]), cls(theta[0], theta[1]), noise_variance=theta[2], parallel=True)


@pytest.mark.parametrize("kname", ["m12", "m32", "m52"])
def test_gradient_equals_dense_gp_gradient(kname):
    cls, spec_name, t, y = grad_case(kname, 200, 11)
    theta = np.array([1.3, 0.7, 0.2])
    ll, g = _model(cls, theta, t, y).log_likelihood_and_grad()
    dense = lambda th: O.dense_gp((spec_name, th[0], th[1]), t, y, th[2])
    assert abs(ll - dense(theta)) < 1e-8 * abs(dense(theta))
    assert relerr(g, fd_grad(dense, theta)) < 1e-6


@pytest.mark.parametrize("kname,n", [("m12", 70001), ("m32", 100000), ("m52", 50003)])
def test_gradient_long_series_with_missing(kname, n):
    """Many blocks (spine fold + block scans on duals) and missing observations: against 4th-order
    finite differences of the C sequential oracle's log-likelihood, and the value against the
    ordinary fused log-likelihood path."""
    from pssgp import _backend as B
    cls, _, t, y = grad_case(kname, n, 17, nan_frac=0.15)
    theta = np.array([0.9, 0.8, 0.3])
    m = _model(cls, theta, t, y)
    ll, g = m.log_likelihood_and_grad()

    def seq(th):
        sde = cls(th[0], th[1]).get_sde()
        Fs, Qs = B.discretise(sde.F, sde.P0, t, 0.0)
        return C.kfs((sde.P0, Fs, Qs, np.asarray(sde.H).reshape(1, -1), np.array([[th[2]]])), y)[4]

    assert abs(ll - seq(theta)) < 1e-9 * abs(seq(theta))
    assert abs(ll - float(m.maximum_log_likelihood_objective())) < 1e-10 * abs(ll)
    assert relerr(g, fd_grad(seq, theta, rel=1e-4)) < 1e-5


def test_gradient_chunk_geometry_invariance():
    """The same gradient (to rounding) whatever the steps-per-lane geometry, i.e. whichever way the
    scan brackets the dual elements."""
    from pssgp import _backend as B
    cls, _, t, y = grad_case("m32", 30011, 23, nan_frac=0.05)
    m = _model(cls, np.array([1.1, 0.5, 0.15]), t, y)
    ctx = B.get_context()
    res = []
    try:
        for lc in (1, 4, 16, 64):
            ctx.set_chunk(lc)
            res.append(np.concatenate([[m.log_likelihood_and_grad()[0]], m.log_likelihood_and_grad()[1]]))
    finally:
        ctx.set_chunk(0)
    for r in res[1:]:
        assert relerr(r, res[0]) < 1e-10


@pytest.mark.parametrize("n", [5003, 70001])
@pytest.mark.parametrize("kname", ["m12", "m32"])
def test_gradient_one_direction_per_model_equals_all_in_one_dual(kname, n):
    """At d <= 2 short series run one derivative direction per model side by side (a Dual<1> scan tree per direction),
    long ones all directions in one Dual<3> (pgps_set_grad_pack): the same numbers either way."""
    from pssgp import _backend as B
    cls, _, t, y = grad_case(kname, n, 29, nan_frac=0.1)
    m = _model(cls, np.array([1.2, 0.6, 0.25]), t, y)
    ctx = B.get_context()
    res = []
    try:
        for limit in (0, 1 << 30):
            ctx.set_grad_pack(limit)
            ll, g = m.log_likelihood_and_grad()
            res.append(np.concatenate([[ll], g]))
    finally:
        ctx.set_grad_pack(-1)
    assert relerr(res[1], res[0]) < 1e-10


def test_gradient_descends():
    """A few steps of gradient ascent on ll from a poor start increase ll monotonically (the use
    the reference's L-BFGS / HMC drivers make of the gradient)."""
    cls, _, t, y = grad_case("m32", 20000, 29)
    theta = np.log(np.array([3.0, 3.0, 1.0]))
    lls = []
    for _ in range(6):
        ll, g = _model(cls, np.exp(theta), t, y).log_likelihood_and_grad()
        lls.append(ll)
        step = g * np.exp(theta)                       # chain rule to log-parameters
        theta = theta + 0.3 * step / max(1.0, np.linalg.norm(step))
    assert all(b > a for a, b in zip(lls, lls[1:])), lls


def test_gradient_rejects_unsupported():
    """Gradients run on the parallel (HIP) path only: the sequential model says so loudly.  (Every kernel has a
    gradient there: dual numbers for the Matern family, batched differences up to d = 16, one evaluation at a time
    above -- tests/test_gpu_lti.py.)"""
    from pssgp.kernels import RBF
    from pssgp.model import StateSpaceGP
    t = np.linspace(0.0, 1.0, 50)
    m = StateSpaceGP((t[:, None], np.sin(t)[:, None]), RBF(1.0, 1.0, order=6, balancing_iter=5), 0.1, parallel=False)
    with pytest.raises(NotImplementedError):
        m.log_likelihood_and_grad()


# ---- composite kernels: exact gradients (dual numbers through the scan, block-nilpotent drift) --------------------------
def _dense_ll_and_grad(kernel_factory, theta, t, y):
    """Dense-GP log marginal likelihood and its gradient by 6th-order central differences in fp64 (truncation ~1e-10)."""
    def ll(th):
        k, noise = kernel_factory(th)
        K = k.K(t[:, None]) + noise * np.eye(t.size)
        L = np.linalg.cholesky(K)
        a = np.linalg.solve(L, y)
        return -0.5 * a @ a - np.sum(np.log(np.diag(L))) - 0.5 * t.size * np.log(2 * np.pi)
    g = np.zeros(theta.size)
    for i in range(theta.size):
        h = 1e-3 * max(abs(theta[i]), 1e-2)
        e = np.zeros(theta.size); e[i] = h
        g[i] = (45 * (ll(theta + e) - ll(theta - e)) - 9 * (ll(theta + 2 * e) - ll(theta - 2 * e)) + (ll(theta + 3 * e) - ll(theta - 3 * e))) / (60 * h)
    return ll(theta), g


@pytest.mark.parametrize("kind", ["sum", "prod", "sum3"])
def test_composite_matern_gradient_is_exact(kind):
    """`Matern32 + Matern52` and `Matern32 * Matern52` -- the composite kernels of the reference's gradient test
    (tests/test_gp_vs_kfs.py:40-41,53-78, tolerance 1e-2 there) -- and a sum of two Matern-5/2: log_likelihood_and_grad
    runs ONE dual-number pass per parameter through the scan (pgps_gp_ll_grad_blocks_*), no finite differences of the
    likelihood; against the dense GP's gradient at 1e-6."""
    from pssgp.kernels import Matern32, Matern52
    from pssgp.model import StateSpaceGP
    from pssgp import _backend as B
    rng = np.random.default_rng(3)
    t = np.sort(rng.uniform(0.0, 4.0, 160))
    y = np.sin(2.0 * t) + 0.4 * np.cos(5.0 * t) + 0.2 * rng.standard_normal(t.size)

    def factory(th):
        if kind == "sum":
            return Matern32(th[0], th[1]) + Matern52(th[2], th[3]), th[4]
        if kind == "prod":
            return Matern32(th[0], th[1]) * Matern52(th[2], th[3]), th[4]
        return Matern52(th[0], th[1]) + Matern52(th[2], th[3]), th[4]

    theta = np.array([0.8, 0.6, 1.3, 0.9, 0.15])
    kern, noise = factory(theta)
    m = StateSpaceGP((t[:, None], y[:, None]), kern, noise_variance=noise, parallel=True)
    names = [n for _, n in m.trainable_parameters()]
    assert names == ["variance", "lengthscales", "variance", "lengthscales", "noise_variance"]
    rows, sizes = m._grad_rows_composite()
    assert rows is not None and sizes == {"sum": [2, 3], "prod": [6], "sum3": [3, 3]}[kind]
    ll, g = m.log_likelihood_and_grad(method="dual")
    ll_d, g_d = _dense_ll_and_grad(factory, theta, t, y)
    assert abs(ll - ll_d) < 1e-8 * abs(ll_d)
    assert np.max(np.abs(g - g_d)) < 1e-6 * np.max(np.abs(g_d)), (g, g_d)
    # and the device call really is the dual-number one
    calls = []
    orig = B.gp_ll_grad_blocks
    B.gp_ll_grad_blocks = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        m.log_likelihood_and_grad(method="dual")
        assert calls
        # since round 4 the default is the adjoint pass (two passes whatever the number of parameters): same numbers
        del calls[:]
        ll_a, g_a = m.log_likelihood_and_grad()
        assert not calls
    finally:
        B.gp_ll_grad_blocks = orig
    assert abs(float(ll_a) - ll_d) < 1e-8 * abs(ll_d)
    assert np.max(np.abs(g_a - g_d)) < 1e-6 * np.max(np.abs(g_d)), (g_a, g_d)
