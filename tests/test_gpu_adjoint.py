"""GPU parity of the adjoint log-likelihood gradient (pgps_lti_ll_grad_*, include/pgps.h): the device's model adjoints
against the numpy reverse sweep of oracle/np_grad.py, StateSpaceGP.log_likelihood_and_grad against the dense GP's
gradient at the reference's own kernels and tolerances (tests/test_gp_vs_kfs.py:33-41,53-78) and against the batched
difference quotients / dual-number passes of rounds 1-3, on both cooperative families, at every chain geometry."""
import numpy as np
import pytest

from oracle import np_grad as G
from oracle import np_oracle as O
from tests.conftest import make_times

pytestmark = pytest.mark.gpu


def _kernels():
    from pssgp.kernels import Matern32, Matern52, RBF, Periodic, SquaredExponential
    return {
        "m32+m52": lambda: Matern32(1.3, 0.7) + Matern52(0.6, 1.1),                                              # d = 5
        "m32*m52": lambda: Matern32(1.3, 0.7) * Matern52(0.6, 1.1),                                              # d = 6
        "rbf6": lambda: RBF(1.3, 0.7, order=6, balancing_iter=5),                                                # d = 6
        "per2": lambda: Periodic(SquaredExponential(1.3, 0.9), period=1.7, order=2),                             # d = 6
        "c5": lambda: Periodic(SquaredExponential(1., 1.), period=1., order=1) * Matern32(1., 1.) + Matern52(1., 1.),   # d = 11
        "rbf15": lambda: RBF(1.3, 0.7, order=15, balancing_iter=10),                                             # d = 15
        "co2": lambda: Periodic(SquaredExponential(1.2, 0.8), period=1., order=3) * Matern32(1., 30.) + Matern32(2., 1.5),  # d = 18
        "periodic10": lambda: Periodic(SquaredExponential(1., 0.8), period=1.5, order=10),                       # d = 22
        "qp3*m52": lambda: Periodic(SquaredExponential(1., 1.), period=1., order=3) * Matern52(1., 2.),          # d = 24
        "rbf3": lambda: RBF(1.0, 0.5, order=3, balancing_iter=5),                                                # d = 3
        "per0": lambda: Periodic(SquaredExponential(1.0, 1.0), period=2.0, order=0),                             # d = 2
    }


def _series(n, seed, nan_frac=0.15):
    rng = np.random.default_rng(seed)
    t = make_times(n, seed=seed)
    y = np.sin(t) + 0.5 * np.cos(2.3 * t) + 0.3 * rng.standard_normal(n)
    if nan_frac:
        y[rng.uniform(size=n) < nan_frac] = np.nan
    return t, y


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    # (an adjoint that is exactly zero -- Abar of a one-step series: nothing precedes the first step -- comes out of the
    # device as rounding residue of Pp - Pinf, ~1e-18: an absolute floor far below every non-trivial entry)
    return float(np.max(np.abs(a - b)) / max(1e-6, float(np.max(np.abs(b)))))


def _check_stats(dev, ref, tol):
    assert abs(dev[0] - ref[0]) <= 1e-9 * abs(ref[0])
    for name, a, b in zip(("Abar", "Ubar", "Hbar", "Rbar"), dev[1:], ref[1:]):
        assert _rel(a, b) <= tol, (name, _rel(a, b))


@pytest.mark.parametrize("name", list(_kernels()))
@pytest.mark.parametrize("n", [1, 2, 37, 1300])
def test_device_adjoints_match_the_reverse_sweep(name, n):
    """[ll | Abar | Ubar | Hbar | Rbar] of pgps_lti_ll_grad_f64 == oracle/np_grad.py, with missing observations, from one
    step to a few chains."""
    from pssgp import _backend as B
    sde = _kernels()[name]().get_sde()
    t, y = _series(n, seed=3 + n)
    if n <= 2:
        y[:] = 0.3                          # (an all-missing series has no likelihood to differentiate)
    dev = B.lti_ll_grad(sde.F, sde.P0, sde.H, 0.1, t, y)
    ref = G.ll_grad_stats(sde.F, sde.P0, sde.H, 0.1, t, y)
    # (RBF order 15, Periodic: cond Pinf ~ 1e5 and undamped modes -- the sweep's own rounding is ~1e-10 there)
    _check_stats(dev, ref, 5e-8 if name in ("rbf15", "per2", "periodic10", "qp3*m52", "per0") else 1e-9)


@pytest.mark.parametrize("name", ["rbf6", "c5", "rbf15", "co2"])
@pytest.mark.parametrize("family", [0, 2])
def test_both_families_and_every_chain_length(name, family):
    """The row-cooperative kernels (d <= 16) and the wave-cooperative ones (any d; forced with pgps_set_family(2)) give
    the same adjoints whatever the chain length (pgps_set_chunk): ragged last chains, one chain, one step per chain."""
    from pssgp import _backend as B
    sde = _kernels()[name]().get_sde()
    t, y = _series(777, seed=11)
    ref = G.ll_grad_stats(sde.F, sde.P0, sde.H, 0.1, t, y)
    ctx = B.get_context()
    ctx.set_family(family)
    try:
        for chunk in (0, 5, 16, 64, 1000):
            ctx.set_chunk(chunk)
            dev = B.lti_ll_grad(sde.F, sde.P0, sde.H, 0.1, t, y)
            _check_stats(dev, ref, 5e-8 if name == "rbf15" else 1e-9)
    finally:
        ctx.set_chunk(0)
        ctx.set_family(0)


def _dense_gradient(kernel, t, y, R):
    """Richardson differences of the dense GP's log marginal likelihood (the reference's oracle, tests/test_gp_vs_kfs.py)."""
    from pssgp.kernels.sde_grads import leaf_parameters

    def dense(noise=R):
        K = kernel.K(t[:, None]) + noise * np.eye(t.size)
        L = np.linalg.cholesky(K)
        alpha = np.linalg.solve(L.T, np.linalg.solve(L, y))
        return float(-0.5 * y @ alpha - np.sum(np.log(np.diag(L))) - 0.5 * t.size * np.log(2 * np.pi))

    def rich(f, x0):
        h = 1e-4 * max(abs(x0), 1e-3)
        return (4 * (f(x0 + 0.5 * h) - f(x0 - 0.5 * h)) / h - (f(x0 + h) - f(x0 - h)) / (2 * h)) / 3

    g = []
    for o, a in leaf_parameters(kernel):
        x0 = getattr(o, a)

        def f(x, o=o, a=a):
            setattr(o, a, x)
            return dense()
        g.append(rich(f, x0))
        setattr(o, a, x0)
    g.append(rich(lambda r: dense(r), R))
    return dense(), np.array(g)


def test_reference_gradient_contract():
    """tests/test_gp_vs_kfs.py:24-78 on the HIP path: the reference's seven kernels, T = 200 sorted uniform times on
    [0, 1], sinu(t) + obs_noise(0.1), noise 0.1 -- log-likelihood and gradient equal the dense GP's within the
    reference's own (value, gradient) tolerances, atol = rtol: (1e-6, 1e-2) Matern / sum, (1e-2, 1e-2) RBF order 15,
    (1e-3, 1e-3) Periodic order 10, (1e-6, 1e-1) product; the exact state-space forms are held to 1e-5 on the gradient."""
    from pssgp.experiments.toy import obs_noise, sinu
    from pssgp.kernels import Matern12, Matern32, Matern52, RBF, Periodic, SquaredExponential
    from pssgp.model import StateSpaceGP
    rng = np.random.default_rng(31415926)
    t = np.sort(rng.uniform(0, 1, 200))
    y = obs_noise(sinu(t), 0.1, 17)
    m32, m52 = Matern32(variance=1., lengthscales=0.5), Matern52(variance=1., lengthscales=0.5)
    cases = [
        (Matern12(variance=1., lengthscales=0.5), 1e-6, 1e-5), (m32, 1e-6, 1e-5), (m52, 1e-6, 1e-5),
        (RBF(variance=1., lengthscales=0.5, order=15, balancing_iter=10), 1e-2, 1e-2),
        (Periodic(SquaredExponential(variance=1., lengthscales=0.5), period=0.5, order=10), 1e-3, 1e-3),
        (m32 + m52, 1e-6, 1e-5), (m32 * m52, 1e-6, 1e-5),
    ]
    for kernel, vtol, gtol in cases:
        gp = StateSpaceGP((t[:, None], y[:, None]), kernel, noise_variance=0.1, parallel=True)
        ll, g = gp.log_likelihood_and_grad()
        ll_d, g_d = _dense_gradient(kernel, t, y, 0.1)
        assert abs(float(ll) - ll_d) <= vtol + vtol * abs(ll_d), (type(kernel).__name__, ll, ll_d)
        assert np.all(np.abs(g - g_d) <= gtol + gtol * np.abs(g_d)), (type(kernel).__name__, g, g_d)


@pytest.mark.parametrize("name", ["m32+m52", "m32*m52", "rbf6", "per2", "c5", "rbf15", "co2"])
def test_model_gradient_three_ways(name):
    """StateSpaceGP.log_likelihood_and_grad: the adjoint pass (default) against the batched difference quotients and,
    for sums / products of Matern kernels, the dual-number pass of round 2; `wrt`; the resident series and the
    host-array entry point."""
    from pssgp import _backend as B
    from pssgp.kernels.sde_grads import sde_with_grads
    from pssgp.model import StateSpaceGP
    t, y = _series(900, seed=21, nan_frac=0.1)
    gp = StateSpaceGP((t[:, None], y[:, None]), _kernels()[name](), noise_variance=0.1, parallel=True)
    ll, g = gp.log_likelihood_and_grad()
    ll_a, g_a = gp.log_likelihood_and_grad(method="adjoint")
    assert float(ll) == float(ll_a) and np.array_equal(g, g_a)          # the default IS the adjoint pass; bit-reproducible
    ll_f, g_f = gp.log_likelihood_and_grad(method="differences")
    assert abs(float(ll) - ll_f) <= 1e-9 * abs(ll_f)
    # (differences of an O(1e3) likelihood with steps of 1e-3: ~1e-6 of the gradient's scale)
    # (... and the period of a quasi-periodic kernel over ~45 periods is where a difference quotient is worst: its
    # step moves the last harmonic's phase by a radian -- the adjoint is the exact one, see test_grad_host.py)
    assert np.all(np.abs(g - g_f) <= 2e-5 * max(1.0, float(np.max(np.abs(g_f)))) + 5e-4 * np.abs(g_f)), (g, g_f)
    if name in ("m32+m52", "m32*m52"):
        ll_d, g_d = gp.log_likelihood_and_grad(method="dual")
        assert abs(float(ll) - ll_d) <= 1e-10 * abs(ll_d)
        assert np.max(np.abs(g - g_d)) <= 1e-8 * max(1.0, float(np.max(np.abs(g_d))))
    assert abs(float(ll) - float(gp.maximum_log_likelihood_objective())) <= 1e-10 * abs(float(ll))
    # only some directions
    _, g_w = gp.log_likelihood_and_grad(wrt=[0, len(g) - 1])
    assert g_w[0] == g[0] and g_w[-1] == g[-1] and np.all(g_w[1:-1] == 0.0)
    # the host-array entry point gives what the resident series gives
    sde, grads = sde_with_grads(gp.kernel)
    st = B.lti_ll_grad(sde.F, sde.P0, sde.H, 0.1, t, y)
    assert np.max(np.abs(B.contract_grad_stats(st, sde.H, grads) - g)) <= 1e-9 * max(1.0, float(np.max(np.abs(g))))


def test_long_series_against_differences():
    """c5's kernel (d = 11) at 2^17 steps and the CO2 kernel (d = 18) at 2^14: the adjoint pass against difference
    quotients of the device's own likelihood (the numpy sweep would take minutes there), and against itself at another
    chain length."""
    from pssgp import _backend as B
    from pssgp.kernels.sde_grads import sde_with_grads
    for name, n in (("c5", 1 << 17), ("co2", 1 << 14)):
        k = _kernels()[name]()
        sde, grads = sde_with_grads(k)
        t, y = _series(n, seed=5, nan_frac=0.05)
        st = B.lti_ll_grad(sde.F, sde.P0, sde.H, 0.1, t, y)
        g = B.contract_grad_stats(st, sde.H, grads)
        assert abs(st[0] - B.lti_ll(sde.F, sde.P0, sde.H, 0.1, t, y)) <= 1e-10 * abs(st[0])
        # the lengthscale of the last Matern part and the noise, by central differences of pgps_lti_ll_f64
        F, P0, H = np.asarray(sde.F), np.asarray(sde.P0), np.asarray(sde.H)
        for idx, (dF, dP, dH) in ((len(grads) - 1, grads[-1]),):
            e = 1e-6
            up = B.lti_ll(F + e * dF, P0 + e * dP, H + e * dH, 0.1, t, y)
            dn = B.lti_ll(F - e * dF, P0 - e * dP, H - e * dH, 0.1, t, y)
            assert abs((up - dn) / (2 * e) - g[idx]) <= 1e-4 * max(1.0, abs(g[idx])), (name, (up - dn) / (2 * e), g[idx])
        e = 1e-6
        dR = (B.lti_ll(F, P0, H, 0.1 + e, t, y) - B.lti_ll(F, P0, H, 0.1 - e, t, y)) / (2 * e)
        assert abs(dR - g[-1]) <= 1e-4 * max(1.0, abs(g[-1])), (name, dR, g[-1])
        ctx = B.get_context()
        ctx.set_chunk(48)
        try:
            st2 = B.lti_ll_grad(sde.F, sde.P0, sde.H, 0.1, t, y)
        finally:
            ctx.set_chunk(0)
        for a, b in zip(st[1:], st2[1:]):
            assert _rel(a, b) <= 1e-8


def test_parameters_of_any_type_and_reassigned_data_reach_the_device():
    """Advisor, round 3: a hyper-parameter assigned as an int or a 0-d array must change the evaluation (the memo keys
    listed floats only); `model.data = ...` must re-upload the series and drop the memoised likelihood."""
    from pssgp.kernels import Matern32, RBF
    from pssgp.model import StateSpaceGP
    t, y = _series(400, seed=2, nan_frac=0.0)
    for make in (lambda: Matern32(1.0, 0.5), lambda: RBF(1.0, 0.5, order=4, balancing_iter=5)):
        k = make()
        gp = StateSpaceGP((t[:, None], y[:, None]), k, noise_variance=0.1, parallel=True)
        seen = []
        for value in (2, np.array(3.0), np.float32(0.25), 0.5):
            k.lengthscales = value
            ll = float(gp.maximum_log_likelihood_objective())
            want = O.ssgp_log_likelihood(k.get_sde(), t, y, 0.1, parallel=False)
            assert abs(ll - want) <= 1e-8 * abs(want), (value, ll, want)
            _, g = gp.log_likelihood_and_grad()
            assert np.all(np.isfinite(g))
            seen.append(ll)
        assert len(set(seen)) == len(seen)
        # new data through the property: the next evaluation sees it (and predict_f's memo of the likelihood is gone)
        y2 = np.cos(t)
        gp.predict_f(t[:50, None] + 1e-3)
        gp.data = (t[:, None], y2[:, None])
        ll2 = float(gp.maximum_log_likelihood_objective())
        want2 = O.ssgp_log_likelihood(k.get_sde(), t, y2, 0.1, parallel=False)
        assert abs(ll2 - want2) <= 1e-8 * abs(want2)
        # ... and an array replaced behind the property's back (same shapes) is noticed by its stamp
        gp._data = (t[:, None].copy(), (2.0 * y2)[:, None])
        ll3 = float(gp.maximum_log_likelihood_objective())
        want3 = O.ssgp_log_likelihood(k.get_sde(), t, 2.0 * y2, 0.1, parallel=False)
        assert abs(ll3 - want3) <= 1e-8 * abs(want3)


def test_gradient_ascent_on_the_adjoint_gradient():
    """A few steps of gradient ascent in log-parameters with the adjoint gradient raise the likelihood of an RBF model
    monotonically (what the reference's MAP / HMC drivers consume the gradient for: sunspot/map.py:74-82)."""
    from pssgp.kernels import RBF
    from pssgp.model import StateSpaceGP
    t, y = _series(600, seed=9, nan_frac=0.0)
    k = RBF(0.5, 2.0, order=6, balancing_iter=5)
    gp = StateSpaceGP((t[:, None], y[:, None]), k, noise_variance=0.5, parallel=True)
    params = gp.trainable_parameters()
    first, last = None, -np.inf
    for _ in range(12):
        ll, g = gp.log_likelihood_and_grad()
        assert float(ll) >= last - 1e-9
        last = float(ll)
        first = last if first is None else first
        x = np.array([getattr(o, n) for o, n in params])
        step = 0.05 * (g * x) / max(1.0, float(np.max(np.abs(g * x))))
        for (o, n), v in zip(params, x * np.exp(step)):
            setattr(o, n, float(v))
    assert last > first + 1.0


def test_rbf_gradient_through_the_scaled_realisation():
    """An optimiser's loop over a single RBF kernel: after the first evaluation the model and its derivatives are written
    down from the reference realisation by time / variance scaling (no get_sde per step) -- the same gradient as the full
    derivation gives through the oracle's reverse sweep, also after the lengthscale has left the 25 % window."""
    from pssgp.kernels import RBF
    from pssgp.kernels.sde_grads import sde_with_grads
    from pssgp.model import StateSpaceGP
    t, y = _series(500, seed=4, nan_frac=0.1)
    k = RBF(1.1, 0.9, order=6, balancing_iter=5)
    gp = StateSpaceGP((t[:, None], y[:, None]), k, noise_variance=0.2, parallel=True)
    for ell, s2 in ((0.9, 1.1), (0.93, 1.4), (1.05, 0.8), (1.6, 0.8), (0.5, 2.0)):
        k.lengthscales, k.variance = ell, s2
        ll, g = gp.log_likelihood_and_grad()
        sde, grads = sde_with_grads(k)
        ref = G.ll_grad_stats(sde.F, sde.P0, sde.H, 0.2, t, y)
        assert abs(float(ll) - ref[0]) <= 1e-9 * abs(ref[0])
        assert np.max(np.abs(g - G.contract(ref, sde.H, grads))) <= 1e-7 * max(1.0, float(np.max(np.abs(g))))


def test_matern_family_automatic_choice_between_duals_and_the_adjoint_pass():
    """Which pass `log_likelihood_and_grad()` takes for a single Matern kernel (measured crossovers, tools/grad_methods.py):
    Matern-5/2 the adjoint pass of the fused path at every length, Matern-3/2 above the one-launch length, Matern-1/2 and
    short Matern-3/2 series the dual numbers -- and the same numbers either way (1e-9), also against the oracle's reverse
    sweep on the kernel's own get_sde(), at a new setting every call, for a subset of the directions, and through the
    general-LTI kernels when the series is not resident on the device (unsorted times)."""
    from pssgp.kernels import Matern12, Matern32, Matern52
    from pssgp.kernels.sde_grads import sde_with_grads
    from pssgp.model import StateSpaceGP
    t, y = _series(6000, seed=8, nan_frac=0.1)

    def spy(gp):
        calls = {"fused": 0, "lti": 0}
        f, l = gp._fused_adjoint_ll_and_grad, gp._adjoint_ll_and_grad
        gp._fused_adjoint_ll_and_grad = lambda *a, **kw: (calls.__setitem__("fused", calls["fused"] + 1), f(*a, **kw))[1]
        gp._adjoint_ll_and_grad = lambda *a, **kw: (calls.__setitem__("lti", calls["lti"] + 1), l(*a, **kw))[1]
        return calls

    for cls, n, want in ((Matern52, 6000, 1), (Matern52, 400, 1), (Matern32, 6000, 1), (Matern32, 1500, 0), (Matern12, 6000, 0)):
        k = cls(1.2, 0.6)
        gp = StateSpaceGP((t[:n, None], y[:n, None]), k, noise_variance=0.15, parallel=True)
        gp.maximum_log_likelihood_objective()       # (the series becomes resident with the second evaluation of a model)
        calls = spy(gp)
        for ell, s2 in ((0.6, 1.2), (0.9, 0.7), (0.35, 2.0)):
            k.lengthscales, k.variance = ell, s2
            ll, g = gp.log_likelihood_and_grad()
            lld, gd = gp.log_likelihood_and_grad(method="dual")
            assert abs(float(ll) - float(lld)) <= 1e-10 * abs(float(lld))
            assert np.max(np.abs(np.asarray(g) - np.asarray(gd))) <= 1e-9 * max(1.0, float(np.max(np.abs(gd))))
            sde, grads = sde_with_grads(k)
            ref = G.ll_grad_stats(sde.F, sde.P0, sde.H, 0.15, t[:n], y[:n])
            assert np.max(np.abs(np.asarray(g) - G.contract(ref, sde.H, grads))) <= 1e-7 * max(1.0, float(np.max(np.abs(g))))
        assert calls["fused"] == 3 * want and calls["lti"] == 0, (cls.__name__, n, calls)
        _, g1 = gp.log_likelihood_and_grad(wrt=[1])
        _, gall = gp.log_likelihood_and_grad()
        assert g1[1] == gall[1] and g1[0] == 0.0
    # a series the device does not keep (times not sorted): Matern-5/2 above 2048 points goes through the general-LTI kernels
    perm = np.random.default_rng(0).permutation(6000)
    k = Matern52(1.2, 0.6)
    gp = StateSpaceGP((t[perm][:, None], y[perm][:, None]), k, noise_variance=0.15, parallel=True)
    assert gp._device_series(force=True) is None


@pytest.mark.parametrize("kname", ["m12", "m32", "m52"])
@pytest.mark.parametrize("n,chunk", [(1, 0), (2, 0), (37, 0), (700, 0), (1300, 0), (2048, 0), (2049, 0), (5000, 0), (70001, 0), (5000, 3), (3000, 1)])
def test_fused_path_adjoints_match_the_reverse_sweep(kname, n, chunk):
    """The adjoint pass of the fused (Matern-family) path, csrc/pgps_gpadj.hip.h: the device's [ll | Abar | Ubar | Hbar |
    Rbar] against the numpy reverse sweep of oracle/np_grad.py on the kernel's own SDE -- one launch (up to 2048 steps) and
    three, one workgroup and many, every chunk length, 15 % of the observations missing."""
    from pssgp import _backend as B
    from pssgp.kernels import Matern12, Matern32, Matern52
    from pssgp.model import StateSpaceGP
    k = {"m12": Matern12(1.3, 0.7), "m32": Matern32(1.3, 0.7), "m52": Matern52(1.3, 0.7)}[kname]
    t, y = _series(n, seed=11 + n, nan_frac=0.15 if n > 2 else 0.0)
    gp = StateSpaceGP((t[:, None], y[:, None]), k, noise_variance=0.2, parallel=True)
    ctx = B.get_context()
    ctx.set_chunk(chunk)
    try:
        fused = gp._device_forms()[0]
        assert fused is not None
        ser = gp._device_series(force=True)
        assert ser is not None and ser.has_gp_adj
        dev = ser.gp_ll_grad_adj(gp._packed_fused(fused), 0.2)
    finally:
        ctx.set_chunk(0)
    sde = k.get_sde()
    ref = G.ll_grad_stats(sde.F, sde.P0, sde.H, 0.2, t, y)
    _check_stats(dev, ref, 1e-9)


@pytest.mark.parametrize("kname", ["m12", "m32", "m52"])
def test_fused_path_adjoint_gradient_equals_the_dual_number_gradient(kname):
    from pssgp.kernels import Matern12, Matern32, Matern52
    from pssgp.model import StateSpaceGP
    for n in (300, 6000):
        t, y = _series(n, seed=5 + n, nan_frac=0.1)
        k = {"m12": Matern12, "m32": Matern32, "m52": Matern52}[kname](0.9, 0.45)
        gp = StateSpaceGP((t[:, None], y[:, None]), k, noise_variance=0.12, parallel=True)
        gp.maximum_log_likelihood_objective()       # (the series becomes resident with the second evaluation of a model)
        for ell, s2 in ((0.45, 0.9), (1.1, 2.2)):
            k.lengthscales, k.variance = ell, s2
            lla, ga = gp.log_likelihood_and_grad(method="adjoint")
            lld, gd = gp.log_likelihood_and_grad(method="dual")
            assert abs(float(lla) - float(lld)) <= 1e-10 * abs(float(lld))
            assert np.max(np.abs(np.asarray(ga) - np.asarray(gd))) <= 1e-9 * max(1.0, float(np.max(np.abs(gd))))
        _, g1 = gp.log_likelihood_and_grad(wrt=[0], method="adjoint")
        assert g1[0] == ga[0] and g1[1] == 0.0 and g1[2] == 0.0


def test_device_pointer_entry_points_of_both_adjoint_passes():
    """pgps_gp_ll_grad_adj_dev_f64 and pgps_lti_ll_grad_dev_f64 (include/pgps.h): series and results on the device,
    asynchronous on the context's stream -- the same statistics as the resident-series calls and the oracle's sweep."""
    import ctypes
    from pssgp import _backend as B
    from pssgp.kernels import Matern52
    from tests.test_segments import _Dev
    L, I, D = ctypes.c_long, ctypes.c_int, ctypes.c_double
    ctx = B.get_context()
    n = 3000
    t, y = _series(n, seed=21, nan_frac=0.1)
    k = Matern52(1.1, 0.8)
    sde = k.get_sde()
    d = sde.F.shape[0]
    nst = 2 + d * d + 2 * d
    ref = G.ll_grad_stats(sde.F, sde.P0, sde.H, 0.25, t, y)
    ts, ys, out = _Dev(ctx, t), _Dev(ctx, y), _Dev(ctx, shape=(nst,))
    lam, N1, N2 = B.nilpotent_form(sde.F)
    c = lambda a: np.ascontiguousarray(a, np.float64)
    N1c, N2c, Pc, Hc = c(N1), c(N2), c(sde.P0), c(np.asarray(sde.H).reshape(-1))
    pp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    ctx.call("pgps_gp_ll_grad_adj_dev_f64", L(n), I(d), D(float(lam)), pp(N1c), pp(N2c), pp(Pc), pp(Hc), D(0.25), ts.p, D(0.0), ys.p, out.p)
    ctx.synchronize()
    _check_stats(B.split_grad_stats(out.get(), d), ref, 1e-9)
    Fc = c(sde.F)
    out.put(np.zeros(nst))
    ctx.call("pgps_lti_ll_grad_dev_f64", L(n), I(d), pp(Fc), pp(Pc), pp(Hc), D(0.25), ts.p, ys.p, D(0.0), out.p)
    ctx.synchronize()
    _check_stats(B.split_grad_stats(out.get(), d), ref, 1e-9)
