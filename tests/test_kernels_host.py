"""Host-side kernel layer (numpy): known-answer constants from the reference's own tests and
structural checks.  No GPU."""
import numpy as np
import numpy.testing as npt
import pytest

from pssgp import config as pssgp_config
from pssgp.kernels import Matern12, Matern32, Matern52, RBF, Periodic, SquaredExponential
from pssgp.kernels.base import SDEProduct, SDESum
from pssgp.kernels.math_utils import balance_ss, solve_lyap_vec
from pssgp.kernels.periodic import _get_offline_coeffs
from pssgp.model import _merge_sorted


def test_rbf_sde_coefficients():
    """/root/reference/tests/test_rbf.py:26-47 (MATLAB-derived constants, 8 decimals)."""
    F_expected = np.array([[0, 14.520676967550859, 0],
                           [0, 0, 32.857489440296360],
                           [-14.5210953665873, -29.4746060478111, -50.3678777987092]])
    Pinf_expected = np.array([[1.04502531824891, -1.41636387123970e-17, -0.301281550265743],
                              [-1.41636387123970e-17, 0.681741999944955, -1.70331397804495e-17],
                              [-0.301281550265743, -1.70331397804495e-17, 0.611552410634913]])
    Pinf, F, L, H, Q = RBF(variance=1., lengthscales=0.1, order=3, balancing_iter=5).get_sde()
    npt.assert_array_almost_equal(F, F_expected, decimal=8)
    npt.assert_array_almost_equal(L, np.array([0., 0., 1.]).reshape(3, 1), decimal=8)
    npt.assert_array_almost_equal(H, np.array([1., 0., 0.]).reshape(1, 3), decimal=8)
    npt.assert_array_almost_equal(Q, 52.8553179255264, decimal=8)
    npt.assert_array_almost_equal(Pinf, Pinf_expected, decimal=8)


def test_rbf_balancing_convergence():
    """/root/reference/tests/test_rbf.py:49-57."""
    a = RBF(variance=1., lengthscales=0.1, order=3, balancing_iter=5).get_sde()
    b = RBF(variance=1., lengthscales=0.1, order=3, balancing_iter=15).get_sde()
    for x, y in zip(a, b):
        npt.assert_array_almost_equal(x, y, decimal=3)


def test_periodic_offline_coeffs():
    """/root/reference/tests/test_periodic.py:29-40."""
    b, K, div_facto_K = _get_offline_coeffs(2)
    npt.assert_almost_equal(b, np.array([[1, 0, 0], [0, 2, 0], [2, 0, 2]]), decimal=8)
    npt.assert_almost_equal(K, np.array([[0, 0, 0], [1, 1, 1], [2, 2, 2]]), decimal=8)
    npt.assert_almost_equal(div_facto_K, np.array([[1, 1, 1], [1, 1, 1], [0.5, 0.5, 0.5]]), decimal=8)


def test_periodic_sde_coeff():
    """/root/reference/tests/test_periodic.py:42-61 (its Pinf assertion is vacuous at 7 decimals;
    here Pinf is additionally pinned by its defining property: k(0) = sum of the diagonal/2)."""
    F_expected = np.zeros((6, 6))
    F_expected[2, 3] = -6.283185307179586
    F_expected[4, 5] = -12.5663706143592
    F_expected = F_expected - F_expected.T
    cov = Periodic(SquaredExponential(variance=1., lengthscales=0.1), period=1., order=2)
    Pinf, F, L, H, Q = cov.get_sde()
    npt.assert_almost_equal(F, F_expected)
    npt.assert_almost_equal(L, np.eye(6))
    npt.assert_almost_equal(H, np.array([[1, 0, 1, 0, 1, 0]]))
    npt.assert_almost_equal(Q, np.zeros((6, 6)))
    npt.assert_almost_equal(Pinf, np.diag([1.20739740482544e-19, 1.20739740482544e-19, 9.64374923981979e-21,
                                           9.64374923981979e-21, 1.20546865497747e-19, 1.20546865497747e-19]))
    # with enough harmonics H Pinf H^T -> k(0) = variance
    P20, _, _, H20, _ = Periodic(SquaredExponential(1.3, 0.8), period=0.7, order=20).get_sde()
    assert abs((H20 @ P20 @ H20.T).item() - 1.3) < 1e-10


@pytest.mark.parametrize("make,dim", [
    (lambda: Matern12(1.2, 0.7), 1), (lambda: Matern32(1.2, 0.7), 2), (lambda: Matern52(1.2, 0.7), 3),
    (lambda: RBF(1.2, 0.7, order=6), 6), (lambda: Periodic(SquaredExponential(1., 1.), 1., order=3), 8),
    (lambda: Matern32(1., 1.) + Matern52(1., 1.), 5), (lambda: Matern32(1., 1.) * Matern52(1., 1.), 6),
    (lambda: Periodic(SquaredExponential(1., 1.), period=1., order=1) * Matern32(1., 1.) + Matern52(1., 1.), 11),
])
def test_sde_is_stationary_and_shaped(make, dim):
    """Every kernel's P0 solves F P + P F^T + L Q L^T = 0 (what the GPU discretisation relies on),
    k(0) = H P0 H^T, and get_spec reports the state dimension."""
    k = make()
    P0, F, L, H, Q = k.get_sde()
    assert F.shape == (dim, dim) and P0.shape == (dim, dim) and H.shape == (1, dim)
    resid = F @ P0 + P0 @ F.T + L @ np.atleast_2d(Q) @ L.T
    assert np.max(np.abs(resid)) < 1e-9 * max(1.0, np.max(np.abs(F)) * np.max(np.abs(P0)))
    spec = k.get_spec(17)
    assert spec.Fs.shape == (17, dim, dim) and spec.P0.shape == (dim, dim) and spec.H.shape == (1, dim)
    k0 = float(np.asarray(k.K(np.zeros(1)))[0, 0])
    # Matern / sums / products are exact; RBF and Periodic truncate
    exact = not any(isinstance(x, (RBF, Periodic)) for x in getattr(k, "kernels", [k])) and \
        not any(isinstance(y, Periodic) for x in getattr(k, "kernels", []) for y in getattr(x, "kernels", []))
    if exact:
        assert abs((H @ P0 @ H.T).item() - k0) < 1e-9


def test_kernel_algebra_types_and_config():
    k = Matern32() + Matern52()
    assert isinstance(k, SDESum) and len(k.kernels) == 2
    k3 = k + Matern32(2., 3.)
    assert isinstance(k3, SDESum) and len(k3.kernels) == 3 and k3.get_sde().F.shape == (7, 7)
    p = Matern32() * Matern52()
    assert isinstance(p, SDEProduct) and p.get_spec(None).P0.shape == (6, 6)
    with pytest.raises(TypeError):
        SDESum([Matern32(), SquaredExponential()])
    old = pssgp_config.NUMBER_OF_BALANCING_STEPS
    try:
        pssgp_config.set_number_balancing_steps(3)
        assert pssgp_config.NUMBER_OF_BALANCING_STEPS == 3
        assert Matern52()._balancing_iter == 3
    finally:
        pssgp_config.set_number_balancing_steps(old)


def test_balance_and_lyapunov():
    rng = np.random.default_rng(0)
    F = -np.eye(4) * 3 + rng.standard_normal((4, 4)) * np.array([1e-3, 1., 1e3, 1.])[None, :]
    F = F - 5 * np.eye(4) * np.max(np.abs(F))
    L = np.zeros((4, 1)); L[-1, 0] = 1.
    H = np.zeros((1, 4)); H[0, 0] = 1.
    Fb, Lb, Hb, qb = balance_ss(F, L, H, np.array([[2.0]]), 10)
    # similarity transform: spectrum preserved; transfer function H (sI - F)^-1 L sqrt(q) preserved
    npt.assert_allclose(np.sort_complex(np.linalg.eigvals(Fb)), np.sort_complex(np.linalg.eigvals(F)), rtol=1e-8)
    s = 0.3j
    g0 = (H @ np.linalg.solve(s * np.eye(4) - F, L))[0, 0] * np.sqrt(2.0)
    g1 = (Hb @ np.linalg.solve(s * np.eye(4) - Fb, Lb))[0, 0] * np.sqrt(qb.item())
    assert abs(abs(g0) - abs(g1)) < 1e-8 * abs(g0)
    P = solve_lyap_vec(Fb, Lb, qb)
    assert np.max(np.abs(Fb @ P + P @ Fb.T + Lb @ qb @ Lb.T)) < 1e-8 * np.max(np.abs(P)) * np.max(np.abs(Fb))


def test_merge_sorted_semantics():
    """pssgp/model.py:15-55 incl. the tie rule (shorter array's point first)."""
    a = np.array([0., 1., 2., 3.])
    b = np.array([1., 3.])
    c, flag = _merge_sorted(a, b, (np.zeros(4, bool), np.ones(2, bool)))
    assert c.tolist() == [0., 1., 1., 2., 3., 3.]
    assert flag.tolist() == [False, True, False, False, True, False]
    c2, flag2 = _merge_sorted(b, a, (np.ones(2, bool), np.zeros(4, bool)))
    assert c2.tolist() == c.tolist() and flag2.tolist() == flag.tolist()
    rng = np.random.default_rng(1)
    x, y = np.sort(rng.random(100)), np.sort(rng.random(37))
    c, pay = _merge_sorted(x, y, (x[:, None] * 2, y[:, None] * 3))
    assert np.all(np.diff(c) >= 0) and pay.shape == (137, 1)
    empty = np.array([])
    c, = _merge_sorted(x, empty)
    assert np.array_equal(c, x)


def test_nilpotent_form_detection_host():
    """Which kernels qualify for the fused-discretisation GPU path (F = -lam I + N, N nilpotent, d <= 3)."""
    from pssgp import _backend as B
    for k in (Matern12(2., 0.3), Matern32(1., 1.), Matern52(0.5, 2.)):
        F = np.asarray(k.get_sde().F)
        lam, N1, N2 = B.nilpotent_form(F)
        d = F.shape[0]
        assert np.allclose(-lam * np.eye(d) + N1, F) and np.allclose(N2, 0.5 * N1 @ N1)
        assert np.max(np.abs(np.linalg.matrix_power(N1, d))) < 1e-9 * max(1.0, np.max(np.abs(F))) ** d
        # closed form == expm
        import scipy.linalg as sla
        dt = 0.37
        closed = np.exp(-lam * dt) * (np.eye(d) + dt * N1 + dt * dt * N2)
        assert np.max(np.abs(closed - sla.expm(dt * F))) < 1e-12
    assert B.nilpotent_form(RBF(1., 1., order=3).get_sde().F) is None
    assert B.nilpotent_form(Periodic(SquaredExponential(1., 1.), 1., order=1).get_sde().F) is None
    assert B.nilpotent_form((Matern32() + Matern52()).get_sde().F) is None


def test_native_balancing_sweep_equals_the_numpy_loop(monkeypatch):
    """pgps_host_balance_f64 (libpgps host code, used by get_sde when the library is built) against the numpy
    restatement of the reference's numba loop (pssgp/kernels/math_utils.py:10-29), including the 0/0 of an isolated
    state."""
    from pssgp.kernels import math_utils as M
    rng = np.random.default_rng(3)
    for d in (2, 5, 11, 18):
        F = rng.standard_normal((d, d)) * np.exp(rng.uniform(-3, 3, (d, d)))
        native = M._native_balancing_diagonal(F, 7)
        assert native is not None, "libpgps.so not built"
        monkeypatch.setattr(M, "_native_balancing_diagonal", lambda *a: None)
        ref = M._balancing_diagonal(F, 7)
        monkeypatch.undo()
        assert np.max(np.abs(native / ref - 1.0)) < 1e-13
    F = np.zeros((4, 4))
    F[:2, :2] = [[-1., 2.], [0.5, -3.]]
    with np.errstate(all="ignore"):
        out = M._native_balancing_diagonal(F, 3)
        monkeypatch.setattr(M, "_native_balancing_diagonal", lambda *a: None)
        ref = M._balancing_diagonal(F, 3)
    # 0/0 at the isolated states, and 0 * NaN spreads it through their (zero) rows and columns: NaN throughout, as the
    # reference's loop gives (Periodic + Matern52 as a direct sum)
    assert np.all(np.isnan(out)) and np.all(np.isnan(ref))


def test_toy_signals():
    """The reference's test signals (pssgp/toymodels/data_funcs.py:10-97; its tests import sinu / obs_noise from there):
    here they live with the experiment drivers, pssgp/experiments/toy.py."""
    from pssgp.experiments.toy import comp_sinu, obs_noise, rect, sinu
    t = np.linspace(0.0, 1.0, 11)
    for f in (sinu, comp_sinu, rect):
        assert f(t).shape == t.shape and np.all(np.isfinite(f(t)))
    y = obs_noise(sinu(t), 0.1, 7)
    assert y.shape == t.shape and np.all(y == obs_noise(sinu(t), 0.1, 7))


def test_get_sde_is_memoised_per_kernel_object_until_a_parameter_moves():
    """One kernel object shared by many short-lived models (the reference's speed protocol) builds its SDE once; an
    assignment to any kernel's attribute, another number of balancing sweeps or another default float rebuilds it."""
    from pssgp import config
    from pssgp.kernels import Matern32, Matern52, Periodic, RBF, SquaredExponential
    k = RBF(1.0, 0.5, order=6, balancing_iter=10)
    a = k.get_sde()
    assert k.get_sde() is a
    k.lengthscales = 0.7
    b = k.get_sde()
    assert b is not a and np.max(np.abs(np.asarray(b.F) - np.asarray(a.F))) > 0
    k.lengthscales = 0.5
    c = k.get_sde()
    assert c is not a and np.array_equal(np.asarray(c.F), np.asarray(a.F)) and np.array_equal(np.asarray(c.P0), np.asarray(a.P0))
    q = Periodic(SquaredExponential(1.0, 1.0), period=1.0, order=2) * Matern32(1.0, 1.0) + Matern52(1.0, 1.0)
    s1 = q.get_sde()
    assert q.get_sde() is s1
    q.kernels[1].variance = 2.0                  # a leaf two levels down
    s2 = q.get_sde()
    assert s2 is not s1 and abs(np.max(np.abs(np.asarray(s2.P0) - np.asarray(s1.P0))) - 1.0) < 1e-9
    m = Matern52(1.0, 1.0)
    s3 = m.get_sde()
    old = config.NUMBER_OF_BALANCING_STEPS
    try:
        config.set_number_balancing_steps(old + 3)
        assert m.get_sde() is not s3
    finally:
        config.set_number_balancing_steps(old)


def test_get_sde_memo_sees_a_part_swapped_in_place():
    from pssgp.kernels import Matern12, Matern32, Matern52
    q = Matern32(1.0, 1.0) + Matern52(1.0, 1.0)
    a = q.get_sde()
    assert q.get_sde() is a
    q.kernels[0] = Matern12(1.0, 1.0)           # no attribute of any kernel is assigned
    b = q.get_sde()
    assert b is not a and np.asarray(b.F).shape == (4, 4) and np.asarray(a.F).shape == (5, 5)
