"""GPU parity of the wave-cooperative family (csrc/pgps_wc.hip: operands in LDS, 64 lanes share every
matrix operation; state dims up to 32) against the CPU oracle.  It is the automatic choice for d > 6;
here it is also forced at small d, where the lane-chunk family gives a second opinion."""
import numpy as np
import pytest

from oracle import np_oracle as O
from tests.conftest import make_times, relerr, sample_series

pytestmark = pytest.mark.gpu
TOL64 = 1e-9
TOL32 = 1e-3


def _big_kernels():
    from pssgp.kernels import Matern32, Matern52, RBF, Periodic, SquaredExponential
    return {
        "rbf15": lambda: RBF(variance=1., lengthscales=0.5, order=15, balancing_iter=10),                     # d = 15
        "periodic10": lambda: Periodic(SquaredExponential(1., 0.5), period=0.5, order=10),                    # d = 22
        "c5_qp_m52": lambda: Periodic(SquaredExponential(1., 1.), period=1., order=1) * Matern32(1., 1.) +
        Matern52(1., 1.),                                                                                     # d = 11
        "rbf8": lambda: RBF(variance=1., lengthscales=0.7, order=8, balancing_iter=10),                       # d = 8
    }


def _oracle_all(ssm, y):
    fms, fPs, ll = O.kf(ssm, y, True)
    sms, sPs = O.kfs(ssm, y)
    return dict(fms=fms, fPs=fPs, sms=sms, sPs=sPs, ll=np.array([ll]))


def _gpu_all(ssm, y, dtype):
    from pssgp import _backend as B
    ssm_t = tuple(np.asarray(a, dtype=dtype) for a in ssm)
    sms, sPs, fms, fPs, ll = B.pkfs(ssm_t, np.asarray(y, dtype), return_filtered=True, return_loglikelihood=True)
    return dict(fms=fms, fPs=fPs, sms=sms, sPs=sPs, ll=np.array([float(ll)]))


def _check(got, want, tol):
    for name in want:
        e = relerr(got[name], want[name])
        assert e < tol, f"{name}: rel err {e:.3e} >= {tol}"


@pytest.fixture
def wave_family():
    from pssgp import _backend as B
    ctx = B.get_context()
    ctx.set_family(2)
    yield ctx
    ctx.set_family(0)
    ctx.set_chunk(0)


@pytest.mark.parametrize("idx", range(7))
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_forced_wavecoop_small_d(wave_family, kernel_zoo, idx, dtype):
    name, make, _, _ = kernel_zoo[idx]
    t = make_times(1100, seed=idx)
    ssm = O.get_ssm(make().get_sde(), t, 0.1)
    y = sample_series(ssm, seed=idx, nan_frac=0.2)
    _check(_gpu_all(ssm, y, dtype), _oracle_all(ssm, y), TOL64 if dtype == np.float64 else TOL32)


@pytest.mark.parametrize("n,lw", [(1, 32), (2, 32), (31, 32), (32, 32), (33, 32), (2049, 32), (2200, 7), (4097, 1),
                                  (70000, 16)])
def test_wavecoop_ragged_lengths_and_levels(wave_family, n, lw):
    """Chunk / group / multi-group boundaries of the three-level scan (4..64 chunks per group, Kogge-Stone over the
    group totals: 17 groups at n = 2049, 456 at n = 4097, 487 at n = 70000)."""
    from pssgp.kernels import Matern52
    wave_family.set_chunk(lw)
    t = make_times(n, seed=n % 97)
    ssm = O.get_ssm(Matern52(1., 1.).get_sde(), t, 0.1)
    y = sample_series(ssm, seed=3, nan_frac=0.1 if n > 4 else 0.0)
    from oracle import c_oracle as C
    cf, cP, cs, csP, cll = C.kfs(ssm, y)
    _check(_gpu_all(ssm, y, np.float64), dict(fms=cf, fPs=cP, sms=cs, sPs=csP, ll=np.array([cll])), TOL64)


@pytest.mark.parametrize("name", ["rbf8", "c5_qp_m52", "rbf15", "periodic10"])
def test_large_state_dims(name):
    """d = 8, 11, 15, 22: automatic dispatch to the wave-cooperative kernels, pkf / pkfs / discretise."""
    from pssgp import _backend as B
    from pssgp.kalman.parallel import pkf, pkfs
    k = _big_kernels()[name]()
    sde = k.get_sde()
    n = 1500
    t = make_times(n, seed=5)
    ssm = O.get_ssm(sde, t, 0.1)
    y = sample_series(ssm, seed=5, nan_frac=0.2)
    want = _oracle_all(ssm, y)
    got = _gpu_all(ssm, y, np.float64)
    # badly conditioned high-order RBF / tiny-variance periodic harmonics: compare on the model's scale
    tol = 1e-7
    _check(got, want, tol)
    fms, fPs, ll = pkf(ssm, y[:, None], return_loglikelihood=True)
    assert relerr(fms, want["fms"]) < tol and abs(float(ll) - want["ll"][0]) < tol * abs(want["ll"][0])
    sms, sPs = pkfs(ssm, y[:, None])
    assert relerr(sms, want["sms"]) < tol and relerr(sPs, want["sPs"]) < tol
    # discretisation of the same model on the GPU vs the reference's matrix-fraction formula
    gFs, gQs = B.discretise(sde.F, sde.P0, t, 0.0)
    assert np.max(np.abs(gFs - ssm[1])) < 1e-10 * max(1.0, float(np.max(np.abs(ssm[1]))))
    assert np.max(np.abs(gQs - ssm[2])) < 1e-10 * max(1.0, float(np.max(np.abs(ssm[0]))))


def test_reference_equivalence_suite_exact_kernels():
    """/root/reference/tests/test_gp_vs_kfs.py with its exact seven kernels (RBF order 15 = d 15,
    Periodic order 10 = d 22 included) through StateSpaceGP(parallel=True) on the HIP backend."""
    from pssgp.kernels import Matern12, Matern32, Matern52, RBF, Periodic, SquaredExponential
    from pssgp.model import StateSpaceGP
    m32, m52 = Matern32(variance=1., lengthscales=0.5), Matern52(variance=1., lengthscales=0.5)
    covs = [
        (Matern12(variance=1., lengthscales=0.5), ("matern12", 1., 0.5), 1e-6),
        (m32, ("matern32", 1., 0.5), 1e-6),
        (m52, ("matern52", 1., 0.5), 1e-6),
        (RBF(variance=1., lengthscales=0.5, order=15, balancing_iter=10), ("rbf", 1., 0.5), 1e-2),
        (Periodic(SquaredExponential(variance=1., lengthscales=0.5), period=0.5, order=10), ("periodic", 1., 0.5, 0.5), 1e-3),
        (m32 + m52, ("sum", [("matern32", 1., 0.5), ("matern52", 1., 0.5)]), 1e-6),
        (m32 * m52, ("prod", [("matern32", 1., 0.5), ("matern52", 1., 0.5)]), 1e-6),
    ]
    rng = np.random.RandomState(31415926)
    T, K = 200, 50
    t = np.sort(rng.rand(T))
    f = np.sin(np.pi * t) + np.sin(2 * np.pi * t) + np.cos(3 * np.pi * t)
    y = f + np.sqrt(0.1) * rng.normal(f, np.sqrt(0.1), (T,))
    query = np.sort(rng.rand(K, 1), 0)
    for cov, spec, val_tol in covs:
        ll_gp, mean_gp, var_gp = O.dense_gp(spec, t, y, 0.1, query)
        for parallel in (False, True):
            model = StateSpaceGP(data=(t[:, None], y[:, None]), kernel=cov, noise_variance=0.1, parallel=parallel,
                                 max_parallel=T + K)
            np.testing.assert_allclose(float(model.maximum_log_likelihood_objective()), ll_gp, atol=val_tol,
                                       rtol=val_tol)
            mean_ss, var_ss = model.predict_f(query)
            np.testing.assert_allclose(mean_ss[:, 0], mean_gp, atol=val_tol, rtol=val_tol)
            np.testing.assert_allclose(var_ss[:, 0], var_gp, atol=val_tol, rtol=val_tol)


def test_level3_kogge_stone_equals_serial_walk(monkeypatch):
    """Level 3 of the scan -- Kogge-Stone over the group totals -- against the single-wave serial walk it replaced
    (PGPS_WC_SERIAL3=1, read when a context is created), at d = 18 (the CO2 kernel) with 375 groups."""
    from pssgp import _backend as B
    from pssgp.kernels import Matern32, Periodic, SquaredExponential
    k = Periodic(SquaredExponential(1., 1.), period=1., order=3) * Matern32(0.5, 5.) + Matern32(1., 2.)
    t = make_times(6000, seed=12)
    ssm = O.get_ssm(k.get_sde(), t, 0.1)
    y = sample_series(ssm, seed=12, nan_frac=0.15)
    default = B.get_context()
    monkeypatch.setenv("PGPS_WC_SERIAL3", "1")
    serial = B.Context(0)
    monkeypatch.delenv("PGPS_WC_SERIAL3")
    res = []
    try:
        for ctx in (default, serial):
            monkeypatch.setitem(B._contexts, 0, ctx)
            ctx.set_chunk(4)                                  # 1500 chunks, 4 per group
            res.append(_gpu_all(ssm, y, np.float64))
    finally:
        monkeypatch.setitem(B._contexts, 0, default)
        default.set_chunk(0)
        serial.close()
    _check(res[0], res[1], 1e-10)
    _check(res[0], _oracle_all(ssm, y), 1e-7)
