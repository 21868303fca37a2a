"""GPU parity tests (run with `-m gpu` on an MI355X): the HIP scan, called through the C ABI
(ctypes -> libpgps.so), against the CPU oracle on the same seeded inputs.

Tolerances are the north-star's: 1e-5 relative in fp64, 1e-3 in fp32 (BASELINE.json).  In
fp64 the observed agreement is ~1e-12; the tests assert a tighter 1e-9 so regressions show.
"""
import os

import numpy as np
import pytest

from oracle import np_oracle as O
from oracle import c_oracle as C
from tests.conftest import make_times, relerr, sample_series, sample_series_fast

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
TOL64 = 1e-9
TOL32 = 1e-3
# How the tolerances are applied (tests/conftest.py `relerr`): max |got - want| / max |want| over the whole array -- a
# MAX-NORM relative error, as the reference's own comparisons are (np.testing.assert_allclose with atol = rtol on arrays of
# O(1) entries, tests/test_gp_vs_kfs.py:60-99).  Entries much smaller than the array's largest one (far off-diagonal terms of
# a smoothed covariance) are therefore not each held to 1e-3 of their own size in float32.


def _gpu():
    from pssgp import _backend
    return _backend


def _check_all(got, want, tol):
    for name in want:
        e = relerr(got[name], want[name])
        assert e < tol, f"{name}: rel err {e:.3e} >= {tol}"


def _oracle_all(ssm, y):
    fms, fPs, ll = O.kf(ssm, y, True)
    sms, sPs = O.kfs(ssm, y)
    return dict(fms=fms, fPs=fPs, sms=sms, sPs=sPs, ll=np.array([ll]))


def _gpu_all(ssm, y, dtype):
    B = _gpu()
    ssm_t = tuple(np.asarray(a, dtype=dtype) for a in ssm)
    sms, sPs, fms, fPs, ll = B.pkfs(ssm_t, np.asarray(y, dtype), return_filtered=True, return_loglikelihood=True)
    return dict(fms=fms, fPs=fPs, sms=sms, sPs=sPs, ll=np.array([float(ll)]))


@pytest.mark.parametrize("idx", range(7))
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_pkfs_matches_oracle(kernel_zoo, idx, dtype):
    name, make, _, _ = kernel_zoo[idx]
    t = make_times(1500, seed=idx)
    ssm = O.get_ssm(make().get_sde(), t, 0.1)
    y = sample_series(ssm, seed=idx, nan_frac=0.2)
    want = _oracle_all(ssm, y)
    got = _gpu_all(ssm, y, dtype)
    _check_all(got, want, TOL64 if dtype == np.float64 else TOL32)


@pytest.mark.parametrize("n", [1, 2, 3, 63, 64, 65, 255, 256, 257, 1023, 1025, 4097])
def test_ragged_lengths(n):
    """Lengths around the lane / wavefront / workgroup boundaries, and N = 1."""
    from pssgp.kernels import Matern32
    t = make_times(n, seed=n)
    ssm = O.get_ssm(Matern32(1., 1.).get_sde(), t, 0.1)
    y = sample_series(ssm, seed=n, nan_frac=0.1 if n > 4 else 0.0)
    _check_all(_gpu_all(ssm, y, np.float64), _oracle_all(ssm, y), TOL64)


@pytest.mark.parametrize("chunk", [1, 2, 3, 5, 8, 16, 33])
def test_chunk_size_invariance(chunk):
    """Any steps-per-lane setting gives the same answer (different bracketing, round-off only)."""
    from pssgp.kernels import Matern52
    B = _gpu()
    ctx = B.get_context()
    t = make_times(3000, seed=7)
    ssm = O.get_ssm(Matern52(1., 1.).get_sde(), t, 0.1)
    y = sample_series(ssm, seed=7, nan_frac=0.15)
    try:
        ctx.set_chunk(chunk)
        _check_all(_gpu_all(ssm, y, np.float64), _oracle_all(ssm, y), TOL64)
    finally:
        ctx.set_chunk(0)


@pytest.mark.parametrize("lanes", [128, 256])
@pytest.mark.parametrize("dtype,kname,n,chunk", [(np.float64, "m32", 70001, 0), (np.float64, "m32", 1025, 3), (np.float32, "m32", 9000, 0),
                                                 (np.float64, "m52", 33000, 0), (np.float64, "rbf6", 5000, 0),
                                                 (np.float32, "rbf6", 12000, 8), (np.float64, "m12", 257, 0)])
def test_both_workgroup_sizes(lanes, dtype, kname, n, chunk):
    """libpgps carries the lane-chunk scan twice, with 128- and 256-lane workgroups (pgps_set_block; the automatic choice
    is 128 for whole-series calls): each build on its own against the oracle, ragged sizes and forced chunks included;
    stand-alone filter and smoother too."""
    from pssgp.kernels import Matern12, Matern32, Matern52, RBF
    B = _gpu()
    ctx = B.get_context()
    k = {"m12": lambda: Matern12(1., 1.), "m32": lambda: Matern32(1., 1.), "m52": lambda: Matern52(1., 0.7),
         "rbf6": lambda: RBF(1., 0.8, order=6, balancing_iter=10)}[kname]()
    t = make_times(n, seed=n % 97)
    ssm = O.get_ssm(k.get_sde(), t, 0.1)
    y = sample_series(ssm, seed=3, nan_frac=0.1)
    tol = TOL64 if dtype == np.float64 else TOL32
    try:
        ctx.set_family(1)
        ctx.set_block(lanes)
        ctx.set_chunk(chunk)
        _check_all(_gpu_all(ssm, y, dtype), _oracle_all(ssm, y), tol)
        if dtype == np.float64:
            of, oP, _ = O.kf(ssm, y, True)
            s2, sP2 = B.pks(ssm, of, oP)
            os_, osP = O.kfs(ssm, y)
            assert relerr(s2, os_) < 1e-9 and relerr(sP2, osP) < 1e-9
    finally:
        ctx.set_chunk(0)
        ctx.set_block(0)
        ctx.set_family(0)
    with pytest.raises(B.PgpsError):
        ctx.set_block(64)
    # what the automatic choice is, as the library reports it: c2 = 256 workgroups of 128 lanes x 32 steps; long staged
    # series and nothing else on 256 lanes
    assert ctx.get_geometry(1 << 20, 2) == (128, 32, 256) and ctx.get_geometry(1 << 19, 2) == (128, 16, 256)
    assert ctx.get_geometry(1 << 24, 2)[0] == 256 and ctx.get_geometry(1 << 22, 3)[0] == 256
    assert ctx.get_geometry(1 << 20, 6) == (128, 16, 512) and ctx.get_geometry(1 << 22, 6)[0] == 128


@pytest.mark.parametrize("stage", [0, 2, 4])
@pytest.mark.parametrize("chunk", [4, 8, 12])
@pytest.mark.parametrize("dtype,kname", [(np.float64, "m32"), (np.float64, "m12"), (np.float32, "m32"),
                                         (np.float64, "m52"), (np.float32, "m52")])
def test_lds_staged_paths(stage, chunk, dtype, kname):
    """The coalesced global<->LDS staged path (d <= 2) against the oracle, for every sub-tile
    size, with a ragged tail (the last wavefronts fall back to direct accesses)."""
    from pssgp.kernels import Matern12, Matern32, Matern52
    B = _gpu()
    ctx = B.get_context()
    k = {"m32": Matern32(1., 1.), "m12": Matern12(1., 1.), "m52": Matern52(1., 1.)}[kname]
    n = 256 * chunk * 3 + 64 * chunk + 17
    t = make_times(n, seed=chunk)
    ssm = O.get_ssm(k.get_sde(), t, 0.1)
    y = sample_series_fast(ssm, seed=chunk, nan_frac=0.1)
    want = _oracle_all(ssm, y)
    try:
        ctx.set_chunk(chunk)
        ctx.set_stage(stage)
        _check_all(_gpu_all(ssm, y, dtype), want, TOL64 if dtype == np.float64 else TOL32)
    finally:
        ctx.set_chunk(0)
        ctx.set_stage(-1)


@pytest.mark.parametrize("window", [1, 3, 256])
@pytest.mark.parametrize("dtype,kname", [(np.float64, "m32"), (np.float32, "m32"), (np.float64, "m12")])
def test_single_pass_filter_lookback(window, dtype, kname):
    """The single-pass filter kernel (workgroups hand their totals to each other inside one launch)
    against the oracle and against the three-launch path, with tiny look-back windows so that the
    window-closing inclusive-prefix hand-off is exercised, plus a ragged tail."""
    from pssgp.kernels import Matern12, Matern32
    B = _gpu()
    ctx = B.get_context()
    k = Matern32(1., 1.) if kname == "m32" else Matern12(1., 1.)
    n = 256 * 16 * 7 + 64 * 16 + 5          # 8 tiles, the last one ragged
    t = make_times(n, seed=window)
    ssm = O.get_ssm(k.get_sde(), t, 0.1)
    y = sample_series_fast(ssm, seed=window, nan_frac=0.1)
    from oracle import c_oracle as C
    cf, cP, cs, csP, cll = C.kfs(ssm, y)
    want = dict(fms=cf, fPs=cP, sms=cs, sPs=csP, ll=np.array([cll]))
    try:
        ctx.set_chunk(16)
        ctx.set_single_pass(1, window)
        got = _gpu_all(ssm, y, dtype)
        _check_all(got, want, TOL64 if dtype == np.float64 else TOL32)
        again = _gpu_all(ssm, y, dtype)
        for name in got:
            assert np.array_equal(got[name], again[name]), name      # timing-independent combine order
        assert ctx.status() == 0                                    # no look-back spin gave up
        ctx.set_single_pass(0, 0)
        ref = _gpu_all(ssm, y, dtype)
        tol = 1e-12 if dtype == np.float64 else 1e-4
        for name in got:
            assert relerr(got[name], ref[name]) < tol, name
    finally:
        ctx.set_chunk(0)
        ctx.set_single_pass(-1, 256)


def test_missing_data_patterns():
    """All observations missing, first missing, last missing, long gaps."""
    from pssgp.kernels import Matern32
    t = make_times(700, seed=3)
    ssm = O.get_ssm(Matern32(1., 1.).get_sde(), t, 0.1)
    base = sample_series(ssm, seed=3)
    pats = []
    y = base.copy(); y[:] = np.nan; pats.append(y)
    y = base.copy(); y[0] = np.nan; pats.append(y)
    y = base.copy(); y[-1] = np.nan; pats.append(y)
    y = base.copy(); y[100:600] = np.nan; pats.append(y)
    y = base.copy(); y[1::2] = np.nan; pats.append(y)
    for y in pats:
        want, got = _oracle_all(ssm, y), _gpu_all(ssm, y, np.float64)
        if np.all(np.isnan(y)):
            assert got["ll"][0] == 0.0
            want.pop("ll"); got.pop("ll")
        _check_all(got, want, TOL64)


@pytest.mark.parametrize("d", [2, 3, 6, 11, 18, 24])
def test_non_stationary_tuples_follow_the_parallel_reference(d):
    """An LGSSM tuple that does not come from a stationary SDE (F_0 P0 F_0^T + Q_0 != P0).  The reference's two paths differ
    there and a drop-in has to differ the same way: the parallel filter takes the prior itself as the first predicted state
    (parallel.py:13-43) but scores the first observation against F_0 P0 F_0^T + Q_0 (parallel.py:134-139); the sequential
    filter propagates before its first update (sequential.py:16-21).  pkf / pks / pkfs against the oracle's parallel
    restatement, kf / kfs (product host code) against its sequential one, and the two oracles must disagree."""
    from pssgp.kalman.parallel import pkf, pkfs
    from pssgp.kalman.sequential import kf, kfs
    from pssgp.kalman.base import LGSSM
    for n in (1, 2, 37, 700):
        rng = np.random.default_rng(100 * d + n)
        L = rng.standard_normal((d, d)) * 0.4
        P0 = L @ L.T + 0.5 * np.eye(d)
        Fs = np.stack([0.8 * np.eye(d) + 0.15 * rng.standard_normal((d, d)) / np.sqrt(d) for _ in range(n)])
        Lq = rng.standard_normal((n, d, d)) * 0.2
        Qs = Lq @ np.transpose(Lq, (0, 2, 1)) + 0.05 * np.eye(d)
        y = rng.standard_normal(n); y[rng.uniform(size=n) < 0.2] = np.nan
        y[0] = 0.7 if d % 2 else np.nan
        ssm = LGSSM(P0, Fs, Qs, rng.standard_normal((1, d)), np.array([[0.3]]))
        fms, fPs, ll = pkf(ssm, y[:, None], return_loglikelihood=True)
        sms, sPs = pkfs(ssm, y[:, None])
        of, oP, oll = O.pkf(ssm, y, True)
        os_, osP = O.pkfs(ssm, y)
        assert relerr(fms, of) < TOL64 and relerr(fPs, oP) < TOL64 and relerr(sms, os_) < TOL64 and relerr(sPs, osP) < TOL64
        assert abs(float(ll) - oll) <= TOL64 * max(1.0, abs(oll))
        kfm, kfP, kll = kf(ssm, y[:, None], return_loglikelihood=True)
        ksm, ksP = kfs(ssm, y[:, None])
        qf, qP, qll = O.kf(ssm, y, True)
        qs, qsP = O.kfs(ssm, y)
        assert relerr(kfm, qf) < TOL64 and relerr(kfP, qP) < TOL64 and relerr(ksm, qs) < TOL64 and relerr(ksP, qsP) < TOL64
        assert abs(float(kll) - qll) <= TOL64 * max(1.0, abs(qll))
        if n > 1:
            assert relerr(of, qf) > 1e-4, "the inputs do not separate the two paths"


def test_duplicate_times():
    """dt = 0 steps: F = I, Q = 0 (SURVEY 'semantics to preserve')."""
    from pssgp.kernels import Matern32
    t = make_times(300, seed=11)
    t[50:60] = t[50]
    t[200] = t[199]
    ssm = O.get_ssm(Matern32(1., 1.).get_sde(), t, 0.1)
    y = sample_series(ssm, seed=11)
    _check_all(_gpu_all(ssm, y, np.float64), _oracle_all(ssm, y), TOL64)


def test_pkf_and_pks_separately():
    """The reference's separate entry points: pkf (with and without ll), then pks on its output."""
    from pssgp.kalman.parallel import pkf, pks, pkfs
    from pssgp.kalman.base import LGSSM
    from pssgp.kernels import Matern52
    t = make_times(2100, seed=5)
    ssm = LGSSM(*O.get_ssm(Matern52(1., 0.7).get_sde(), t, 0.1))
    y = sample_series(ssm, seed=5, nan_frac=0.2)
    want = _oracle_all(ssm, y)
    fms, fPs = pkf(ssm, y[:, None])
    fms2, fPs2, ll = pkf(ssm, y[:, None], return_loglikelihood=True, max_parallel=4096)
    assert np.array_equal(fms, fms2) and np.array_equal(fPs, fPs2)
    sms, sPs = pks(ssm, fms, fPs)
    sms2, sPs2 = pkfs(ssm, y[:, None])
    _check_all(dict(fms=fms, fPs=fPs, sms=sms, sPs=sPs, ll=np.array([float(ll)])), want, TOL64)
    assert relerr(sms2, sms) < 1e-12 and relerr(sPs2, sPs) < 1e-12


def test_bitwise_deterministic():
    """Fixed launch geometry, fixed combine order: repeated runs are bit-identical."""
    from pssgp.kernels import Matern32
    t = make_times(50000, seed=2)
    ssm = O.get_ssm(Matern32(1., 1.).get_sde(), t, 0.1)
    y = sample_series_fast(ssm, seed=2, nan_frac=0.1)
    a = _gpu_all(ssm, y, np.float64)
    b = _gpu_all(ssm, y, np.float64)
    for k in a:
        assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("idx", [1, 2, 3, 5, 6])
def test_discretise_matches_reference_formula(kernel_zoo, idx):
    """GPU discretisation vs the reference's matrix-fraction expm (restated in the oracle)."""
    B = _gpu()
    name, make, _, _ = kernel_zoo[idx]
    sde = make().get_sde()
    t = make_times(2000, seed=idx)
    t[10] = t[9]                               # a zero step
    P0, Fs, Qs, H, R = O.get_ssm(sde, t, 0.1)
    gFs, gQs = B.discretise(sde.F, sde.P0, t, 0.0)
    scale = max(1.0, float(np.max(np.abs(P0))))
    assert np.max(np.abs(gFs - Fs)) < 1e-12 * max(1.0, float(np.max(np.abs(Fs))))
    assert np.max(np.abs(gQs - Qs)) < 1e-12 * scale
    gFs32, gQs32 = B.discretise(sde.F.astype(np.float32), sde.P0.astype(np.float32), t.astype(np.float32), 0.0)
    assert np.max(np.abs(gFs32 - Fs)) < 2e-4 * max(1.0, float(np.max(np.abs(Fs))))


def test_golden_c1_statespacegp():
    """BASELINE config c1 end to end through the public API: StateSpaceGP(parallel=True)
    log-likelihood and predict_f against the committed dense-GP golden vectors."""
    from pssgp.kernels import Matern32
    from pssgp.model import StateSpaceGP
    g = np.load(os.path.join(GOLD, "c1_matern32_n4096.npz"))
    k = Matern32(variance=float(g["variance"]), lengthscales=float(g["lengthscale"]))
    model = StateSpaceGP((g["t"][:, None], g["y"][:, None]), k, noise_variance=float(g["noise"]), parallel=True,
                         max_parallel=8192)
    ll = model.maximum_log_likelihood_objective()
    assert abs(float(ll) - float(g["ll_dense"])) < 1e-5 * abs(float(g["ll_dense"]))
    mean, var = model.predict_f(g["tq"][:, None])
    assert mean.shape == (1024, 1) and var.shape == (1024, 1)
    assert np.max(np.abs(mean[:, 0] - g["mean_dense"])) < 1e-5 * np.max(np.abs(g["mean_dense"]))
    assert np.max(np.abs(var[:, 0] - g["var_dense"])) < 1e-5 * np.max(np.abs(g["var_dense"]))


def test_golden_small_d():
    g = np.load(os.path.join(GOLD, "small_d_n1024.npz"))
    from pssgp.kernels import RBF, Matern32, Matern52
    from pssgp.kalman.parallel import pkf, pkfs
    kernels = {"rbf6": RBF(1., 1., order=6, balancing_iter=10), "m32+m52": Matern32(1., 1.) + Matern52(1., 1.),
               "m32*m52": Matern32(1., 1.) * Matern52(1., 1.)}
    for name, k in kernels.items():
        ssm = k.get_ssm(g["t"][:, None], 0.1)
        fms, fPs, ll = pkf(ssm, g["y"][:, None], return_loglikelihood=True)
        sms, sPs = pkfs(ssm, g["y"][:, None])
        h = np.asarray(ssm.H).reshape(-1)
        assert abs(float(ll) - float(g[name + "/ll"])) < 1e-7 * abs(float(g[name + "/ll"]))
        assert relerr(fms @ h, g[name + "/fmean"]) < 1e-7
        assert relerr(sms @ h, g[name + "/smean"]) < 1e-7
        assert relerr(np.einsum("i,nij,j->n", h, sPs, h), g[name + "/svar"]) < 1e-7


@pytest.mark.parametrize("idx", range(7))
def test_statespacegp_equals_dense_gp(kernel_zoo, idx):
    """The reference's own equivalence test (tests/test_gp_vs_kfs.py) on the HIP backend, for
    the kernels whose state-space form is exact; the approximate ones are checked against the
    oracle's state-space result at 1e-9."""
    from pssgp.model import StateSpaceGP
    name, make, spec, tol = kernel_zoo[idx]
    rng = np.random.RandomState(31415926)
    T, K = 200, 50
    t = np.sort(rng.rand(T))
    f = np.sin(np.pi * t) + np.sin(2 * np.pi * t) + np.cos(3 * np.pi * t)
    y = f + np.sqrt(0.1) * rng.normal(f, np.sqrt(0.1), (T,))
    tq = np.sort(rng.rand(K))
    k = make()
    for parallel in (False, True):
        m = StateSpaceGP((t[:, None], y[:, None]), k, noise_variance=0.1, parallel=parallel, max_parallel=T + K)
        ll = float(m.maximum_log_likelihood_objective())
        mean, var = m.predict_f(tq[:, None])
        if spec is not None:
            ll_gp, mean_gp, var_gp = O.dense_gp(spec, t, y, 0.1, tq)
            np.testing.assert_allclose(ll, ll_gp, atol=tol, rtol=tol)
            np.testing.assert_allclose(mean[:, 0], mean_gp, atol=tol, rtol=tol)
            np.testing.assert_allclose(var[:, 0], var_gp, atol=tol, rtol=tol)
        sde = k.get_sde()
        ll_o = O.ssgp_log_likelihood(sde, t, y, 0.1, parallel=False)
        mean_o, var_o = O.ssgp_predict_f(sde, t, y, 0.1, tq, parallel=False)
        assert abs(ll - ll_o) < 1e-8 * abs(ll_o)
        assert np.max(np.abs(mean[:, 0] - mean_o)) < 1e-8 and np.max(np.abs(var[:, 0] - var_o)) < 1e-8


@pytest.mark.parametrize("dtype,kname,n", [(np.float64, "m32", 1 << 20), (np.float32, "rbf6", 1 << 18),
                                           (np.float32, "rbf6", 1 << 20), (np.float64, "c5", 1 << 20),
                                           (np.float64, "m32", 1 << 21), (np.float64, "m32", 1 << 24)])
def test_full_size_against_c_oracle(dtype, kname, n):
    """BASELINE sizes -- N = 2^20 Matern-3/2 fp64 (config c2), RBF order 6 fp32 (c3), the quasi-periodic d = 11 kernel
    fp64 (c5), 2^21 and 2^24 Matern-3/2 (one GPU's share of c4, and all of it) -- against the C sequential oracle on the same arrays, plus
    size-independent properties: the smoothed state of the last step equals its filtered state and smoothed variances
    never exceed filtered ones."""
    from pssgp.kernels import Matern32, Matern52, Periodic, RBF, SquaredExponential
    B = _gpu()
    k = {"m32": lambda: Matern32(1., 1.), "rbf6": lambda: RBF(1., 1., order=6, balancing_iter=10),
         "c5": lambda: Periodic(SquaredExponential(1., 1.), period=1., order=1) * Matern32(1., 1.) + Matern52(1., 1.)}[kname]()
    sde = k.get_sde()
    t = make_times(n, seed=0)
    Fs, Qs = B.discretise(sde.F, sde.P0, t, 0.0)                    # fp64 discretisation on the GPU
    ssm = (sde.P0, Fs, Qs, np.asarray(sde.H).reshape(1, -1), np.array([[0.1]]))
    y = sample_series_fast(ssm, seed=0, nan_frac=0.2)
    cf, cP, cs, csP, cll = C.kfs(ssm, y, np.float64)
    got = _gpu_all(ssm, y, dtype)
    tol = 1e-7 if dtype == np.float64 else TOL32
    assert relerr(got["fms"], cf) < tol and relerr(got["fPs"], cP) < tol
    assert relerr(got["sms"], cs) < tol and relerr(got["sPs"], csP) < tol
    assert abs(got["ll"][0] - cll) < (1e-9 if dtype == np.float64 else 1e-4) * abs(cll)
    # properties
    # (exact in fp64 and on the lane-chunk kernels; the fp32 row-cooperative kernels symmetrise the last smoothed
    # covariance once more: a rounding of an already symmetric matrix in the last bit of its small entries)
    ptol = 1e-12 if dtype == np.float64 else 1e-6
    assert relerr(got["sms"][-1], got["fms"][-1]) < ptol and relerr(got["sPs"][-1], got["fPs"][-1]) < ptol
    d = Fs.shape[1]
    slack = 1e-9 if dtype == np.float64 else 1e-3
    for i in range(d):
        assert np.all(got["sPs"][:, i, i] <= got["fPs"][:, i, i] * (1 + slack) + slack)


@pytest.mark.parametrize("n", [4096, 32768, 1 << 20])
@pytest.mark.parametrize("kname", ["m32", "m52", "rbf6"])
def test_fp32_on_the_reference_grid(kname, n):
    """The reference's own benchmark grid -- N equally spaced points on [0, 4] (pssgp/experiments/toy_models/common.py:31-32,
    with --dtype float32: speed_and_stability.py:68) -- in float32 at the north star's tolerance (1e-3), against the fp64 C
    oracle on the same (fp64-discretised) model.  At 2^20 points the spacing is 3.8e-6: F is the identity to five digits
    and Q is a million times smaller than Pinf.

    float32 ARITHMETIC cannot hold 1e-3 on the smoothed moments of RBF order 6 there (round 3 measured 3.7e-3 at 32768
    points and 1.1e-1 at 2^20 in every kernel family, and 1.1e-2 for the sequential RTS form in float32:
    profiles/r03_fp32_reference_grid.txt), so since round 4 a float32 call that runs a smoother probes the grid and, where
    it is that dense, computes in fp64 on the float32 arrays (pgps_set_f32_policy, include/pgps.h): every kernel, every
    size holds TOL32, and pgps_status tells which calls were promoted (profiles/r04_fp32_reference_grid.txt)."""
    from pssgp.kernels import Matern32, Matern52, RBF
    B = _gpu()
    ctx = B.get_context()
    k = {"m32": lambda: Matern32(1., 1.), "m52": lambda: Matern52(1., 1.),
         "rbf6": lambda: RBF(1., 1., order=6, balancing_iter=10)}[kname]()
    sde = k.get_sde()
    t = np.linspace(0.0, 4.0, n)
    Fs, Qs = B.discretise(sde.F, sde.P0, t, 0.0)
    ssm = (sde.P0, Fs, Qs, np.asarray(sde.H).reshape(1, -1), np.array([[0.1]]))
    y = sample_series_fast(ssm, seed=n % 89, nan_frac=0.1)
    cf, cP, cs, csP, cll = C.kfs(ssm, y, np.float64)
    ctx.status()
    got = _gpu_all(ssm, y, np.float32)
    promoted = bool(ctx.status() & 4)
    assert got["sms"].dtype == np.float32
    assert relerr(got["fms"], cf) < TOL32 and relerr(got["fPs"], cP) < TOL32
    assert relerr(got["sms"], cs) < TOL32 and relerr(got["sPs"], csP) < TOL32
    assert abs(got["ll"][0] - cll) < TOL32 * abs(cll)
    # the two cases round 3 measured beyond the tolerance are among the promoted ones
    if kname == "rbf6" and n >= 32768:
        assert promoted
    # a filter-only call is never probed (its float32 arithmetic holds 1e-3 on every grid measured)
    fms, fPs, ll = B.pkf(tuple(np.asarray(a, np.float32) for a in ssm), y.astype(np.float32), return_loglikelihood=True)
    assert not (ctx.status() & 4)
    assert relerr(fms, cf) < TOL32 and relerr(fPs, cP) < TOL32 and abs(float(ll) - cll) < TOL32 * abs(cll)


def test_fp32_policy_and_status():
    """pgps_set_f32_policy: on BASELINE's c3 grid (steps of ~0.05: far from dense) the automatic policy keeps float32
    arithmetic -- the configuration bench.py measures is not touched; policy 2 forces fp64 arithmetic (and is then as
    close to the oracle as the float32 arrays allow), policy 1 keeps float32 arithmetic even on the dense grid, where it
    visibly misses what the promoted call holds; the stand-alone smoother (pks) takes the same road."""
    from pssgp.kernels import RBF
    B = _gpu()
    ctx = B.get_context()
    sde = RBF(1., 1., order=6, balancing_iter=10).get_sde()
    H, R = np.asarray(sde.H).reshape(1, -1), np.array([[0.1]])
    try:
        # c3's grid
        t = make_times(20000, seed=0)
        Fs, Qs = B.discretise(sde.F, sde.P0, t, 0.0)
        ssm = (sde.P0, Fs, Qs, H, R)
        y = sample_series_fast(ssm, seed=0, nan_frac=0.2)
        ref = C.kfs(ssm, y, np.float64)
        ctx.status()
        native = _gpu_all(ssm, y, np.float32)
        assert not (ctx.status() & 4)
        ctx.set_f32_policy(2)
        wide = _gpu_all(ssm, y, np.float32)
        assert ctx.status() & 4
        assert relerr(native["sPs"], ref[3]) < TOL32 and relerr(wide["sPs"], ref[3]) < 1e-5
        # the dense grid
        t = np.linspace(0.0, 4.0, 32768)
        Fs, Qs = B.discretise(sde.F, sde.P0, t, 0.0)
        ssm = (sde.P0, Fs, Qs, H, R)
        y = sample_series_fast(ssm, seed=1, nan_frac=0.1)
        ref = C.kfs(ssm, y, np.float64)
        ctx.set_f32_policy(1)
        native = _gpu_all(ssm, y, np.float32)
        assert not (ctx.status() & 4)
        ctx.set_f32_policy(0)
        auto = _gpu_all(ssm, y, np.float32)
        assert ctx.status() & 4
        assert relerr(auto["sPs"], ref[3]) < 1e-4 < relerr(native["sPs"], ref[3])
        # pks on the float32 filtered moments of the promoted call
        ssm32 = tuple(np.asarray(a, np.float32) for a in ssm)
        sms, sPs = B.pks(ssm32, auto["fms"], auto["fPs"])
        assert ctx.status() & 4
        assert sms.dtype == np.float32 and relerr(sms, ref[2]) < TOL32 and relerr(sPs, ref[3]) < TOL32
    finally:
        ctx.set_f32_policy(0)


@pytest.mark.parametrize("kname", ["m12", "m32", "m52"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n,chunk", [(1, 0), (5, 0), (300, 0), (5000, 0), (256 * 16 * 3 + 77, 16), (256 * 8 * 2 + 9, 8),
                                     (9000, 5)])
def test_fused_discretisation_path(kname, dtype, n, chunk):
    """pgps_gp_*: (t, y) in, filtered / smoothed moments and log-likelihood out, F_k and Q_k formed in
    registers -- against the oracle fed with the reference's expm discretisation."""
    from pssgp.kernels import Matern12, Matern32, Matern52
    B = _gpu()
    ctx = B.get_context()
    k = {"m12": Matern12(1.3, 0.7), "m32": Matern32(1.3, 0.7), "m52": Matern52(1.3, 0.7)}[kname]
    sde = k.get_sde()
    form = B.nilpotent_form(sde.F)
    assert form is not None
    t = make_times(n, seed=n % 89)
    if n > 10:
        t[7] = t[6]                                            # a zero step
    ssm = O.get_ssm(sde, t, 0.1)
    y = sample_series_fast(ssm, seed=n % 89, nan_frac=0.15 if n > 4 else 0.0)
    from oracle import c_oracle as C
    cf, cP, cs, csP, cll = C.kfs(ssm, y)
    tol = TOL64 if dtype == np.float64 else TOL32
    try:
        ctx.set_chunk(chunk)
        res = B.gp(form, sde.P0, sde.H, 0.1, t.astype(dtype), y.astype(dtype), want_smoothed=True)
        assert relerr(res["fms"], cf) < tol and relerr(res["fPs"], cP) < tol
        assert relerr(res["sms"], cs) < tol and relerr(res["sPs"], csP) < tol
        lltol = 1e-9 if dtype == np.float64 else 1e-4
        if not np.all(np.isnan(y)):
            assert abs(float(res["ll"]) - cll) < lltol * max(1.0, abs(cll))
        only_ll = B.gp(form, sde.P0, sde.H, 0.1, t.astype(dtype), y.astype(dtype))
        assert float(only_ll["ll"]) == float(res["ll"])                     # same arithmetic, nothing stored
        filt = B.gp(form, sde.P0, sde.H, 0.1, t.astype(dtype), y.astype(dtype), want_filtered=True)
        assert np.array_equal(filt["fms"], res["fms"]) and np.array_equal(filt["fPs"], res["fPs"])
    finally:
        ctx.set_chunk(0)


def test_nilpotent_form_detection():
    from pssgp.kernels import Matern12, Matern32, Matern52, RBF, Periodic, SquaredExponential
    B = _gpu()
    for k in (Matern12(2., 0.3), Matern32(1., 1.), Matern52(0.5, 2.)):
        lam, N1, N2 = B.nilpotent_form(k.get_sde().F)
        F = np.asarray(k.get_sde().F)
        assert np.allclose(-lam * np.eye(F.shape[0]) + N1, F) and np.allclose(N2, 0.5 * N1 @ N1)
    assert B.nilpotent_form(RBF(1., 1., order=3).get_sde().F) is None
    assert B.nilpotent_form(Periodic(SquaredExponential(1., 1.), 1., order=1).get_sde().F) is None
    assert B.nilpotent_form((Matern32() + Matern12()).get_sde().F) is None


def test_two_host_threads_with_their_own_contexts():
    """one context per host thread.  Two threads, each with its own context on the same GPU, run
    different workloads (lane-chunk d = 2, row-cooperative d = 11) concurrently and repeatedly; every result equals
    the single-threaded one bit for bit."""
    import ctypes
    import threading
    from pssgp import _backend as B
    from pssgp.kernels import Matern32, Matern52, Periodic, SquaredExponential
    kernels = [Matern32(1., 1.), Periodic(SquaredExponential(1., 1.), period=1., order=1) * Matern32(1., 1.) + Matern52(1., 1.)]
    jobs = []
    for i, k in enumerate(kernels):
        t = make_times(30011 if i == 0 else 6007, seed=40 + i)
        ssm = O.get_ssm(k.get_sde(), t, 0.1)
        y = sample_series(ssm, seed=40 + i, nan_frac=0.1)
        jobs.append((ssm, y))

    def run(ctx, ssm, y):
        P0, Fs, Qs, H, R = (np.ascontiguousarray(a, np.float64) for a in ssm)
        N, d = Fs.shape[0], Fs.shape[1]
        sms, sPs = np.empty((N, d)), np.empty((N, d, d))
        fms, fPs = np.empty((N, d)), np.empty((N, d, d))
        ll = ctypes.c_double(0.0)
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        ctx.call("pgps_pkfs_f64", ctypes.c_long(N), ctypes.c_int(d), p(P0), p(Fs), p(Qs), p(H.reshape(-1)),
                 ctypes.c_double(float(R.reshape(()))), p(np.ascontiguousarray(y)), p(fms), p(fPs), p(sms), p(sPs),
                 ctypes.cast(ctypes.byref(ll), ctypes.c_void_p))
        return sms, sPs, ll.value

    ctxs = [B.Context(0), B.Context(0)]
    try:
        want = [run(ctxs[i], *jobs[i]) for i in range(2)]
        errors = []

        def worker(i):
            try:
                for _ in range(15):
                    got = run(ctxs[i], *jobs[i])
                    if not (np.array_equal(got[0], want[i][0]) and np.array_equal(got[1], want[i][1]) and got[2] == want[i][2]):
                        errors.append(f"thread {i}: result changed under concurrency")
                        return
            except Exception as e:                         # noqa: BLE001
                errors.append(f"thread {i}: {e!r}")

        threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
        for th in threads:
            th.start()
        for th in threads:
            th.join(timeout=120)
        assert not any(th.is_alive() for th in threads), "a worker did not finish"
        assert not errors, errors
    finally:
        for c in ctxs:
            c.close()
    os_, osP = O.kfs(jobs[1][0], jobs[1][1])
    assert relerr(want[1][0], os_) < 1e-9 and relerr(want[1][1], osP) < 1e-9


def test_two_host_threads_on_the_shared_default_context():
    """Two Python threads evaluating models through the module-level wrappers share ONE context per device
    (`_backend.get_context`) -- parallel MCMC chains in the reference's drivers do exactly that under TensorFlow.  ctypes
    releases the GIL, libpgps allows one call in flight per context: the context's lock makes the threads take turns.
    Different workloads (fused Matern path, general-LTI path, array path) hammer the shared scratch; every result must
    equal the single-threaded one bit for bit."""
    import threading
    from pssgp.kernels import Matern32, Matern52, RBF
    from pssgp.model import StateSpaceGP
    from pssgp.kalman.parallel import pkfs
    rng = np.random.default_rng(8)
    models = []
    for i, k in enumerate([Matern32(1.3, 0.7), RBF(1., 0.8, order=6, balancing_iter=10), Matern52(0.9, 1.1)]):
        t = make_times(20011 + 977 * i, seed=60 + i)
        y = np.sin(t) + 0.3 * rng.standard_normal(t.size)
        models.append(StateSpaceGP((t[:, None], y[:, None]), k, noise_variance=0.2, parallel=True))
    ta = make_times(9001, seed=70)
    ssm = O.get_ssm(Matern32(1., 1.).get_sde(), ta, 0.1)
    ya = sample_series(ssm, seed=70, nan_frac=0.1)

    def job(i):
        if i < 3:
            m = models[i]
            tq = np.linspace(0.1, float(m.data[0][-1, 0]), 501)[:, None]
            mean, var = m.predict_f(tq)
            return float(m.maximum_log_likelihood_objective()), mean.copy(), var.copy()
        sms, sPs = pkfs(ssm, ya)
        return 0.0, np.asarray(sms).copy(), np.asarray(sPs).copy()

    for i in range(4):
        job(i)          # (a model's first evaluation goes through the host-array entry points, the later ones through its
                        # resident series: compare like with like)
    want = [job(i) for i in range(4)]
    errors = []

    def worker(w):
        try:
            for rep in range(12):
                for i in ((0, 1, 2, 3) if w == 0 else (3, 2, 1, 0)):
                    got = job(i)
                    if not (got[0] == want[i][0] and np.array_equal(got[1], want[i][1]) and np.array_equal(got[2], want[i][2])):
                        errors.append(f"thread {w}, job {i}, repetition {rep}: result changed under concurrency")
                        return
        except Exception as e:                             # noqa: BLE001
            errors.append(f"thread {w}: {e!r}")

    threads = [threading.Thread(target=worker, args=(w,)) for w in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=300)
    assert not any(th.is_alive() for th in threads), "a worker did not finish"
    assert not errors, errors


def test_error_codes_across_the_abi():
    """Errors are return codes, never exceptions or faults (include/pgps.h; the reference raises
    InvalidArgumentError from TensorFlow): bad sizes, unsupported state dimensions, null and misaligned pointers, a
    model whose log-likelihood is not finite."""
    import ctypes
    from pssgp import _backend as B
    lib, ctx = B.load_library(), B.get_context()
    h = ctx.handle
    P = ctypes.c_void_p
    N, d = 64, 2
    t = make_times(N, seed=1)
    from pssgp.kernels import Matern32
    ssm = O.get_ssm(Matern32(1., 1.).get_sde(), t, 0.1)
    P0, Fs, Qs, Hm, R = (np.ascontiguousarray(a, np.float64) for a in ssm)
    y = sample_series(ssm, seed=1)
    sms, sPs = np.empty((N, d)), np.empty((N, d, d))
    ll = ctypes.c_double(0.0)
    p = lambda a: a.ctypes.data_as(P)
    args = lambda n, dim: (h, ctypes.c_long(n), ctypes.c_int(dim), p(P0), p(Fs), p(Qs), p(Hm.reshape(-1)),
                           ctypes.c_double(0.1), p(y), None, None, p(sms), p(sPs), ctypes.cast(ctypes.byref(ll), P))
    assert lib.pgps_pkfs_f64(*args(N, d)) == 0
    assert lib.pgps_pkfs_f64(*args(0, d)) == -1                        # PGPS_E_INVALID
    assert lib.pgps_pkfs_f64(*args(N, 33)) == -2                       # PGPS_E_UNSUPPORTED_DIM
    assert lib.pgps_pkfs_f64(*args(N, 0)) in (-1, -2)
    bad = list(args(N, d))
    bad[4] = None                                                       # Fs = NULL
    assert lib.pgps_pkfs_f64(*bad) != 0
    # device entry point with a misaligned pointer
    dptr = ctx.malloc(Fs.nbytes + 64)
    try:
        rc = lib.pgps_pkf_dev_f64(h, ctypes.c_long(N), ctypes.c_int(d), P(dptr), P(dptr + 8), P(dptr), P(dptr),
                                  ctypes.c_double(0.1), P(dptr), P(dptr), P(dptr), None)
        assert rc == -1
    finally:
        ctx.free(dptr)
    # non-finite result: negative noise variance large enough to make an innovation variance negative
    with pytest.raises(B.PgpsError) as err:
        B.lti_ll(np.array([[-1., 0.], [0., -2.]]), np.diag([1., 1.]), np.array([1., 1.]), -5.0, t, y)
    assert err.value.code == -5                                         # PGPS_E_NUMERIC
    # the context is still usable afterwards
    assert lib.pgps_pkfs_f64(*args(N, d)) == 0
    assert isinstance(lib.pgps_strerror(-1), bytes) and len(lib.pgps_strerror(-1)) > 0
