"""Pins the CPU oracle (oracle/) to what the reference's own tests pin
(/root/reference/tests/test_gp_vs_kfs.py): state-space log-likelihood and posterior == dense GP.
The dense GP (oracle.np_oracle.dense_gp) shares no code with the state-space routines.
Also pins the C restatement (oracle/kalman_seq.c) to the numpy one, and the committed golden
fixtures to the oracle that generated them."""
import os

import numpy as np
import pytest

from oracle import np_oracle as O
from oracle import c_oracle as C
from tests.conftest import relerr

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _toy_data(T=200, K=50, seed=31415926):
    """Same distribution as tests/test_gp_vs_kfs.py:27-31 of the reference (seeded here;
    the reference's noise is unseeded): t = sort(U(0,1)^T), y = obs_noise(sinu(t), 0.1)."""
    rng = np.random.RandomState(seed)
    t = np.sort(rng.rand(T))
    f = np.sin(np.pi * t) + np.sin(2 * np.pi * t) + np.cos(3 * np.pi * t)
    y = f + np.sqrt(0.1) * rng.normal(f, np.sqrt(0.1), (T,))      # data_funcs.py:95-97 literally
    tq = np.sort(rng.rand(K))
    return t, y, tq


def _reference_kernels():
    from pssgp.kernels import Matern12, Matern32, Matern52, RBF, Periodic, SquaredExponential
    m32, m52 = Matern32(1., 0.5), Matern52(1., 0.5)
    return [
        (Matern12(1., 0.5), ("matern12", 1., 0.5), 1e-6),
        (m32, ("matern32", 1., 0.5), 1e-6),
        (m52, ("matern52", 1., 0.5), 1e-6),
        (RBF(1., 0.5, order=15, balancing_iter=10), ("rbf", 1., 0.5), 1e-2),
        (Periodic(SquaredExponential(1., 0.5), period=0.5, order=10), ("periodic", 1., 0.5, 0.5), 1e-3),
        (m32 + m52, ("sum", [("matern32", 1., 0.5), ("matern52", 1., 0.5)]), 1e-6),
        (m32 * m52, ("prod", [("matern32", 1., 0.5), ("matern52", 1., 0.5)]), 1e-6),
    ]


@pytest.mark.parametrize("idx", range(7))
def test_loglikelihood_equals_dense_gp(idx):
    """tests/test_gp_vs_kfs.py:45-73 (values; gradients are out of scope)."""
    kernel, spec, tol = _reference_kernels()[idx]
    t, y, _ = _toy_data()
    ll_gp = O.dense_gp(spec, t, y, 0.1)
    sde = kernel.get_sde()
    for parallel in (False, True):
        ll_ss = O.ssgp_log_likelihood(sde, t, y, 0.1, parallel=parallel)
        np.testing.assert_allclose(ll_ss, ll_gp, atol=tol, rtol=tol)


@pytest.mark.parametrize("idx", range(7))
def test_posterior_equals_dense_gp(idx):
    """tests/test_gp_vs_kfs.py:80-99."""
    kernel, spec, tol = _reference_kernels()[idx]
    t, y, tq = _toy_data()
    _, mean_gp, var_gp = O.dense_gp(spec, t, y, 0.1, tq)
    sde = kernel.get_sde()
    for parallel in (False, True):
        mean_ss, var_ss = O.ssgp_predict_f(sde, t, y, 0.1, tq, parallel=parallel)
        np.testing.assert_allclose(mean_ss, mean_gp, atol=tol, rtol=tol)
        np.testing.assert_allclose(var_ss, var_gp, atol=tol, rtol=tol)


def test_parallel_equals_sequential_and_bracketing():
    """Associativity: tree (tfp) bracketing == left fold == sequential filter, to round-off."""
    from pssgp.kernels import Matern32
    t, y, _ = _toy_data(T=257)
    y = y.copy()
    y[::9] = np.nan
    ssm = O.get_ssm(Matern32(1., 0.5).get_sde(), t, 0.1)
    fms, fPs, ll = O.kf(ssm, y, True)
    sms, sPs = O.kfs(ssm, y)
    for br in ("tree", "sequential"):
        pf, pP, pll = O.pkf(ssm, y, True, br)
        ps, psP = O.pks(ssm, pf, pP, br)
        assert relerr(pf, fms) < 1e-12 and relerr(pP, fPs) < 1e-12
        assert relerr(ps, sms) < 1e-12 and relerr(psP, sPs) < 1e-12
        assert abs(pll - ll) < 1e-10 * abs(ll)


def test_every_prefix_has_zero_A():
    """The structural fact the GPU down-sweep relies on (SURVEY hard-parts (ii))."""
    from pssgp.kernels import Matern52
    t, y, _ = _toy_data(T=64)
    ssm = O.get_ssm(Matern52(1., 0.5).get_sde(), t, 0.1)
    P0, Fs, Qs, H, R = ssm
    el = O.make_associative_filtering_elements(np.zeros(3), P0, Fs, Qs, H, R, y)
    fin = O.scan_associative(O.filtering_operator, el)
    assert np.max(np.abs(fin[0])) == 0.0


def test_stationary_Q_identity():
    """Qs from the reference's matrix-fraction expm == Pinf - F Pinf F^T (what the GPU uses)."""
    from pssgp.kernels import Matern32, Matern52, RBF
    t = np.cumsum(np.random.default_rng(3).uniform(0.01, 0.2, 50))
    for k in (Matern32(1.3, 0.7), Matern52(0.8, 1.1), RBF(1., 0.9, order=6),
              Matern32(1., 1.) * Matern52(1., 1.)):
        sde = k.get_sde()
        P0, Fs, Qs, *_ = O.get_ssm(sde, t, 0.1)
        Q2 = P0[None] - Fs @ P0[None] @ np.swapaxes(Fs, 1, 2)
        assert np.max(np.abs(Qs - Q2)) < 1e-12 * max(1.0, np.max(np.abs(P0)))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_c_oracle_matches_numpy_oracle(dtype):
    from pssgp.kernels import Matern32, Matern52, RBF
    rng = np.random.default_rng(5)
    t = np.cumsum(0.05 * rng.uniform(0.5, 1.5, 700))
    for k in (Matern32(1., 1.), RBF(1., 1., order=6, balancing_iter=10), Matern32(1., 1.) + Matern52(1., 1.)):
        ssm = O.get_ssm(k.get_sde(), t, 0.1)
        y = rng.standard_normal(t.size)
        y[rng.random(t.size) < 0.2] = np.nan
        fms, fPs, ll = O.kf(ssm, y, True)
        sms, sPs = O.kfs(ssm, y)
        cf, cP, cs, csP, cll = C.kfs(ssm, y, dtype)
        tol = 1e-11 if dtype == np.float64 else 2e-3
        assert relerr(cf, fms) < tol and relerr(cP, fPs) < tol
        assert relerr(cs, sms) < tol and relerr(csP, sPs) < tol
        assert abs(cll - ll) < tol * abs(ll)


@pytest.mark.parametrize("nthreads", [1, 3, 8])
def test_all_cores_c_oracle_matches_sequential_c_oracle(nthreads):
    """oracle/kalman_par.c (chunked scan with OpenMP: the all-cores CPU baseline of bench.py, a C restatement of the
    filtering / smoothing elements and operators of pssgp/kalman/parallel.py at chunk granularity) against the
    sequential C oracle and the numpy oracle, with missing observations, ragged chunk sizes and fewer steps than
    chunks."""
    from pssgp.kernels import Matern32, Matern52, RBF
    rng = np.random.default_rng(9)
    for k, n in ((Matern32(1., 1.), 5003), (Matern32(1., 1.) + Matern52(1., .5), 1201),
                 (RBF(1., 1., order=6, balancing_iter=10), 700), (Matern52(1., 1.), 5), (Matern32(1., 1.), 1)):
        t = np.cumsum(0.05 * rng.uniform(0.5, 1.5, n))
        ssm = O.get_ssm(k.get_sde(), t, 0.1)
        y = rng.standard_normal(n)
        y[rng.random(n) < 0.2] = np.nan
        cf, cP, cs, csP, cll = C.kfs(ssm, y)
        pf, pP, ps, psP, pll = C.par_kfs(ssm, y, nthreads)
        assert relerr(pf, cf) < 1e-11 and relerr(pP, cP) < 1e-11
        assert relerr(ps, cs) < 1e-11 and relerr(psP, csP) < 1e-11
        assert abs(pll - cll) <= 1e-11 * max(1.0, abs(cll))
        if n <= 1201:
            sms, sPs = O.kfs(ssm, y)
            assert relerr(ps, sms) < 1e-10 and relerr(psP, sPs) < 1e-10


def test_merge_sorted_matches_stable_argsort():
    rng = np.random.default_rng(0)
    a = np.sort(rng.integers(0, 50, 40).astype(float))
    b = np.sort(rng.integers(0, 50, 13).astype(float))
    for x, y in ((a, b), (b, a)):
        c, pay, flag = O.merge_sorted(x, y, (x * 10, y * 100), (np.zeros(x.size, bool), np.ones(y.size, bool)))
        assert np.all(np.diff(c) >= 0)
        assert sorted(c.tolist()) == sorted(np.concatenate([x, y]).tolist())
        assert np.array_equal(np.sort(c[flag]), y) and np.array_equal(np.sort(c[~flag]), x)
        assert np.allclose(pay[flag], c[flag] * 100) and np.allclose(pay[~flag], c[~flag] * 10)
    # ties: the shorter array's point comes first (searchsorted side='left', model.py:43)
    c, flag = O.merge_sorted(np.array([0., 1., 2., 3.]), np.array([1., 3.]),
                             (np.zeros(4, bool), np.ones(2, bool)))
    assert flag.tolist() == [False, True, False, False, True, False]


def test_golden_fixtures_reproduce():
    """tests/golden/*.npz were written by tests/golden/make_golden.py from this oracle."""
    path = os.path.join(GOLD, "c1_matern32_n4096.npz")
    g = np.load(path)
    from pssgp.kernels import Matern32
    sde = Matern32(float(g["variance"]), float(g["lengthscale"])).get_sde()
    ll = O.ssgp_log_likelihood(sde, g["t"], g["y"], float(g["noise"]), parallel=False)
    assert abs(ll - float(g["ll_dense"])) < 1e-8 * abs(float(g["ll_dense"]))
    mean, var = O.ssgp_predict_f(sde, g["t"], g["y"], float(g["noise"]), g["tq"], parallel=False)
    assert np.max(np.abs(mean - g["mean_dense"])) < 1e-8
    assert np.max(np.abs(var - g["var_dense"])) < 1e-8
    # large_d: the sequential oracle reproduces its own stored vectors (drift guard), and its parallel-scan
    # restatement agrees with them
    g = np.load(os.path.join(GOLD, "large_d_n1024.npz"))
    from pssgp.kernels import Matern52, Periodic, SquaredExponential
    k = Periodic(SquaredExponential(1., 1.), period=1., order=1) * Matern32(1., 1.) + Matern52(1., 1.)
    ssm = O.get_ssm(k.get_sde(), g["t"], 0.1)
    fms, fPs, ll = O.kf(ssm, g["y"], True)
    assert abs(ll - float(g["c5/ll"])) < 1e-10 * abs(float(g["c5/ll"]))
    pf, pP, pll = O.pkf(ssm, g["y"], True)
    assert abs(pll - float(g["c5/ll"])) < 1e-9 * abs(float(g["c5/ll"]))
    h = ssm[3].reshape(-1)
    assert np.max(np.abs(pf @ h - g["c5/fmean"])) < 1e-9
    # d = 18: the reference's CO2 kernel at its own order
    k18 = Periodic(SquaredExponential(1., 1.), period=1., order=3) * Matern32(0.5, 5.) + Matern32(1., 2.)
    ssm = O.get_ssm(k18.get_sde(), g["t"], 0.1)
    assert ssm[1].shape[1] == 18
    ll18 = O.kf(ssm, g["y"], True)[2]
    assert abs(ll18 - float(g["co2_d18/ll"])) < 1e-9 * abs(float(g["co2_d18/ll"]))
