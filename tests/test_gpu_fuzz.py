"""Randomised parity of the row-cooperative family and the general-LTI entry points: random stable SDEs of random state
dimension (2..16), random series lengths (1..2600, ragged against the chain length), random chain lengths, random
fractions of missing observations -- filter, smoother, log-likelihood, stand-alone smoother, device predict and the
batched log-likelihood against the numpy oracle."""
import numpy as np
import pytest

from oracle import np_oracle as O
from tests.conftest import make_times, relerr, sample_series

pytestmark = pytest.mark.gpu


def _random_model(rng, d):
    from scipy.linalg import solve_continuous_lyapunov
    M = rng.standard_normal((d, d)) * 0.6
    F = -(1.0 + rng.uniform(0, 1)) * np.eye(d) + 0.5 * (M - M.T) + np.tril(rng.standard_normal((d, d)) * 0.2, -1)
    Lq = rng.standard_normal((d, max(1, d // 2)))
    P = solve_continuous_lyapunov(F, -(Lq @ Lq.T + 0.05 * np.eye(d)))
    return F, 0.5 * (P + P.T), rng.standard_normal((1, d))


def _ssm(F, P, H, t, R):
    from scipy.linalg import expm
    dts = np.diff(np.concatenate([[0.0], t]))
    Fs = np.stack([expm(dt * F) for dt in dts])
    Qs = P[None] - np.einsum("kij,jl,kml->kim", Fs, P, Fs)
    return (P, Fs, 0.5 * (Qs + np.transpose(Qs, (0, 2, 1))), H, np.array([[R]]))


@pytest.mark.parametrize("seed", range(6))
def test_random_models_sizes_and_chains(seed):
    from pssgp import _backend as B
    rng = np.random.default_rng(1000 + seed)
    ctx = B.get_context()
    try:
        for case in range(6):
            d = int(rng.integers(2, 17))
            n = int(rng.choice([1, 2, 3, 5, 9, 17, 33, 64, 100, 257, 700, 1500, 2600]))
            chunk = int(rng.choice([0, 1, 2, 3, 5, 8, 13, 32, 100]))
            F, P, H = _random_model(rng, d)
            t = make_times(n, seed=seed * 100 + case)
            ssm = _ssm(F, P, H, t, 0.2)
            y = sample_series(ssm, seed=case, nan_frac=float(rng.choice([0.0, 0.2, 0.6])) if n > 3 else 0.0)
            ctx.set_family(3)
            ctx.set_chunk(chunk)
            sms, sPs, fms, fPs, ll = B.pkfs(ssm, y, return_filtered=True, return_loglikelihood=True)
            of, oP, oll = O.kf(ssm, y, True)
            os_, osP = O.kfs(ssm, y)
            tag = f"d={d} n={n} chunk={chunk}"
            assert relerr(fms, of) < 1e-7 and relerr(fPs, oP) < 1e-7, tag
            assert relerr(sms, os_) < 1e-7 and relerr(sPs, osP) < 1e-7, tag
            assert abs(float(ll) - oll) <= 1e-8 * abs(oll) + 1e-12, tag
            s2, sP2 = B.pks(ssm, of, oP)
            assert relerr(s2, os_) < 1e-7 and relerr(sP2, osP) < 1e-7, tag
            # general-LTI entry points on the same model (they discretise on the device)
            ctx.set_chunk(0)
            assert abs(B.lti_ll(F, P, H.reshape(-1), 0.2, t, y) - oll) <= 1e-8 * abs(oll) + 1e-12, tag
            k = int(rng.choice([1, 7, 90]))
            tq = np.sort(rng.uniform(-0.2, t[-1] + 0.3, k))
            tq[tq < 0] = 0.0
            mean, var, ll3 = B.lti_predict(F, P, H.reshape(-1), 0.2, t, y, tq)
            all_t, all_y, flags = O.merge_sorted(t, tq, (y, np.full(tq.shape, np.nan)),
                                                 (np.zeros(t.shape, bool), np.ones(tq.shape, bool)))
            ms, Ps = O.kfs(_ssm(F, P, H, all_t, 0.2), all_y)
            h = H.reshape(-1)
            assert np.max(np.abs(mean - ms[flags] @ h)) < 1e-7 * max(1.0, float(np.max(np.abs(ms)))), tag
            assert np.max(np.abs(var - np.einsum("i,nij,j->n", h, Ps[flags], h))) < 1e-7 * max(1.0, float(np.max(np.abs(Ps)))), tag
            lls = B.lti_ll_batch([(F, P, h, 0.2), (0.7 * F, 1.3 * P, h, 0.4)], t, y)
            assert abs(lls[0] - oll) <= 1e-8 * abs(oll) + 1e-12, tag
            oll2 = O.kf(_ssm(0.7 * F, 1.3 * P, H, t, 0.4), y, True)[2]
            assert abs(lls[1] - oll2) <= 1e-8 * abs(oll2) + 1e-12, tag
    finally:
        ctx.set_family(0)
        ctx.set_chunk(0)


@pytest.mark.parametrize("seed", range(4))
def test_random_models_fp32_row_cooperative(seed):
    """The fp32 instantiations of the row-cooperative family (own kernels since round 2, forced here also below d = 7):
    random stable SDEs of state dimension 2..16, ragged lengths against random chain lengths (whole-record LDS staging
    with 4-, 8- and 16-byte record tails), missing observations -- filter, smoother and log-likelihood against the fp64
    oracle at the fp32 tolerance of the north star (1e-3 of the largest entry; 3e-3 asserted for the random models)."""
    from pssgp import _backend as B
    rng = np.random.default_rng(7000 + seed)
    ctx = B.get_context()
    try:
        for case in range(6):
            d = int(rng.integers(2, 17))
            n = int(rng.choice([1, 2, 5, 17, 64, 257, 700, 1500, 2600, 6000]))
            chunk = int(rng.choice([0, 1, 3, 8, 13, 32, 100]))
            F, P, H = _random_model(rng, d)
            t = make_times(n, seed=seed * 100 + case)
            ssm = _ssm(F, P, H, t, 0.2)
            y = sample_series(ssm, seed=case, nan_frac=float(rng.choice([0.0, 0.2, 0.6])) if n > 3 else 0.0)
            ssm32 = tuple(np.asarray(a, np.float32) for a in ssm)
            ctx.set_family(3)
            ctx.set_chunk(chunk)
            sms, sPs, fms, fPs, ll = B.pkfs(ssm32, y.astype(np.float32), return_filtered=True, return_loglikelihood=True)
            assert sms.dtype == np.float32
            of, oP, oll = O.kf(ssm, y, True)
            os_, osP = O.kfs(ssm, y)
            tag = f"d={d} n={n} chunk={chunk}"
            tol = 3e-3
            assert relerr(fms, of) < tol and relerr(fPs, oP) < tol, tag
            assert relerr(sms, os_) < tol and relerr(sPs, osP) < tol, tag
            assert abs(float(ll) - oll) <= 1e-3 * abs(oll) + 1e-3, tag
            s2, sP2 = B.pks(ssm32, of.astype(np.float32), oP.astype(np.float32))
            assert relerr(s2, os_) < tol and relerr(sP2, osP) < tol, tag
    finally:
        ctx.set_family(0)
        ctx.set_chunk(0)


def random_quad_case(seed, cases=6):
    """Random stable SDEs of state dimension 5..8 in float32 on the quad-cooperative kernels (family 4; pgps_qc.hip.h):
    ragged lengths against random chain lengths (sixteen chains per wave: partially filled waves, chains beyond the end,
    the whole-record LDS road and -- at odd d -- the lane-by-lane one), missing observations; filter, smoother and
    log-likelihood against the fp64 oracle, and whole series against two random segments of the same series."""
    from pssgp import _backend as B
    from tests.test_segments import run_segments_on_one_gpu
    rng = np.random.default_rng(11000 + seed)
    ctx = B.get_context()
    tags = []
    try:
        for case in range(cases):
            d = int(rng.integers(5, 9))
            n = int(rng.choice([1, 2, 15, 16, 17, 63, 65, 255, 257, 700, 1500, 2600, 6000, 20000]))
            chunk = int(rng.choice([0, 0, 1, 3, 8, 13, 32, 64]))
            F, P, H = _random_model(rng, d)
            t = make_times(n, seed=seed * 100 + case)
            ssm = _ssm(F, P, H, t, 0.2)
            y = sample_series(ssm, seed=case, nan_frac=float(rng.choice([0.0, 0.2, 0.6])) if n > 3 else 0.0)
            ssm32 = tuple(np.asarray(a, np.float32) for a in ssm)
            tag = f"seed={seed} d={d} n={n} chunk={chunk}"
            ctx.set_family(4)
            ctx.set_chunk(chunk)
            sms, sPs, fms, fPs, ll = B.pkfs(ssm32, y.astype(np.float32), return_filtered=True, return_loglikelihood=True)
            ctx.set_chunk(0)
            ctx.set_family(0)
            of, oP, oll = O.kf(ssm, y, True)
            os_, osP = O.kfs(ssm, y)
            tol = 3e-3
            assert relerr(fms, of) < tol and relerr(fPs, oP) < tol, tag
            assert relerr(sms, os_) < tol and relerr(sPs, osP) < tol, tag
            assert abs(float(ll) - oll) <= 1e-3 * abs(oll) + 1e-3, tag
            if n >= 4:
                cut = int(rng.integers(1, n))
                got, lls, _, _ = run_segments_on_one_gpu(ssm, y, [(0, cut), (cut, n)], np.float32, 4)
                assert relerr(got["sms"], os_) < 5e-3 and relerr(got["sPs"], osP) < 5e-3, tag + f" cut={cut}"
                assert all(abs(v - oll) <= 2e-3 * abs(oll) + 1e-2 for v in lls), tag + f" cut={cut}"
            tags.append(tag)
    finally:
        ctx.set_family(0)
        ctx.set_chunk(0)
    return tags


@pytest.mark.parametrize("seed", range(3))
def test_random_models_fp32_quad_cooperative(seed):
    random_quad_case(seed)


def random_segments_case(seed):
    """One random sharded series: state dimension 1..24 (all three kernel families: lane-chunk up to 6, row-cooperative
    up to 16, wave-cooperative above -- or forced), 1..6 ranks with ragged boundaries (segments down to a single step),
    fp64 or fp32, through the three pgps_seg_* phases against the unsegmented oracle."""
    from tests.test_segments import run_segments_on_one_gpu
    rng = np.random.default_rng(9000 + seed)
    d = int(rng.choice([1, 2, 3, 5, 6, 7, 9, 12, 16, 17, 20, 24]))
    dtype = np.float32 if rng.random() < 0.25 else np.float64
    family = 0
    if rng.random() < 0.3:
        family = int(rng.choice([f for f in (1, 2, 3) if (f != 1 or d <= 6) and (f != 3 or 2 <= d <= 16)]))
    world = int(rng.integers(1, 7))
    n = int(rng.choice([world, world + 3, 40, 333, 1200, 2500]))
    n = max(n, world)
    cuts = np.sort(rng.choice(np.arange(1, n), size=world - 1, replace=False)) if world > 1 else np.array([], int)
    edges = [0] + [int(c) for c in cuts] + [n]
    bounds = list(zip(edges[:-1], edges[1:]))
    if d == 1:
        F, P, H = np.array([[-1.3]]), np.array([[0.8]]), np.array([[1.0]])
    else:
        F, P, H = _random_model(rng, d)
    t = make_times(n, seed=seed)
    ssm = _ssm(F, P, H, t, 0.2)
    y = sample_series(ssm, seed=seed, nan_frac=float(rng.choice([0.0, 0.3])) if n > 3 else 0.0)
    tag = f"seed={seed} d={d} n={n} world={world} bounds={bounds} dtype={np.dtype(dtype).name} family={family}"
    got, lls, _, _ = run_segments_on_one_gpu(ssm, y, bounds, dtype, family)
    of, oP, oll = O.kf(ssm, y, True)
    os_, osP = O.kfs(ssm, y)
    tol = 1e-7 if dtype == np.float64 else 5e-3
    assert relerr(got["fms"], of) < tol and relerr(got["fPs"], oP) < tol, tag
    assert relerr(got["sms"], os_) < tol and relerr(got["sPs"], osP) < tol, tag
    for v in lls:
        assert abs(v - oll) <= (1e-8 if dtype == np.float64 else 2e-3) * abs(oll) + (1e-10 if dtype == np.float64 else 1e-2), (tag, v, oll)
    return tag


@pytest.mark.parametrize("seed", range(10))
def test_random_segments(seed):
    random_segments_case(seed)


@pytest.mark.parametrize("seed", range(6))
def test_random_adjoint_statistics(seed):
    """pgps_lti_ll_grad_f64 on random stable models of random state dimension (2..32: row-cooperative kernels up to 16, the
    wave-cooperative ones above and -- forced -- below), random lengths ragged against random chain lengths, missing
    observations: log-likelihood and the model's adjoints against the numpy reverse sweep (oracle/np_grad.py)."""
    from oracle import np_grad as G
    from pssgp import _backend as B
    rng = np.random.default_rng(7000 + seed)
    ctx = B.get_context()
    try:
        for case in range(5):
            d = int(rng.integers(2, 33))
            n = int(rng.choice([1, 2, 3, 5, 17, 33, 64, 100, 257, 700, 1500]))
            chunk = int(rng.choice([0, 1, 2, 3, 5, 8, 13, 32, 100]))
            family = int(rng.choice([0, 0, 2]))
            F, P, H = _random_model(rng, d)
            t = make_times(n, seed=seed * 100 + case)
            y = np.sin(t) + 0.3 * rng.standard_normal(n)
            if n > 3:
                y[rng.uniform(size=n) < float(rng.choice([0.0, 0.2, 0.6]))] = np.nan
            if np.all(np.isnan(y)):
                y[0] = 0.1
            ctx.set_family(family)
            ctx.set_chunk(chunk)
            dev = B.lti_ll_grad(F, P, H.reshape(-1), 0.2, t, y)
            ref = G.ll_grad_stats(F, P, H.reshape(-1), 0.2, t, y)
            tag = f"seed={seed} d={d} n={n} chunk={chunk} family={family}"
            assert abs(dev[0] - ref[0]) <= 1e-8 * abs(ref[0]) + 1e-12, tag
            for name, a, b in zip(("Abar", "Ubar", "Hbar", "Rbar"), dev[1:], ref[1:]):
                err = float(np.max(np.abs(np.asarray(a) - np.asarray(b)))) / max(1e-6, float(np.max(np.abs(b))))
                assert err < 1e-7, (tag, name, err)
    finally:
        ctx.set_family(0)
        ctx.set_chunk(0)


@pytest.mark.parametrize("seed", range(6))
def test_random_fused_path_adjoint_statistics(seed):
    """pgps_series_gp_ll_grad_adj_f64 (the adjoint pass on the fused Matern-family kernels) at random hyper-parameters,
    lengths ragged against random chunk lengths, one launch and three, any share of missing observations, duplicate time
    stamps: log-likelihood and the model's adjoints against the numpy reverse sweep (oracle/np_grad.py) on the kernel's own SDE."""
    from oracle import np_grad as G
    from pssgp import _backend as B
    from pssgp.kernels import Matern12, Matern32, Matern52
    from pssgp.model import StateSpaceGP
    rng = np.random.default_rng(9000 + seed)
    ctx = B.get_context()
    try:
        for case in range(6):
            cls = [Matern12, Matern32, Matern52][int(rng.integers(0, 3))]
            k = cls(float(rng.uniform(0.3, 3.0)), float(rng.uniform(0.1, 2.0)))
            n = int(rng.choice([1, 2, 3, 5, 17, 64, 257, 700, 2048, 2049, 5000, 9001]))
            chunk = int(rng.choice([0, 0, 1, 2, 3, 5, 8, 13]))
            one = int(rng.choice([-1, -1, 0]))
            R = float(rng.uniform(0.01, 1.0))
            t = make_times(n, seed=seed * 100 + case)
            if n > 20:
                t[7:10] = t[7]                          # dt = 0 steps
            y = np.sin(t) + 0.3 * rng.standard_normal(n)
            if n > 3:
                y[rng.uniform(size=n) < float(rng.choice([0.0, 0.2, 0.6]))] = np.nan
            if np.all(np.isnan(y)):
                y[0] = 0.1
            gp = StateSpaceGP((t[:, None], y[:, None]), k, noise_variance=R, parallel=True)
            ctx.set_chunk(chunk)
            ctx.set_one_launch(one)
            ser = gp._device_series(force=True)
            dev = ser.gp_ll_grad_adj(gp._packed_fused(gp._device_forms()[0]), R)
            sde = k.get_sde()
            ref = G.ll_grad_stats(sde.F, sde.P0, sde.H, R, t, y)
            tag = f"seed={seed} {cls.__name__} n={n} chunk={chunk} one_launch={one}"
            assert abs(dev[0] - ref[0]) <= 1e-9 * abs(ref[0]) + 1e-12, tag
            for name, a, b in zip(("Abar", "Ubar", "Hbar", "Rbar"), dev[1:], ref[1:]):
                err = float(np.max(np.abs(np.asarray(a) - np.asarray(b)))) / max(1e-6, float(np.max(np.abs(b))))
                assert err < 1e-8, (tag, name, err)
    finally:
        ctx.set_chunk(0)
        ctx.set_one_launch(-1)
