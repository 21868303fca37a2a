"""The two-rows level-1 kernels (csrc/pgps_rc2.hip.h): state dimensions 17..32 on the paddings 18, 24 and 32.

Random stable models of every padding's extreme dimensions, series lengths ragged against the chunk length (odd
numbers of chunks: the second chain of the last wave has nothing to do; a last chunk shorter than the others; a
single step), chunk lengths from 1 step, missing observations -- against the numpy oracle; the same calls through
the LDS-tile kernels the family used before (PGPS_WC_ROWS2=0, read when a context is created); float32."""
import numpy as np
import pytest

from oracle import np_oracle as O
from tests.conftest import make_times, relerr, sample_series
from tests.test_gpu_fuzz import _random_model, _ssm

pytestmark = pytest.mark.gpu


def _all(B, ssm, y):
    sms, sPs, fms, fPs, ll = B.pkfs(ssm, y, return_filtered=True, return_loglikelihood=True)
    return dict(fms=fms, fPs=fPs, sms=sms, sPs=sPs, ll=np.array([float(ll)]))


def _want(ssm, y):
    of, oP, oll = O.kf(ssm, y, True)
    os_, osP = O.kfs(ssm, y)
    return dict(fms=of, fPs=oP, sms=os_, sPs=osP, ll=np.array([oll]))


def _close(got, want, tol, tag):
    for k in ("fms", "fPs", "sms", "sPs"):
        assert relerr(got[k], want[k]) < tol, (tag, k, relerr(got[k], want[k]))
    assert abs(got["ll"][0] - want["ll"][0]) <= tol * abs(want["ll"][0]) + 1e-10, (tag, got["ll"], want["ll"])


@pytest.mark.parametrize("d", [17, 18, 19, 24, 25, 32])
def test_two_rows_random_models(d):
    from pssgp import _backend as B
    rng = np.random.default_rng(4200 + d)
    ctx = B.get_context()
    F, P, H = _random_model(rng, d)
    try:
        for n, chunk, nan_frac in [(1, 0, 0.0), (2, 1, 0.0), (3, 2, 0.0), (33, 16, 0.2), (64, 16, 0.0), (65, 16, 0.6),
                                   (97, 3, 0.2), (700, 7, 0.2), (1500, 0, 0.2), (2600, 33, 0.0)]:
            t = make_times(n, seed=7 * d + n)
            ssm = _ssm(F, P, H, t, 0.2)
            y = sample_series(ssm, seed=n, nan_frac=nan_frac)
            ctx.set_chunk(chunk)
            _close(_all(B, ssm, y), _want(ssm, y), 1e-7, f"d={d} n={n} chunk={chunk}")
            # filter only (no smoothing total in the apply kernel)
            fms, fPs, ll = B.pkf(ssm, y, return_loglikelihood=True)
            of, oP, oll = O.kf(ssm, y, True)
            assert relerr(fms, of) < 1e-7 and relerr(fPs, oP) < 1e-7 and abs(float(ll) - oll) <= 1e-7 * abs(oll) + 1e-10
    finally:
        ctx.set_chunk(0)


@pytest.mark.parametrize("d", [23, 24, 27, 29, 32])
def test_large_dimensions_long_series_against_the_c_oracle(d):
    """2^15 steps at the top of the range against oracle/kalman_seq.c -- d = 23 is the last dimension the two-rows kernels
    serve in fp64, 24 .. 32 run on the LDS-tile kernels since round 4 (their two-rows kernels needed scratch memory:
    pgps_wc_args.h, tools/scratch_gate.py; the one wrong result of this family, at the first two-rows commit, was a d = 32
    kernel with 444 B of it -- tools/d32_ghost.py reproduces it from that commit's library)."""
    from oracle import c_oracle as C
    from pssgp import _backend as B
    rng = np.random.default_rng(5100 + d)
    F, P, H = _random_model(rng, d)
    n = 1 << 15
    t = make_times(n, seed=3 * d)
    ssm = _ssm(F, P, H, t, 0.2)
    y = sample_series(ssm, seed=d, nan_frac=0.2)
    cf, cP, cs, csP, cll = C.kfs(ssm, y, np.float64)
    got = _all(B, ssm, y)
    assert relerr(got["fms"], cf) < 1e-8 and relerr(got["fPs"], cP) < 1e-8
    assert relerr(got["sms"], cs) < 1e-8 and relerr(got["sPs"], csP) < 1e-8
    assert abs(got["ll"][0] - cll) <= 1e-9 * abs(cll)
    fms, fPs, ll = B.pkf(ssm, y, return_loglikelihood=True)
    assert relerr(fms, cf) < 1e-8 and relerr(fPs, cP) < 1e-8 and abs(float(ll) - cll) <= 1e-9 * abs(cll)


@pytest.mark.parametrize("d", [18, 22, 29])
def test_two_rows_equal_the_lds_tile_kernels(d, monkeypatch):
    from pssgp import _backend as B
    rng = np.random.default_rng(77 + d)
    F, P, H = _random_model(rng, d)
    t = make_times(3001, seed=d)
    ssm = _ssm(F, P, H, t, 0.1)
    y = sample_series(ssm, seed=d, nan_frac=0.15)
    default = B.get_context()
    monkeypatch.setenv("PGPS_WC_ROWS2", "0")
    tiles = B.Context(0)
    monkeypatch.delenv("PGPS_WC_ROWS2")
    res = []
    try:
        for ctx in (default, tiles):
            monkeypatch.setitem(B._contexts, 0, ctx)
            ctx.set_chunk(5)
            res.append(_all(B, ssm, y))
    finally:
        monkeypatch.setitem(B._contexts, 0, default)
        default.set_chunk(0)
        tiles.close()
    _close(res[0], res[1], 1e-9, f"d={d} two rows vs tiles")
    _close(res[0], _want(ssm, y), 1e-7, f"d={d} two rows vs oracle")


@pytest.mark.parametrize("d", [18, 24, 31])
def test_two_rows_float32(d):
    from pssgp import _backend as B
    rng = np.random.default_rng(900 + d)
    F, P, H = _random_model(rng, d)
    t = make_times(1200, seed=d)
    ssm = _ssm(F, P, H, t, 0.2)
    y = sample_series(ssm, seed=d, nan_frac=0.1)
    ssm32 = tuple(np.asarray(a, np.float32) for a in ssm)
    got = _all(B, ssm32, y.astype(np.float32))
    assert got["sms"].dtype == np.float32
    # (since round 4 float32 smoother calls above d = 16 compute in fp64 on the float32 arrays: the north star's 1e-3)
    _close(got, _want(ssm, y), 1e-3, f"d={d} float32")
    assert B.get_context().status() & 4
    # the float32 arithmetic of the two-rows kernels themselves, on request (2e-3 on this grid: what round 3 asserted)
    ctx = B.get_context()
    ctx.set_f32_policy(1)
    try:
        native = _all(B, ssm32, y.astype(np.float32))
        assert not (ctx.status() & 4)
    finally:
        ctx.set_f32_policy(0)
    _close(native, _want(ssm, y), 2e-3, f"d={d} float32 arithmetic")
