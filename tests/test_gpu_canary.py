"""Canary for the kernels that pass operands between lanes (row-cooperative d = 2..16 in both types, quad-cooperative
d = 5..8, two-rows d = 17..31, LDS-tile up to 32, the lane-chunk scans and the resident launch): series whose chain totals
are known in closed form, so that a lane which comes out of a cross-lane product with a wrong row shows as an exact
mismatch at a named row -- the failure mode of the round-3 `rc2_reduce1<double, 32>` (three lanes of a chain returned a
zero row of F C F^T + Q, profiles/r04_experiments.txt item 1; every kernel with scratch memory and cross-lane reads is
refused at build time since, tools/scratch_gate.py), which the parity tests only sample at a few dimensions.

    identity   F = I, Q = 0, nothing observed:   fm = 0, fP = P0, sm = 0, sP = P0  EXACTLY, at every step
    noise      F = I, Q = q I, nothing observed: fP = P0 + k q I (the copy of Q into C, the smoother's gain E = P Pp^-1)
    dense      F = I + eps M, Q small, observed through a random H: against the numpy oracle (every product, the rank-one
               updates, J and eta of the chain totals, the smoothing elements)

Reference semantics: pssgp/kalman/parallel.py:13-72 (elements), 100-118 / 176-184 (operators), 121-152 / 187-201 (pkf, pks)."""
import numpy as np
import pytest

from oracle import np_oracle as O
from tests.conftest import relerr

pytestmark = pytest.mark.gpu


def _B():
    from pssgp import _backend
    return _backend


def _cases(d, n, dtype):
    rng = np.random.default_rng(100 + d)
    L = rng.standard_normal((d, d)) * 0.3
    P0 = L @ L.T + np.diag(np.arange(1, d + 1, dtype=float))
    if dtype == np.float32:                 # exactly representable priors keep the identity case exact in float32 too
        P0 = np.round(P0 * 64.0) / 64.0
        P0 = 0.5 * (P0 + P0.T)
    eye = np.eye(d)
    e0 = np.zeros((1, d))
    e0[0, 0] = 1.0
    hr = rng.standard_normal((1, d))
    M = rng.standard_normal((d, d)) * (0.05 / np.sqrt(d))
    fdense = np.stack([0.97 * eye + M * (0.5 + 0.1 * (k % 5)) for k in range(n)])          # contractive: the states stay O(1)
    fdense[0] = eye
    nan = np.full(n, np.nan)
    yobs = rng.standard_normal(n)
    ident = np.broadcast_to(eye, (n, d, d)).copy()
    zero = np.zeros((n, d, d))
    # step 0 carries F = I, Q = 0: the parallel path takes the prior as it is at the first element (reference parallel.py:23-29,
    # the sequential oracle predicts first) -- with a trivial first step the two agree and the sequential oracle can judge
    qnoise = np.broadcast_to(0.25 * eye, (n, d, d)).copy()
    qdense = np.broadcast_to(0.05 * eye, (n, d, d)).copy()
    qnoise[0] = 0.0
    qdense[0] = 0.0
    return P0, [("identity", ident, zero, e0, nan), ("noise", ident, qnoise, e0, nan), ("dense", fdense, qdense, hr, yobs)]


def _rows_wrong(got, want, d, tol):
    """Rows (= owning lanes of the cooperative layouts) of the first wrong step, for the failure message."""
    got = np.asarray(got, float).reshape(len(got), -1)
    want = np.asarray(want, float).reshape(len(want), -1)
    bad = np.abs(got - want) > tol * (np.abs(want).max() + 1e-300)
    steps = np.flatnonzero(bad.any(axis=1))
    if steps.size == 0:
        return None
    k = int(steps[0])
    ent = np.flatnonzero(bad[k])
    rows = sorted(set((ent // d).tolist())) if got.shape[1] == d * d else ent.tolist()
    return f"{steps.size} steps wrong, first {k}: rows / entries {rows}, worst {np.abs(got[k] - want[k]).max():.3e}"


def _run(ctx, family, dtype, d, n, chunk):
    B = _B()
    P0, cases = _cases(d, n, dtype)
    # (round-off of the wider operands: the parity tests of the cooperative families use the same two levels)
    tol = (1e-9 if d <= 5 else 1e-7) if dtype == np.float64 else 2e-3
    ctx.set_family(family)
    ctx.set_chunk(chunk)
    ctx.set_resident(1 if family == 0 else 0)
    try:
        for name, Fs, Qs, H, y in cases:
            ssm = (P0, Fs, Qs, H, np.array([[0.3]]))
            sms, sPs, fms, fPs, ll = B.pkfs(tuple(np.asarray(a, dtype) for a in ssm), np.asarray(y, dtype), return_filtered=True,
                                            return_loglikelihood=True)
            if name == "identity":
                want_P = np.broadcast_to(np.asarray(P0, dtype), (n, d, d))
                for label, got, want in (("fm", fms, np.zeros((n, d))), ("fP", fPs, want_P), ("sm", sms, np.zeros((n, d)))):
                    assert np.array_equal(np.asarray(got), want.astype(dtype)), \
                        f"{label} [{name}, d={d}, {np.dtype(dtype).name}, family {family}]: {_rows_wrong(got, want, d, 0.0)}"
                # (the smoothed covariance goes through a solve with P P^-1: rounding, not exactness)
                assert relerr(sPs, want_P) < tol, f"sP [{name}, d={d}]: {_rows_wrong(sPs, want_P, d, tol)}"
                assert float(ll) == 0.0
                continue
            of, oP, oll = O.kf(ssm, y, True)
            os_, osP = O.kfs(ssm, y)
            for label, got, want in (("fm", fms, of), ("fP", fPs, oP), ("sm", sms, os_), ("sP", sPs, osP)):
                scale = max(1.0, float(np.abs(want).max()))
                err = float(np.abs(np.asarray(got, float) - want).max()) / scale
                assert err < tol, f"{label} [{name}, d={d}, {np.dtype(dtype).name}, family {family}]: {_rows_wrong(got, want, d, tol)}"
            assert abs(float(ll) - oll) <= tol * max(1.0, abs(oll))
    finally:
        ctx.set_family(0)
        ctx.set_chunk(0)
        ctx.set_resident(-1)
        assert ctx.status() in (0, 4)           # (4: a float32 call ran in fp64 arithmetic -- allowed, reported)


ROW = [(3, dt, d) for dt in (np.float64, np.float32) for d in range(2, 17)]
QUAD = [(4, np.float32, d) for d in range(5, 9)]
WIDE = [(2, dt, d) for dt in (np.float64, np.float32) for d in range(17, 33)]       # two-rows where they exist, LDS tiles above
LANE = [(1, dt, d) for dt in (np.float64, np.float32) for d in range(1, 7)]


@pytest.mark.parametrize("family,dtype,d", ROW + QUAD + WIDE + LANE,
                         ids=lambda v: v.__name__ if isinstance(v, type) else str(v))
def test_closed_form_chains_through_every_cross_lane_instantiation(family, dtype, d):
    """Several chains per launch (n = 200 with the library's chunking: the chain totals, their scan, the carried states and
    the smoothing totals are all exercised), float32 under policy 1 (native arithmetic: the promoted road is the fp64 row)."""
    ctx = _B().get_context()
    ctx.set_f32_policy(1)
    try:
        _run(ctx, family, dtype, d, 200, 0)
    finally:
        ctx.set_f32_policy(0)


@pytest.mark.parametrize("family,dtype,d,chunk", [(3, np.float64, 11, 7), (3, np.float32, 15, 5), (4, np.float32, 6, 3),
                                                  (2, np.float64, 18, 9), (2, np.float64, 23, 16), (2, np.float32, 31, 4),
                                                  (2, np.float64, 32, 16), (1, np.float64, 2, 3), (0, np.float64, 2, 0)],
                         ids=lambda v: v.__name__ if isinstance(v, type) else str(v))
def test_closed_form_chains_long_series_and_forced_chunks(family, dtype, d, chunk):
    """Longer series (many chains, several scan levels) and chunk lengths that leave ragged chains; family 0 at d = 2 is the
    resident launch."""
    ctx = _B().get_context()
    ctx.set_f32_policy(1)
    try:
        _run(ctx, family, dtype, d, 3000 if d <= 16 else 700, chunk)
    finally:
        ctx.set_f32_policy(0)
