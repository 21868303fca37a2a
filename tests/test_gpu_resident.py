"""The resident one-launch filter + smoother (csrc/pgps_resident.hip.h: fp64, d = 2, up to 4096 steps per CU) through the
C ABI against the CPU oracle and against the three-launch path: array form (pgps_pkfs_*: the reference's pkf + pks
contract, pssgp/kalman/parallel.py:121-201) and fused form (pgps_gp_*: the StateSpaceGP road, pssgp/model.py:92-117).

Tolerance: the north star's 1e-5 relative in fp64; asserted at 1e-9 (observed ~1e-15: same algebra, other bracketing)."""
import ctypes

import numpy as np
import pytest

from oracle import np_oracle as O
from oracle import c_oracle as C
from tests.conftest import make_times, relerr, sample_series, sample_series_fast

pytestmark = pytest.mark.gpu
TOL64 = 1e-9
PGPS_FAMILY_RESIDENT = 12


def _B():
    from pssgp import _backend
    return _backend


@pytest.fixture()
def ctx():
    c = _B().get_context()
    c.set_resident(1)
    yield c
    c.set_resident(-1)
    assert c.status() == 0


def _m32(ls=1.0, var=1.0):
    from pssgp.kernels import Matern32
    return Matern32(variance=var, lengthscales=ls)


def _oracle_all(ssm, y):
    fms, fPs, ll = O.kf(ssm, y, True)
    sms, sPs = O.kfs(ssm, y)
    return dict(fms=fms, fPs=fPs, sms=sms, sPs=sPs, ll=np.array([ll]))


def _array_all(ssm, y):
    sms, sPs, fms, fPs, ll = _B().pkfs(tuple(np.asarray(a, np.float64) for a in ssm), np.asarray(y, np.float64),
                                       return_filtered=True, return_loglikelihood=True)
    return dict(fms=fms, fPs=fPs, sms=sms, sPs=sPs, ll=np.array([float(ll)]))


def _fused_all(sde, t, y, r):
    B = _B()
    out = B.gp(B.nilpotent_form(sde.F), sde.P0, np.asarray(sde.H).reshape(-1), r, t, y, want_filtered=True, want_smoothed=True)
    return dict(fms=out["fms"], fPs=out["fPs"], sms=out["sms"], sPs=out["sPs"], ll=np.array([float(out["ll"])]))


def _fused_ll(sde, t, y, r, filtered=False):
    B = _B()
    out = B.gp(B.nilpotent_form(sde.F), sde.P0, np.asarray(sde.H).reshape(-1), r, t, y, want_filtered=filtered)
    return (float(out["ll"]), out["fms"], out["fPs"]) if filtered else float(out["ll"])


def _check(got, want, tol=TOL64):
    for name in want:
        e = relerr(got[name], want[name])
        assert e < tol, f"{name}: rel err {e:.3e} >= {tol}"


@pytest.mark.parametrize("n", [1, 2, 3, 15, 16, 17, 63, 64, 65, 1023, 1024, 1025, 4095, 4096, 4097, 3 * 4096 + 17, 20000])
def test_resident_array_and_fused_match_oracle_at_ragged_lengths(ctx, n):
    """Lengths around the lane (16 steps), wave (1024) and workgroup (4096) boundaries: padding steps, the series' last
    element inside a chunk / at a chunk's end / at a wave's end, one workgroup and several; 20 % missing observations."""
    sde = _m32(0.7).get_sde()
    t = make_times(n, seed=n)
    ssm = O.get_ssm(sde, t, 0.1)
    y = sample_series(ssm, seed=n, nan_frac=0.2 if n > 4 else 0.0)
    assert ctx.get_family(n, 2) == PGPS_FAMILY_RESIDENT
    want = _oracle_all(ssm, y)
    _check(_array_all(ssm, y), want)
    _check(_fused_all(sde, t, y, 0.1), want)


def test_resident_matches_the_parallel_oracle_and_the_dense_gp(ctx):
    """The reference's own equivalence pin (tests/test_gp_vs_kfs.py:45-99): log-likelihood against the dense GP, and the
    parallel restatement (oracle O3: elements + operators, tree bracketing) on every output."""
    n = 700
    sde = _m32(0.5).get_sde()
    rng = np.random.default_rng(3)
    t = np.sort(rng.uniform(0, 1, n)) * 6.0
    ssm = O.get_ssm(sde, t, 0.1)
    y = sample_series(ssm, seed=4)
    got = _array_all(ssm, y)
    fms, fPs, ll = O.pkf(ssm, y, True)
    sms, sPs = O.pks(ssm, fms, fPs)
    _check(got, dict(fms=fms, fPs=fPs, sms=sms, sPs=sPs, ll=np.array([ll])))
    ll_dense = O.dense_gp(("matern32", 1.0, 0.5), t, y, 0.1)
    assert abs(got["ll"][0] - ll_dense) < 1e-8 * abs(ll_dense)


@pytest.mark.parametrize("pattern", ["all", "first", "last", "every-other", "block"])
def test_resident_missing_observations(ctx, pattern):
    """NaN observation = pure predict, no log-likelihood term (parallel.py:46-53, 147-149) -- including the first and the
    last step of the series and a whole chunk of a lane."""
    n = 5000
    sde = _m32().get_sde()
    t = make_times(n, seed=2)
    ssm = O.get_ssm(sde, t, 0.1)
    y = sample_series(ssm, seed=2)
    if pattern == "all":
        y[:] = np.nan
    elif pattern == "first":
        y[:40] = np.nan
    elif pattern == "last":
        y[-40:] = np.nan
    elif pattern == "every-other":
        y[::2] = np.nan
    else:
        y[1000:3000] = np.nan
    want = _oracle_all(ssm, y)
    got = _array_all(ssm, y)
    if pattern == "all":
        assert got["ll"][0] == 0.0
        want.pop("ll"), got.pop("ll")
    _check(got, want)
    gotf = _fused_all(sde, t, y, 0.1)
    if pattern == "all":
        gotf.pop("ll")
    _check(gotf, want)


def test_resident_equals_three_launch_path_and_is_deterministic(ctx):
    """Same algebra, other bracketing: round-off apart (1e-12); and bit-identical from run to run (fixed geometry, fixed
    combine order, barriers instead of arrival-ordered hand-offs)."""
    n = (1 << 18) + 333
    sde = _m32().get_sde()
    t = make_times(n, seed=5)
    ssm = O.get_ssm(sde, t, 0.1)
    y = sample_series_fast(ssm, seed=5, nan_frac=0.1)
    a = _array_all(ssm, y)
    b = _array_all(ssm, y)
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    ctx.set_resident(0)
    assert ctx.get_family(n, 2) != PGPS_FAMILY_RESIDENT
    c = _array_all(ssm, y)
    ctx.set_resident(1)
    _check(a, c, 1e-12)


def test_resident_full_size_against_c_oracle(ctx):
    """Config c2 itself: 2^20 steps = 256 workgroups, every CU of the chip, against the C restatement of sequential.py."""
    n = 1 << 20
    sde = _m32().get_sde()
    t = make_times(n, seed=11)
    ssm = tuple(np.asarray(a, np.float64) for a in O.get_ssm(sde, t, 0.1))
    y = sample_series_fast(ssm, seed=11, nan_frac=0.05)
    assert ctx.get_family(n, 2) == PGPS_FAMILY_RESIDENT
    got = _array_all(ssm, y)
    fms, fPs, sms, sPs, ll = C.kfs(ssm, y)
    _check(got, dict(fms=fms, fPs=fPs, sms=sms, sPs=sPs, ll=np.array([ll])))
    gotf = _fused_all(sde, t, y, 0.1)
    _check(gotf, dict(fms=fms, fPs=fPs, sms=sms, sPs=sPs, ll=np.array([ll])))


def test_resident_automatic_choice_and_its_limits(ctx):
    """Automatic from 2^17 steps, never beyond 4096 steps per CU, never for other dimensions / float32 / pinned geometries /
    filter-only or smoother-only calls; a misaligned observation array takes the three launches and still answers."""
    B = _B()
    ctx.set_resident(-1)
    assert ctx.get_family((1 << 17) - 1, 2) != PGPS_FAMILY_RESIDENT
    assert ctx.get_family(1 << 17, 2) == PGPS_FAMILY_RESIDENT
    assert ctx.get_family(1 << 20, 2) == PGPS_FAMILY_RESIDENT
    assert ctx.get_family((1 << 20) + 1, 2) != PGPS_FAMILY_RESIDENT           # 257 workgroups do not fit 256 CUs
    assert ctx.get_family(1 << 19, 3) != PGPS_FAMILY_RESIDENT
    assert ctx.get_family(1 << 19, 2, f32=True) != PGPS_FAMILY_RESIDENT
    assert ctx.get_family(1 << 19, 2, what=0) == PGPS_FAMILY_RESIDENT         # pkf: the filter-only form of the launch
    assert ctx.get_family(1 << 19, 2, what=1) != PGPS_FAMILY_RESIDENT         # pks
    ctx.set_chunk(8)
    try:
        assert ctx.get_family(1 << 19, 2) != PGPS_FAMILY_RESIDENT
    finally:
        ctx.set_chunk(0)
    ctx.set_resident(1)
    # misaligned ys (8 bytes off a 16-byte boundary) on device pointers
    n = 6000
    sde = _m32().get_sde()
    t = make_times(n, seed=8)
    ssm = tuple(np.asarray(a, np.float64) for a in O.get_ssm(sde, t, 0.1))
    y = sample_series(ssm, seed=8, nan_frac=0.1)
    P0, Fs, Qs, H, R = ssm
    bufs = {}
    for name, arr in (("P0", P0), ("Fs", Fs), ("Qs", Qs), ("H", np.asarray(H, np.float64).reshape(-1)), ("ys", np.concatenate([[0.0], y]))):
        arr = np.ascontiguousarray(arr, np.float64)
        bufs[name] = ctx.malloc(arr.nbytes)
        ctx.h2d(bufs[name], arr)
    outs = {k: ctx.malloc(n * m * 8) for k, m in (("fms", 2), ("fPs", 4), ("sms", 2), ("sPs", 4))}
    ll = ctx.malloc(8)
    try:
        ctx.call("pgps_pkfs_dev_f64", ctypes.c_long(n), ctypes.c_int(2), bufs["P0"], bufs["Fs"], bufs["Qs"], bufs["H"],
                 ctypes.c_double(float(np.asarray(R).reshape(-1)[0])), ctypes.c_void_p(bufs["ys"] + 8), outs["fms"], outs["fPs"], outs["sms"], outs["sPs"], ll)
        sms = np.empty((n, 2))
        ctx.d2h(sms, outs["sms"])
        sm_o, _ = O.kfs(ssm, y)
        assert relerr(sms, sm_o) < TOL64
    finally:
        for p in list(bufs.values()) + list(outs.values()) + [ll]:
            ctx.free(p)
    assert B is not None


def test_resident_hand_offs_under_back_to_back_launches(ctx):
    """Forty launches back to back on device-resident arrays whose inputs change every launch (consumers' caches warm with the
    previous launch's totals): every launch must reproduce the three-launch path's log-likelihood and smoothed means -- a
    stale total in either grid-wide hand-off would show as an O(1) error in the workgroups behind it."""
    n = 1 << 19
    sde = _m32().get_sde()
    t = make_times(n, seed=21)
    ssm = tuple(np.asarray(a, np.float64) for a in O.get_ssm(sde, t, 0.1))
    P0, Fs, Qs, H, R = ssm
    dev = {}
    for name, arr in (("P0", P0), ("Fs", Fs), ("Qs", Qs), ("H", np.asarray(H, np.float64).reshape(-1))):
        arr = np.ascontiguousarray(arr, np.float64)
        dev[name] = ctx.malloc(arr.nbytes)
        ctx.h2d(dev[name], arr)
    dev["ys"] = ctx.malloc(n * 8)
    outs = {k: ctx.malloc(n * m * 8) for k, m in (("fms", 2), ("fPs", 4), ("sms", 2), ("sPs", 4))}
    ll_d = ctx.malloc(8)
    rng = np.random.default_rng(0)
    base = sample_series_fast(ssm, seed=21)

    def run():
        ctx.call("pgps_pkfs_dev_f64", ctypes.c_long(n), ctypes.c_int(2), dev["P0"], dev["Fs"], dev["Qs"], dev["H"],
                 ctypes.c_double(float(np.asarray(R).reshape(-1)[0])), dev["ys"], outs["fms"], outs["fPs"], outs["sms"], outs["sPs"], ll_d)
        sms = np.empty((n, 2))
        llv = np.empty(1)
        ctx.d2h(sms, outs["sms"])
        ctx.d2h(llv, ll_d)
        return sms, llv[0]

    try:
        for it in range(40):
            y = base * (1.0 + 0.5 * it) + rng.standard_normal(n) * 0.01
            ctx.h2d(dev["ys"], np.ascontiguousarray(y))
            ctx.set_resident(1)
            sm_r, ll_r = run()
            if it % 8 == 0:
                ctx.set_resident(0)
                sm_3, ll_3 = run()
                ctx.set_resident(1)
                assert relerr(sm_r, sm_3) < 1e-11 and abs(ll_r - ll_3) < 1e-11 * abs(ll_3), it
            else:
                assert np.isfinite(ll_r) and np.all(np.isfinite(sm_r[::4097]))
    finally:
        for p in list(dev.values()) + list(outs.values()) + [ll_d]:
            ctx.free(p)


def test_statespacegp_predict_takes_the_resident_pass_unchanged(ctx):
    """The model-level API on top (pssgp/model.py:92-117): same posterior whichever road the filter + smoother take."""
    from pssgp.model import StateSpaceGP
    rng = np.random.default_rng(5)
    n, k = 3000, 200
    t = np.cumsum(0.05 * rng.uniform(0.5, 1.5, n))
    y = np.sin(t) + 0.3 * rng.standard_normal(n)
    tq = np.sort(rng.uniform(t[0], t[-1], k))
    kern = _m32()
    model = StateSpaceGP((t[:, None], y[:, None]), kern, noise_variance=0.1, parallel=True)
    mean, var = model.predict_f(tq[:, None])
    mean_o, var_o = O.ssgp_predict_f(kern.get_sde(), t, y, 0.1, tq, parallel=False)
    assert np.max(np.abs(mean[:, 0] - mean_o)) < 1e-8 and np.max(np.abs(var[:, 0] - var_o)) < 1e-8


# ------------------------------------------------------------------------------------------------
# the forgetting shortcut for the carry across workgroups (csrc/pgps_kernels.hip.h, pgps_set_shortcut)
# ------------------------------------------------------------------------------------------------
def _both_roads(ctx, ssm, y, resident):
    ctx.set_resident(1 if resident else 0)
    out = {}
    for on in (1, 0):
        ctx.set_shortcut(on)
        out[on] = _array_all(ssm, y)
    ctx.set_shortcut(1)
    return out[1], out[0]


@pytest.mark.parametrize("resident", [True, False])
def test_forgetting_shortcut_gives_the_same_bits_where_it_applies(ctx, resident):
    """An ordinary model (Matern-3/2, observations at every step): every workgroup total has |A| ~ 1e-150 and below, the
    shortcut is taken -- and the result is bit for bit the general fold's (0 * w + b = b in any bracketing)."""
    n = (1 << 19) + 77              # (the three-launch kernels try the shortcut from 2048 steps per workgroup: 2^19 steps here)
    sde = _m32().get_sde()
    t = make_times(n, seed=31)
    ssm = tuple(np.asarray(a, np.float64) for a in O.get_ssm(sde, t, 0.1))
    y = sample_series_fast(ssm, seed=31, nan_frac=0.1)
    fast, slow = _both_roads(ctx, ssm, y, resident)
    for k in fast:
        assert np.array_equal(fast[k], slow[k]), k
    fms, fPs, sms, sPs, ll = C.kfs(ssm, y)
    _check(fast, dict(fms=fms, fPs=fPs, sms=sms, sPs=sPs, ll=np.array([ll])))


@pytest.mark.parametrize("resident", [True, False])
def test_forgetting_shortcut_steps_aside_where_the_filter_remembers(ctx, resident):
    """A model that does NOT forget within a workgroup's span: lengthscale 2e4 (F within 1e-5 of the identity per step) and
    long stretches without observations, some of them covering whole workgroups -- there the totals' A is O(1), the test
    fails and the general fold runs; where observations return the shortcut applies again.  Both settings against the C
    oracle, and against each other to round-off."""
    n = 1 << 19
    sde = _m32(ls=2.0e4).get_sde()
    t = make_times(n, seed=32)
    ssm = tuple(np.asarray(a, np.float64) for a in O.get_ssm(sde, t, 0.1))
    y = sample_series_fast(ssm, seed=32)
    y[3000:180000] = np.nan
    y[280000:280010] = np.nan
    y[360000:] = np.nan
    fast, slow = _both_roads(ctx, ssm, y, resident)
    fms, fPs, sms, sPs, ll = C.kfs(ssm, y)
    want = dict(fms=fms, fPs=fPs, sms=sms, sPs=sPs, ll=np.array([ll]))
    _check(fast, want, 1e-8)
    _check(slow, want, 1e-8)
    _check(fast, slow, 1e-12)


def test_forgetting_shortcut_on_the_fused_road_and_in_float32(ctx):
    """pgps_gp_* (three launches, 2^21 steps: beyond the resident launch) and a float32 series on the lane-chunk kernels."""
    B = _B()
    n = 1 << 21
    sde = _m32().get_sde()
    t = make_times(n, seed=33)
    y = np.sin(0.3 * t) + 0.3 * np.random.default_rng(33).standard_normal(n)
    form = B.nilpotent_form(sde.F)
    res = {}
    for on in (1, 0):
        ctx.set_shortcut(on)
        res[on] = B.gp(form, sde.P0, np.asarray(sde.H).reshape(-1), 0.1, t, y, want_filtered=True, want_smoothed=True)
    ctx.set_shortcut(1)
    for k in ("fms", "fPs", "sms", "sPs"):
        assert np.array_equal(res[1][k], res[0][k]), k
    assert float(res[1]["ll"]) == float(res[0]["ll"])
    n = 1 << 17
    ssm = tuple(np.asarray(a, np.float32) for a in O.get_ssm(sde, t[:n], 0.1))
    y32 = y[:n].astype(np.float32)
    ctx.set_f32_policy(1)
    try:
        out = {}
        for on in (1, 0):
            ctx.set_shortcut(on)
            out[on] = B.pkfs(ssm, y32, return_filtered=True, return_loglikelihood=True)
        ctx.set_shortcut(1)
    finally:
        ctx.set_f32_policy(0)
    for a_, b_ in zip(out[1][:4], out[0][:4]):
        assert np.array_equal(a_, b_)


def test_resident_launch_is_not_taken_under_stream_capture(ctx):
    """The launch's barrier set and hand-off epoch are per-launch host state: a call made while its stream is captured into a
    graph takes the three launches (a replay would present the same epoch again)."""
    import torch
    dev = torch.device("cuda:0")
    s = torch.cuda.Stream(device=dev)
    ctx.set_resident(1)
    ctx.set_stream(s.cuda_stream)
    try:
        assert ctx.get_family(1 << 19, 2) == PGPS_FAMILY_RESIDENT
        g = torch.cuda.CUDAGraph()
        x = torch.zeros(16, device=dev)
        with torch.cuda.graph(g, stream=s):
            inside = ctx.get_family(1 << 19, 2)
            x += 1.0
        assert inside != PGPS_FAMILY_RESIDENT
        assert ctx.get_family(1 << 19, 2) == PGPS_FAMILY_RESIDENT
    finally:
        torch.cuda.synchronize()
        ctx.use_own_stream()


@pytest.mark.parametrize("chunk,n", [(8, 4096 + 5), (16, 4096 + 5), (8, (1 << 17) + 123), (16, (1 << 17) + 123), (0, 1 << 18), (8, 1 << 19)])
def test_resident_launch_with_eight_and_sixteen_steps_per_lane(ctx, chunk, n):
    """The launch exists with 16 steps per lane and -- for series that then still fit the chip -- with 8 (twice the
    workgroups); chunk 8 / 16 under mode 1 pins one of them, 0 lets the library choose.  Every output against the three
    launches on the same inputs."""
    B = _B()
    sde = _m32(ls=0.7).get_sde()
    t = make_times(n, seed=n % 1000)
    Fs, Qs = B.discretise(sde.F, sde.P0, t, 0.0)
    ssm = (sde.P0, Fs, Qs, sde.H, np.array([[0.1]]))
    y = sample_series_fast(ssm, seed=3, nan_frac=0.1)
    ctx.set_resident(1)
    ctx.set_chunk(chunk)
    try:
        assert ctx.get_family(n, 2) == PGPS_FAMILY_RESIDENT
        got = B.pkfs(ssm, y, return_filtered=True, return_loglikelihood=True)
    finally:
        ctx.set_chunk(0)
    ctx.set_resident(0)
    want = B.pkfs(ssm, y, return_filtered=True, return_loglikelihood=True)
    for g, w in zip(got, want):
        assert relerr(g, w) < 1e-11


@pytest.mark.parametrize("n,chunk,seed", [(1 << 18, 8, 1), ((1 << 19) - 7, 16, 2), ((1 << 19) + 4097, 0, 3), (1 << 20, 0, 4)])
def test_resident_hand_offs_when_some_workgroups_remember_and_others_forget(ctx, n, chunk, seed):
    """A time grid with stretches a million times denser than the rest: the workgroups inside them hand on totals that have
    NOT forgotten their past (general fold behind the grid-wide wait), the others take their carry from one neighbour -- both
    roads and both kinds of wait inside one launch, twice back to back (another epoch), against the three launches with the
    shortcut off (tools/res_stress.py runs many more of these)."""
    B = _B()
    rng = np.random.default_rng(seed)
    sde = _m32(ls=1.0).get_sde()
    dt = 0.05 * rng.uniform(0.5, 1.5, n)
    for _ in range(3):
        a = int(rng.integers(0, n))
        dt[a:a + int(rng.integers(3000, 30000))] *= 1e-6
    t = np.cumsum(dt)
    Fs, Qs = B.discretise(sde.F, sde.P0, t, 0.0)
    y = np.sin(0.7 * t) + 0.3 * rng.standard_normal(n)
    y[n // 3: n // 3 + 5000] = np.nan
    ssm = (sde.P0, Fs, Qs, sde.H, np.array([[0.1]]))
    ctx.set_resident(0)
    ctx.set_shortcut(0)
    try:
        want = B.pkfs(ssm, y, return_filtered=True, return_loglikelihood=True)
    finally:
        ctx.set_shortcut(1)
    ctx.set_resident(1)
    ctx.set_chunk(chunk)
    try:
        assert ctx.get_family(n, 2) == PGPS_FAMILY_RESIDENT
        got = B.pkfs(ssm, y, return_filtered=True, return_loglikelihood=True)
        again = B.pkfs(ssm, y, return_filtered=True, return_loglikelihood=True)
    finally:
        ctx.set_chunk(0)
    for g, w, g2 in zip(got, want, again):
        assert relerr(g, w) < 1e-9
        assert np.array_equal(np.asarray(g), np.asarray(g2), equal_nan=True)


@pytest.mark.parametrize("n", [1, 2, 63, 4096, 4097, (1 << 16) + 5, (1 << 19) - 3, (1 << 19) + 4097, 1 << 20])
def test_resident_filter_alone_matches_the_three_launches(ctx, n):
    """pkf (filtered moments + log-likelihood) and the fused log-likelihood through the filter-only form of the resident launch
    (phases 1 and 2, one hand-off) against the three-launch kernels, and at small sizes against the oracle."""
    B = _B()
    sde = _m32(ls=0.8).get_sde()
    t = make_times(n, seed=n % 997)
    Fs, Qs = B.discretise(sde.F, sde.P0, t, 0.0)
    ssm = (sde.P0, Fs, Qs, sde.H, np.array([[0.1]]))
    y = sample_series_fast(ssm, seed=7, nan_frac=0.1 if n > 8 else 0.0)
    ctx.set_resident(1)
    assert ctx.get_family(n, 2, what=0) == PGPS_FAMILY_RESIDENT
    fm, fP, ll = B.pkf(ssm, y, return_loglikelihood=True)
    ll_fused = _fused_ll(sde, t, y, 0.1)
    llf2, ffm, ffP = _fused_ll(sde, t, y, 0.1, filtered=True)
    assert llf2 == ll_fused and relerr(ffm, fm) < 1e-9 and relerr(ffP, fP) < 1e-9
    ctx.set_resident(0)
    fm0, fP0, ll0 = B.pkf(ssm, y, return_loglikelihood=True)
    ll_fused0 = _fused_ll(sde, t, y, 0.1)
    assert relerr(fm, fm0) < 1e-11 and relerr(fP, fP0) < 1e-11
    assert abs(float(ll) - float(ll0)) <= 1e-11 * max(1.0, abs(float(ll0)))
    assert abs(ll_fused - ll_fused0) <= 1e-11 * max(1.0, abs(ll_fused0))
    assert abs(ll_fused - float(ll)) <= 1e-9 * max(1.0, abs(float(ll)))
    if n <= 4097:
        of, oP, oll = O.kf(ssm, y, True)
        assert relerr(fm, of) < 1e-10 and relerr(fP, oP) < 1e-10 and abs(float(ll) - oll) <= 1e-10 * max(1.0, abs(oll))
