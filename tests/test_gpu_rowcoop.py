"""GPU parity of the row-cooperative family (csrc/pgps_rc.hip.h: a 16-lane DPP row owns a chain of steps,
lane j holds column j of every operand, products are v_fmac_f64_dpp row_newbcast chains; fp64, state dims up
to 16) against the CPU oracle.  It is the automatic choice for 6 < d <= 16 in fp64; here it is also forced at
small d, where the lane-chunk family and the dense GP give further opinions."""
import numpy as np
import pytest

from oracle import np_oracle as O
from tests.conftest import make_times, relerr, sample_series

pytestmark = pytest.mark.gpu
TOL64 = 1e-9


def _kernels():
    from pssgp.kernels import Matern32, Matern52, RBF, Periodic, SquaredExponential
    return {
        "rbf7": lambda: RBF(variance=1., lengthscales=0.7, order=7, balancing_iter=10),                       # d = 7
        "rbf8": lambda: RBF(variance=1., lengthscales=0.7, order=8, balancing_iter=10),                       # d = 8
        "rbf9": lambda: RBF(variance=1., lengthscales=0.7, order=9, balancing_iter=10),                       # d = 9
        "c5_qp_m52": lambda: Periodic(SquaredExponential(1., 1.), period=1., order=1) * Matern32(1., 1.) +
        Matern52(1., 1.),                                                                                     # d = 11
        "periodic5": lambda: Periodic(SquaredExponential(1., 0.5), period=0.5, order=5),                      # d = 12
        "rbf13": lambda: RBF(variance=1., lengthscales=0.6, order=13, balancing_iter=10),                     # d = 13
        "periodic7": lambda: Periodic(SquaredExponential(1., 0.5), period=0.5, order=7),                      # d = 16
    }


def _oracle_all(ssm, y):
    fms, fPs, ll = O.kf(ssm, y, True)
    sms, sPs = O.kfs(ssm, y)
    return dict(fms=fms, fPs=fPs, sms=sms, sPs=sPs, ll=np.array([ll]))


def _gpu_all(ssm, y):
    from pssgp import _backend as B
    ssm_t = tuple(np.asarray(a, dtype=np.float64) for a in ssm)
    sms, sPs, fms, fPs, ll = B.pkfs(ssm_t, np.asarray(y, np.float64), return_filtered=True, return_loglikelihood=True)
    return dict(fms=fms, fPs=fPs, sms=sms, sPs=sPs, ll=np.array([float(ll)]))


def _check(got, want, tol):
    for name in want:
        e = relerr(got[name], want[name])
        assert e < tol, f"{name}: rel err {e:.3e} >= {tol}"


@pytest.fixture
def row_family():
    from pssgp import _backend as B
    ctx = B.get_context()
    ctx.set_family(3)
    yield ctx
    ctx.set_family(0)
    ctx.set_chunk(0)


@pytest.mark.parametrize("idx", range(1, 7))         # d = 2, 3, 6, 6, 5, 6 (the family starts at d = 2)
def test_forced_rowcoop_small_d(row_family, kernel_zoo, idx):
    name, make, _, _ = kernel_zoo[idx]
    t = make_times(1100, seed=idx)
    ssm = O.get_ssm(make().get_sde(), t, 0.1)
    y = sample_series(ssm, seed=idx, nan_frac=0.2)
    _check(_gpu_all(ssm, y), _oracle_all(ssm, y), TOL64)


@pytest.mark.parametrize("n,lw", [(1, 8), (2, 8), (7, 8), (8, 8), (9, 8), (31, 8), (33, 8), (2049, 32), (2200, 7),
                                  (4097, 1), (70000, 16), (70001, 0)])
def test_rowcoop_ragged_lengths_and_levels(row_family, n, lw):
    """Chain boundaries, partially filled waves (four chains each), non-power-of-two chain counts in the
    Kogge-Stone levels, steps beyond the end of the series inside the last chain."""
    from pssgp.kernels import Matern52
    row_family.set_chunk(lw)
    t = make_times(n, seed=n % 97)
    ssm = O.get_ssm(Matern52(1., 1.).get_sde(), t, 0.1)
    y = sample_series(ssm, seed=3, nan_frac=0.1 if n > 4 else 0.0)
    from oracle import c_oracle as C
    cf, cP, cs, csP, cll = C.kfs(ssm, y)
    _check(_gpu_all(ssm, y), dict(fms=cf, fPs=cP, sms=cs, sPs=csP, ll=np.array([cll])), TOL64)


@pytest.mark.parametrize("name", ["rbf7", "rbf8", "rbf9", "c5_qp_m52", "periodic5", "rbf13", "periodic7"])
def test_rowcoop_state_dims(name):
    """d = 7 .. 16 through the automatic dispatch (one kernel instantiation per d)."""
    from pssgp.kalman.parallel import pkf, pkfs
    sde = _kernels()[name]().get_sde()
    n = 1500
    t = make_times(n, seed=5)
    ssm = O.get_ssm(sde, t, 0.1)
    y = sample_series(ssm, seed=5, nan_frac=0.2)
    want = _oracle_all(ssm, y)
    tol = 1e-7      # badly conditioned high-order RBF / tiny-variance periodic harmonics: the model's scale
    _check(_gpu_all(ssm, y), want, tol)
    fms, fPs, ll = pkf(ssm, y[:, None], return_loglikelihood=True)      # filter-only launch sequence
    assert relerr(fms, want["fms"]) < tol and relerr(fPs, want["fPs"]) < tol
    assert abs(float(ll) - want["ll"][0]) < tol * abs(want["ll"][0])
    sms, sPs = pkfs(ssm, y[:, None])
    assert relerr(sms, want["sms"]) < tol and relerr(sPs, want["sPs"]) < tol
    # discretisation of the same model on the GPU vs the reference's matrix-fraction formula
    from pssgp import _backend as B
    gFs, gQs = B.discretise(sde.F, sde.P0, t, 0.0)
    assert np.max(np.abs(gFs - ssm[1])) < 1e-10 * max(1.0, float(np.max(np.abs(ssm[1]))))
    assert np.max(np.abs(gQs - ssm[2])) < 1e-10 * max(1.0, float(np.max(np.abs(ssm[0]))))


@pytest.mark.parametrize("name", ["rbf7", "c5_qp_m52", "periodic7"])
def test_rowcoop_discretise_mixed_step_sizes(name):
    """Steps from 1e-4 to 40 time units in one series: zero to several squarings, different in the four rows of
    a wave; a repeated time stamp (dt = 0: F = I, Q = 0); ragged length."""
    from pssgp import _backend as B
    sde = _kernels()[name]().get_sde()
    rng = np.random.default_rng(12)
    n = 1003
    dt = 10.0 ** rng.uniform(-4, 1.6, n)
    dt[17] = 0.0
    t = np.cumsum(dt)
    ssm = O.get_ssm(sde, t, 0.1)
    gFs, gQs = B.discretise(sde.F, sde.P0, t, 0.0)
    scale = max(1.0, float(np.max(np.abs(ssm[0]))))
    assert np.max(np.abs(gFs - ssm[1])) < 1e-9 * max(1.0, float(np.max(np.abs(ssm[1]))))
    # the reference's matrix-fraction Q (kernels/base.py:39-46) exponentiates -F^T as well and loses all digits
    # once dt * |F| is large; it is the yardstick for the short steps, P - F P F^T from the oracle's F for all
    short = dt < 0.3
    assert np.max(np.abs(gQs[short] - ssm[2][short])) < 1e-9 * scale
    P = np.asarray(ssm[0])
    q_stable = P[None] - np.einsum("kij,jl,kml->kim", ssm[1], P, ssm[1])
    assert np.max(np.abs(gQs - q_stable)) < 1e-9 * scale
    assert np.max(np.abs(gFs[17] - np.eye(gFs.shape[1]))) < 1e-15 and np.max(np.abs(gQs[17])) < 1e-14 * scale


def test_rowcoop_first_observation_missing_and_all_missing(row_family):
    from pssgp.kernels import Matern52
    t = make_times(300, seed=11)
    ssm = O.get_ssm(Matern52(1., 1.).get_sde(), t, 0.1)
    y = sample_series(ssm, seed=4, nan_frac=0.3)
    y[0] = np.nan
    _check(_gpu_all(ssm, y), _oracle_all(ssm, y), TOL64)
    y[:] = np.nan
    got, want = _gpu_all(ssm, y), _oracle_all(ssm, y)
    assert got["ll"][0] == 0.0
    for k in ("fms", "fPs", "sms", "sPs"):
        assert relerr(got[k], want[k]) < TOL64 or np.max(np.abs(got[k] - want[k])) < 1e-12


def test_rowcoop_c5_large_n_vs_c_oracle():
    """config c5's model (d = 11) at 2^16 steps, 20 % missing, against the sequential C oracle; reruns are
    bit-identical (fixed chain geometry, fixed combine order)."""
    from oracle import c_oracle as C
    sde = _kernels()["c5_qp_m52"]().get_sde()
    n = 1 << 16
    t = make_times(n, seed=9)
    ssm = O.get_ssm(sde, t, 0.1)
    y = sample_series(ssm, seed=9, nan_frac=0.2)
    cf, cP, cs, csP, cll = C.kfs(ssm, y)
    got = _gpu_all(ssm, y)
    _check(got, dict(fms=cf, fPs=cP, sms=cs, sPs=csP, ll=np.array([cll])), 1e-8)
    again = _gpu_all(ssm, y)
    for k in got:
        assert np.array_equal(got[k], again[k]), k


@pytest.mark.parametrize("name,n", [("rbf7", 1500), ("c5_qp_m52", 1201), ("periodic7", 700), ("rbf8", 1)])
def test_rowcoop_standalone_pks(name, n):
    """pks(lgssm, fms, fPs) (parallel.py:187-196) on its own for d > 6: smoothing elements from GIVEN filtered
    moments (here the oracle's), against the oracle's smoother."""
    from pssgp.kalman.parallel import pks
    sde = _kernels()[name]().get_sde()
    t = make_times(n, seed=13)
    ssm = O.get_ssm(sde, t, 0.1)
    y = sample_series(ssm, seed=13, nan_frac=0.2 if n > 4 else 0.0)
    fms, fPs, _ = O.kf(ssm, y, True)
    sms_o, sPs_o = O.kfs(ssm, y)
    sms, sPs = pks(ssm, fms, fPs)
    assert relerr(sms, sms_o) < 1e-7 and relerr(sPs, sPs_o) < 1e-7


def test_rowcoop_standalone_pks_forced_small_d(row_family):
    from pssgp.kalman.parallel import pks
    from pssgp.kernels import Matern52
    row_family.set_chunk(7)
    t = make_times(2200, seed=4)
    ssm = O.get_ssm(Matern52(1., 1.).get_sde(), t, 0.1)
    y = sample_series(ssm, seed=2, nan_frac=0.1)
    fms, fPs, _ = O.kf(ssm, y, True)
    sms_o, sPs_o = O.kfs(ssm, y)
    sms, sPs = pks(ssm, fms, fPs)
    assert relerr(sms, sms_o) < TOL64 and relerr(sPs, sPs_o) < TOL64


@pytest.mark.parametrize("name", ["rbf8", "c5_qp_m52", "rbf13"])
def test_fp32_series_at_large_d_run_through_fp64(name):
    """fp32 arrays at 7 <= d <= 16: widened, run on the fp64 row-cooperative kernels, narrowed (csrc/pgps_core.hip
    scan_f32_via_f64) -- pkf, pkfs, stand-alone pks and discretise, against the fp64 oracle at fp32 tolerance."""
    from pssgp import _backend as B
    from pssgp.kalman.parallel import pkf, pkfs, pks
    sde = _kernels()[name]().get_sde()
    t = make_times(1300, seed=6)
    ssm = O.get_ssm(sde, t, 0.1)
    y = sample_series(ssm, seed=6, nan_frac=0.2)
    want = _oracle_all(ssm, y)
    ssm32 = tuple(np.asarray(a, np.float32) for a in ssm)
    y32 = y.astype(np.float32)
    tol = 2e-3
    fms, fPs, ll = pkf(ssm32, y32[:, None], return_loglikelihood=True)
    assert fms.dtype == np.float32 and relerr(fms, want["fms"]) < tol and relerr(fPs, want["fPs"]) < tol
    assert abs(float(ll) - want["ll"][0]) < tol * abs(want["ll"][0])
    sms, sPs = pkfs(ssm32, y32[:, None])
    assert sms.dtype == np.float32 and relerr(sms, want["sms"]) < tol and relerr(sPs, want["sPs"]) < tol
    sms2, sPs2 = pks(ssm32, fms, fPs)
    assert relerr(sms2, want["sms"]) < tol and relerr(sPs2, want["sPs"]) < tol
    gFs, gQs = B.discretise(np.asarray(sde.F, np.float32), np.asarray(sde.P0, np.float32), t.astype(np.float32), 0.0)
    assert gFs.dtype == np.float32
    # time stamps rounded to fp32 (eps * t ~ 4e-6 on steps of 0.05) bound what any fp32 discretisation can reach
    assert np.max(np.abs(gFs - ssm[1])) < 1e-3 * max(1.0, float(np.max(np.abs(ssm[1]))))
    assert np.max(np.abs(gQs - ssm[2])) < 1e-3 * max(1.0, float(np.max(np.abs(ssm[0]))))


@pytest.mark.parametrize("name", ["c5", "rbf15", "co2_d18"])
def test_golden_large_d(name):
    """The committed sequential-oracle vectors for d = 11 (config c5's kernel), d = 15 and d = 18 (the reference's CO2
    kernel: wave-cooperative kernels) (tests/golden/large_d_n1024.npz, written by tests/golden/make_golden.py):
    through pkf / pkfs on the arrays and through the general-LTI device path."""
    import os
    from pssgp import _backend as B
    from pssgp.kalman.parallel import pkf, pkfs
    from pssgp.kernels import Matern32, Matern52, Periodic, RBF, SquaredExponential
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "large_d_n1024.npz"))
    k = {"c5": Periodic(SquaredExponential(1., 1.), period=1., order=1) * Matern32(1., 1.) + Matern52(1., 1.),
         "rbf15": RBF(1., 0.5, order=15, balancing_iter=10),
         "co2_d18": Periodic(SquaredExponential(1., 1.), period=1., order=3) * Matern32(0.5, 5.) + Matern32(1., 2.)}[name]
    ssm = k.get_ssm(g["t"][:, None], 0.1)
    fms, fPs, ll = pkf(ssm, g["y"][:, None], return_loglikelihood=True)
    sms, sPs = pkfs(ssm, g["y"][:, None])
    h = np.asarray(ssm.H).reshape(-1)
    steps = g["full_steps"]
    tol = 1e-7
    assert abs(float(ll) - float(g[name + "/ll"])) < tol * abs(float(g[name + "/ll"]))
    assert relerr(fms @ h, g[name + "/fmean"]) < tol and relerr(np.einsum("i,nij,j->n", h, fPs, h), g[name + "/fvar"]) < tol
    assert relerr(sms @ h, g[name + "/smean"]) < tol and relerr(np.einsum("i,nij,j->n", h, sPs, h), g[name + "/svar"]) < tol
    assert relerr(fPs[steps], g[name + "/fPs_full"]) < tol and relerr(sPs[steps], g[name + "/sPs_full"]) < tol
    assert relerr(sms[steps], g[name + "/sms_full"]) < tol
    sde = k.get_sde()
    assert abs(B.lti_ll(sde.F, sde.P0, sde.H, 0.1, g["t"], g["y"]) - float(g[name + "/ll"])) < tol * abs(float(g[name + "/ll"]))
    mean, var, _ = B.lti_predict(sde.F, sde.P0, sde.H, 0.1, g["t"], g["y"], g["t"][100:900:7])
    # a query AT a training time comes before it on ties, so it is smoothed with that observation still to come:
    # its posterior equals the smoothed moments of the training row only where that row is missing
    miss = np.isnan(g["y"][100:900:7])
    assert np.max(np.abs(mean[miss] - g[name + "/smean"][100:900:7][miss])) < 1e-6 * max(1.0, float(np.max(np.abs(g[name + "/smean"]))))
    assert np.max(np.abs(var[miss] - g[name + "/svar"][100:900:7][miss])) < 1e-6 * max(1.0, float(np.max(g[name + "/svar"])))


@pytest.mark.parametrize("dtype,kname,n", [(np.float64, "c5_qp_m52", 1 << 18), (np.float32, "rbf8", 1 << 15),
                                           (np.float32, "rbf6", 1 << 17), (np.float64, "m32", 1 << 18)])
def test_device_arrays_that_end_on_a_page_boundary(dtype, kname, n):
    """The device entry points on arrays of EXACTLY N records whose size is a multiple of 2 MiB -- what a framework's
    allocator hands out (bench.py's torch tensors) -- so that a single byte read or written beyond an array faults
    instead of landing in the slack of libpgps' own staging buffers (which is where every host-array test runs).  Round 3's
    first attempt to send the last workgroup of a series down the wide-load road read 8 bytes beyond the smoothed
    covariances at d = 11 -- the caller's array, where the smoothing elements wait -- and only bench.py noticed."""
    import ctypes
    from pssgp import _backend as B
    from pssgp.kernels import Matern32, RBF
    ks = dict(_kernels(), rbf6=lambda: RBF(1., 0.8, order=6, balancing_iter=10), m32=lambda: Matern32(1., 1.))
    sde = ks[kname]().get_sde()
    d = np.asarray(sde.F).shape[0]
    w = np.dtype(dtype).itemsize
    assert (n * d * d * w) % (1 << 21) == 0 and (n * w) % 4096 == 0
    suf = "f64" if dtype == np.float64 else "f32"
    t = make_times(n, seed=3)
    Fs, Qs = B.discretise(sde.F, sde.P0, t, 0.0)
    ssm = (np.asarray(sde.P0, np.float64), Fs, Qs, np.asarray(sde.H, np.float64).reshape(1, -1), np.array([[0.1]]))
    from tests.conftest import sample_series_fast
    y = sample_series_fast(ssm, seed=4, nan_frac=0.1)
    ctx = B.Context(0)
    host = dict(P0=ssm[0], Fs=Fs, Qs=Qs, H=ssm[3].reshape(-1), ys=y.reshape(-1))
    bufs, ptr = {}, {}
    try:
        for name, arr in host.items():
            a = np.ascontiguousarray(arr, dtype)
            ptr[name] = ctx.malloc(a.nbytes)            # one allocation per array, no slack
            ctx.h2d(ptr[name], a)
        for name, shape in (("fms", (n, d)), ("fPs", (n, d, d)), ("sms", (n, d)), ("sPs", (n, d, d))):
            bufs[name] = np.empty(shape, dtype)
            ptr[name] = ctx.malloc(bufs[name].nbytes)
        ptr["ll"] = ctx.malloc(16)
        P, L, I = ctypes.c_void_p, ctypes.c_long, ctypes.c_int
        R = (ctypes.c_double if dtype == np.float64 else ctypes.c_float)(0.1)
        ctx.call(f"pgps_pkfs_dev_{suf}", L(n), I(d), P(ptr["P0"]), P(ptr["Fs"]), P(ptr["Qs"]), P(ptr["H"]), R, P(ptr["ys"]),
                 P(ptr["fms"]), P(ptr["fPs"]), P(ptr["sms"]), P(ptr["sPs"]), P(ptr["ll"]))
        ctx.synchronize()
        for name in bufs:
            ctx.d2h(bufs[name], ptr[name])
        ll = np.empty(2, np.float64)
        ctx.d2h(ll, ptr["ll"])
    finally:
        for p in ptr.values():
            ctx.free(p)
        ctx.close()
    ssm_t = tuple(np.asarray(a, dtype) for a in ssm)
    sms, sPs, fms, fPs, ll_h = B.pkfs(ssm_t, np.asarray(y, dtype), return_filtered=True, return_loglikelihood=True)
    tol = 1e-12 if dtype == np.float64 else 1e-5
    for name, ref in (("fms", fms), ("fPs", fPs), ("sms", sms), ("sPs", sPs)):
        assert relerr(bufs[name], ref) < tol, name
    assert abs(ll[0] - float(ll_h)) <= tol * abs(float(ll_h))
