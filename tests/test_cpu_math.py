"""CPU check of the device algebra (parallel-gps_amd/csrc/pgps_math.h): the header is compiled
with g++ into a small harness (tests/cpu_math/emul.cpp) that runs the same chunked three-phase
scan the HIP kernels run, as plain host loops, and is compared with the numpy oracle.  This
validates the math the kernels are built from without a GPU; the kernels themselves are checked
on the GPU in test_gpu_parity.py."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from oracle import np_oracle as O
from tests.conftest import make_times, relerr, sample_series

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = os.environ.get("PGPS_SAN") == "1"             # ASan + UBSan build of the harness (tests/test_sanitizers.py)
BUILD = os.path.join(ROOT, "tests", "cpu_math", "_build_san" if SAN else "_build")
SAN_FLAGS = ["-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]


@pytest.fixture(scope="module")
def emul():
    os.makedirs(BUILD, exist_ok=True)
    so = os.path.join(BUILD, "libemul.so")
    src = os.path.join(ROOT, "tests", "cpu_math", "emul.cpp")
    hdr = os.path.join(ROOT, "parallel-gps_amd", "csrc", "pgps_math.h")
    hdr2 = os.path.join(ROOT, "parallel-gps_amd", "csrc", "pgps_dual.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in (src, hdr, hdr2)):
        subprocess.run(["g++"] + (SAN_FLAGS if SAN else ["-O2"]) + ["-std=c++17", "-shared", "-fPIC", "-I",
                                                                     os.path.dirname(hdr), src, "-o", so], check=True)
    return ctypes.CDLL(so)


def run_emul(lib, ssm, ys, Lc, W, dtype, dform=False):
    P0, Fs, Qs, H, R = ssm
    N, d = Fs.shape[0], Fs.shape[1]
    c = lambda a: np.ascontiguousarray(a, dtype=dtype)
    P0, Fs, Qs, H, ys = c(P0), c(Fs), c(Qs), c(H).reshape(-1), c(ys)
    fms, sms = np.empty((N, d), dtype), np.empty((N, d), dtype)
    fPs, sPs = np.empty((N, d, d), dtype), np.empty((N, d, d), dtype)
    ll = ctypes.c_double()
    if dform:
        fn, real = (lib.emul_pkfs_dform_f64, ctypes.c_double) if dtype == np.float64 else (lib.emul_pkfs_dform_f32, ctypes.c_float)
    else:
        fn, real = (lib.emul_pkfs_f64, ctypes.c_double) if dtype == np.float64 else (lib.emul_pkfs_f32, ctypes.c_float)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rc = fn(ctypes.c_int(d), ctypes.c_long(N), ctypes.c_int(Lc), ctypes.c_int(W), p(P0), p(Fs), p(Qs), p(H),
            real(float(np.asarray(R).reshape(()))), p(ys), p(fms), p(fPs), p(sms), p(sPs), ctypes.byref(ll))
    assert rc == 0
    return fms, fPs, sms, sPs, ll.value


@pytest.mark.parametrize("idx", range(7))
def test_chunked_scan_math_fp64(emul, kernel_zoo, idx):
    name, make, _, _ = kernel_zoo[idx]
    t = make_times(333, seed=idx)
    ssm = O.get_ssm(make().get_sde(), t, 0.1)
    y = sample_series(ssm, seed=idx, nan_frac=0.2)
    fms, fPs, ll = O.pkf(ssm, y, True)
    sms, sPs = O.pks(ssm, fms, fPs)
    for Lc, W in ((5, 8), (1, 64), (16, 4)):
        e = run_emul(emul, ssm, y, Lc, W, np.float64)
        assert relerr(e[0], fms) < 1e-11 and relerr(e[1], fPs) < 1e-11
        assert relerr(e[2], sms) < 1e-11 and relerr(e[3], sPs) < 1e-11
        assert abs(e[4] - ll) < 1e-11 * abs(ll)


@pytest.mark.parametrize("idx", range(7))
def test_innovation_form_totals_reproduce_the_rts_smoother(emul, kernel_zoo, idx):
    """The smoothing totals kept relative to the filtered moments (pgps_math.h kf_step_u / smth_extend_u: the element's L
    is the rank-one -v v^T / S, the fold of a step one matrix product and a rank-one update) give the reference's smoother
    (parallel.py:159-184) to round-off -- every kernel of the zoo, three chunkings, 20 % missing observations, fp64 and fp32,
    and series that end inside a chunk."""
    name, make, _, _ = kernel_zoo[idx]
    for n in (333, 64, 17):
        t = make_times(n, seed=idx + n)
        ssm = O.get_ssm(make().get_sde(), t, 0.1)
        y = sample_series(ssm, seed=idx, nan_frac=0.2)
        fms, fPs, ll = O.pkf(ssm, y, True)
        sms, sPs = O.pks(ssm, fms, fPs)
        for Lc, W in ((5, 8), (1, 64), (16, 4)):
            e = run_emul(emul, ssm, y, Lc, W, np.float64, dform=True)
            assert relerr(e[0], fms) < 1e-11 and relerr(e[1], fPs) < 1e-11
            assert relerr(e[2], sms) < 1e-10 and relerr(e[3], sPs) < 1e-10, (name, n, Lc, W)
            assert abs(e[4] - ll) < 1e-11 * abs(ll)
    e32 = run_emul(emul, ssm, y, 7, 16, np.float32, dform=True)
    assert relerr(e32[2], sms) < 1e-3 and relerr(e32[3], sPs) < 1e-3


@pytest.mark.parametrize("idx", [1, 3, 6])
def test_chunked_scan_math_fp32(emul, kernel_zoo, idx):
    """fp32 operands (fp64 log-likelihood accumulation): within the north-star's 1e-3."""
    name, make, _, _ = kernel_zoo[idx]
    t = make_times(400, seed=idx)
    ssm = O.get_ssm(make().get_sde(), t, 0.1)
    y = sample_series(ssm, seed=idx, nan_frac=0.1)
    fms, fPs, ll = O.kf(ssm, y, True)
    sms, sPs = O.kfs(ssm, y)
    e = run_emul(emul, ssm, y, 7, 16, np.float32)
    assert relerr(e[0], fms) < 1e-3 and relerr(e[1], fPs) < 1e-3
    assert relerr(e[2], sms) < 1e-3 and relerr(e[3], sPs) < 1e-3
    assert abs(e[4] - ll) < 1e-3 * abs(ll)


@pytest.mark.parametrize("n", [1, 2, 3, 9])
def test_tiny_series(emul, n):
    from pssgp.kernels import Matern32
    t = make_times(n, seed=n)
    ssm = O.get_ssm(Matern32(1., 1.).get_sde(), t, 0.1)
    y = sample_series(ssm, seed=n)
    fms, fPs, ll = O.kf(ssm, y, True)
    sms, sPs = O.kfs(ssm, y)
    e = run_emul(emul, ssm, y, 2, 4, np.float64)
    assert relerr(e[0], fms) < 1e-12 and relerr(e[2], sms) < 1e-12 and abs(e[4] - ll) < 1e-12 * abs(ll)


# ----------------------------------------------------------------------------------------------
# log-likelihood gradient on dual numbers (pgps_dual.h; the algebra of k_grad_reduce / k_grad_apply)
# ----------------------------------------------------------------------------------------------
def fd_grad(fun, x, rel=1e-5):
    """4th-order central differences of a scalar function of a parameter vector."""
    g = np.zeros(len(x))
    for i in range(len(x)):
        def at(h):
            z = np.array(x, float)
            z[i] += h
            return fun(z)
        h = rel * max(abs(x[i]), 1e-2)
        g[i] = (8.0 * (at(h) - at(-h)) - (at(2 * h) - at(-2 * h))) / (12.0 * h)
    return g


def grad_case(kname, n, seed, nan_frac=0.0):
    from pssgp.kernels import Matern12, Matern32, Matern52
    cls, spec_name = {"m12": (Matern12, "matern12"), "m32": (Matern32, "matern32"), "m52": (Matern52, "matern52")}[kname]
    rng = np.random.RandomState(seed)
    t = np.sort(rng.rand(n)) * (n / 100.0)
    y = np.sin(2.0 * t) + 0.4 * rng.randn(n)
    if nan_frac:
        y[rng.rand(n) < nan_frac] = np.nan
    return cls, spec_name, t, y


@pytest.mark.parametrize("kname", ["m12", "m32", "m52"])
@pytest.mark.parametrize("Lc", [1, 7, 16])
def test_dual_loglik_gradient_vs_dense_gp(emul, kname, Lc):
    """d ll / d (variance, lengthscale, noise) from ONE dual-number pass of the chunked filter ==
    finite differences of the dense GP marginal likelihood (the reference checks its autodiff
    gradient the same way, tests/test_gp_vs_kfs.py:53-78, at 1e-2; here 1e-6)."""
    from pssgp import _backend
    from pssgp.model import StateSpaceGP
    cls, spec_name, t, y = grad_case(kname, 150, 3)
    theta = np.array([1.3, 0.7, 0.2])
    m = StateSpaceGP((t[:, None], y[:, None]), cls(theta[0], theta[1]), noise_variance=theta[2], parallel=True)
    model, d, npar = _backend.pack_grad_model(m._grad_blocks())
    out = np.zeros(1 + npar)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rc = emul.emul_ll_grad(ctypes.c_int(d), ctypes.c_long(t.size), ctypes.c_int(Lc), ctypes.c_int(npar), p(model),
                           p(np.ascontiguousarray(t)), ctypes.c_double(0.0), p(np.ascontiguousarray(y)), p(out))
    assert rc == 0
    dense = lambda th: O.dense_gp((spec_name, th[0], th[1]), t, y, th[2])
    assert abs(out[0] - dense(theta)) < 1e-8 * abs(dense(theta))
    g = fd_grad(dense, theta)
    assert relerr(out[1:], g) < 1e-6, (out[1:], g)


def test_dual_loglik_gradient_with_missing(emul):
    """Missing observations (NaN) contribute nothing to ll or to its gradient: against finite
    differences of the oracle's sequential filter on a longer series."""
    from pssgp import _backend
    from pssgp.model import StateSpaceGP
    cls, _, t, y = grad_case("m32", 3000, 5, nan_frac=0.2)
    theta = np.array([0.8, 1.1, 0.3])
    m = StateSpaceGP((t[:, None], y[:, None]), cls(theta[0], theta[1]), noise_variance=theta[2], parallel=True)
    model, d, npar = _backend.pack_grad_model(m._grad_blocks())
    out = np.zeros(1 + npar)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    assert emul.emul_ll_grad(ctypes.c_int(d), ctypes.c_long(t.size), ctypes.c_int(16), ctypes.c_int(npar), p(model),
                             p(np.ascontiguousarray(t)), ctypes.c_double(0.0), p(np.ascontiguousarray(y)), p(out)) == 0
    seq = lambda th: float(O.ssgp_log_likelihood(cls(th[0], th[1]).get_sde(), t, y, th[2], parallel=False))
    assert abs(out[0] - seq(theta)) < 1e-9 * abs(seq(theta))
    assert relerr(out[1:], fd_grad(seq, theta)) < 1e-6


def test_grad_model_blocks_are_exact():
    """The host-side model derivatives (Richardson central differences of get_sde()) against the
    closed forms for Matern-3/2: lam = sqrt(3)/l, N = [[lam, 1], [-lam^2, -lam]], Pinf = s2 diag(1, lam^2)."""
    from pssgp.kernels import Matern32
    from pssgp.model import StateSpaceGP
    s2, l, r = 1.7, 0.6, 0.25
    m = StateSpaceGP((np.zeros((1, 1)), np.zeros((1, 1))), Matern32(s2, l), noise_variance=r, parallel=True)
    base, d_s2, d_l, d_r = m._grad_blocks()
    lam = np.sqrt(3.0) / l
    dlam = -lam / l
    assert abs(base[0] - lam) < 1e-14
    assert np.allclose(d_s2[2], np.diag([1.0, lam ** 2]), rtol=1e-9, atol=1e-9) and abs(d_s2[0]) < 1e-9
    assert abs(d_l[0] - dlam) < 1e-8 * abs(dlam)
    assert np.allclose(d_l[1], np.array([[dlam, 0.0], [-2 * lam * dlam, -dlam]]), rtol=1e-8, atol=1e-8)
    assert np.allclose(d_l[2], np.diag([0.0, 2 * lam * dlam * s2]), rtol=1e-8, atol=1e-8)
    assert abs(d_r[4] - 1.0) < 1e-9 and np.allclose(d_r[2], 0.0, atol=1e-9)
    assert m.kernel.lengthscales == l and m.kernel.variance == s2 and m.noise_variance == r


@pytest.mark.parametrize("kname", ["m12", "m32", "m52"])
def test_closed_form_grad_blocks_match_the_differences(kname):
    """A single Matern kernel's model derivatives come in closed form (no get_sde() beyond the evaluation's own); they
    are the Richardson differences of get_sde() -- for Matern-5/2 that is the statement that the balancing iteration
    commutes with the lengthscale's time scaling (model.py _grad_blocks_matern)."""
    from pssgp.kernels import Matern12, Matern32, Matern52
    from pssgp.model import StateSpaceGP
    cls = {"m12": Matern12, "m32": Matern32, "m52": Matern52}[kname]
    rng = np.random.default_rng(17)
    for _ in range(4):
        s2, l, r = float(rng.uniform(0.2, 3.0)), float(rng.uniform(0.05, 2.0)), float(rng.uniform(0.01, 1.0))
        m = StateSpaceGP((np.zeros((1, 1)), np.zeros((1, 1))), cls(s2, l), noise_variance=r, parallel=True)
        closed, diffs = m._grad_blocks(), m._grad_blocks(closed_form=False)
        assert m._grad_blocks_matern() is not None and len(closed) == len(diffs) == 4
        for rc, rd in zip(closed, diffs):
            for a, b in zip(rc, rd):
                a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
                assert a.shape == b.shape and np.max(np.abs(a - b)) <= 1e-9 * (1.0 + np.max(np.abs(b)))
    # anything else keeps the differences
    m = StateSpaceGP((np.zeros((1, 1)), np.zeros((1, 1))), Matern32(1.0, 1.0) + Matern52(1.0, 1.0), parallel=True)
    assert m._grad_blocks_matern() is None


def test_composite_gradient_rows_give_way_when_the_block_structure_moves(monkeypatch):
    """A coupling entry near nilpotent_blocks' threshold can put x0 + h and x0 - h into different block partitions; the
    dual-number rows are then not defined and the model falls back to differencing the likelihood (no exception)."""
    from pssgp import _backend
    from pssgp.kernels import Matern32, Matern52
    from pssgp.model import StateSpaceGP
    m = StateSpaceGP((np.zeros((1, 1)), np.zeros((1, 1))), Matern32(1.0, 1.0) + Matern52(0.5, 2.0), parallel=True)
    rows, sizes = m._grad_rows_composite()
    assert rows is not None and sizes == [2, 3] and len(rows) == 1 + len(m.trainable_parameters())
    real = _backend.nilpotent_blocks
    calls = []

    def moving(F, *a, **k):
        calls.append(1)
        out = real(F, *a, **k)
        return out if len(calls) < 3 else out[:1] + [(2, 1, out[1][2], out[1][3][:1, :1]), (3, 2, out[1][2], out[1][3][1:, 1:])]
    monkeypatch.setattr(_backend, "nilpotent_blocks", moving)
    assert m._grad_rows_composite() == (None, None)
    assert m.kernel.kernels[0].variance == 1.0 and m.kernel.kernels[1].lengthscales == 2.0     # parameters restored


@pytest.mark.parametrize("kname", ["m12", "m32", "m52"])
def test_matern_forms_without_building_the_sde(kname):
    """The model's fast path for a single Matern kernel (a new hyper-parameter setting in an optimiser loop: scalar
    formulas instead of get_sde + stationarity check + nilpotent form) gives what the general path gives."""
    from pssgp import _backend
    from pssgp.kernels import Matern12, Matern32, Matern52
    from pssgp.model import StateSpaceGP
    cls = {"m12": Matern12, "m32": Matern32, "m52": Matern52}[kname]
    rng = np.random.default_rng(5)
    m = StateSpaceGP((np.zeros((1, 1)), np.zeros((1, 1))), cls(1.0, 1.0), noise_variance=0.1, parallel=True)
    for _ in range(5):
        m.kernel.variance, m.kernel.lengthscales = float(rng.uniform(0.1, 5.0)), float(rng.uniform(0.02, 3.0))
        fast = m._matern_forms()
        assert fast is not None
        sde = m.kernel.get_sde()
        form = _backend.nilpotent_form(sde.F)
        for a, b in ((fast[1][0], form[0]), (fast[1][1], form[1]), (fast[1][2], form[2]), (fast[0].P0, sde.P0),
                     (np.asarray(fast[0].H).reshape(-1), np.asarray(sde.H).reshape(-1))):
            a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
            assert a.shape == b.shape and np.max(np.abs(a - b)) <= 1e-11 * (1.0 + np.max(np.abs(b)))
        assert m._device_forms()[0][1][0] == fast[1][0]             # and it is what the model uses
    both = StateSpaceGP((np.zeros((1, 1)), np.zeros((1, 1))), Matern32(1.0, 1.0) + Matern52(1.0, 1.0), parallel=True)
    assert both._matern_forms() is None


def test_form_memos_follow_the_kernel_object():
    """The memoised forms are keyed on the kernel OBJECT and its numbers: a model whose kernel is replaced by another
    class with the same hyper-parameters (or whose RBF changes order) must not find the old forms."""
    from pssgp.kernels import Matern32, Matern52, RBF
    from pssgp.model import StateSpaceGP
    m = StateSpaceGP((np.zeros((1, 1)), np.zeros((1, 1))), Matern32(1.3, 0.7), noise_variance=0.1, parallel=True)
    f32 = m._device_forms()[0][1]
    assert m._device_forms()[0][1] is f32                        # memo hit
    m.kernel = Matern52(1.3, 0.7)
    f52 = m._device_forms()[0][1]
    assert f52[1].shape == (3, 3) and f32[1].shape == (2, 2)
    m.kernel = RBF(1.0, 0.5, order=4, balancing_iter=5)
    a = m._device_forms()[1]
    m.kernel = RBF(1.0, 0.5, order=6, balancing_iter=5)
    b = m._device_forms()[1]
    assert a.F.shape == (4, 4) and b.F.shape == (6, 6)
    m.kernel.lengthscales = 0.5 * 1.1                            # nearby: scaled from the reference setting, same object
    c = m._device_forms()[1]
    ref = m.kernel.get_sde()
    assert c.F.shape == (6, 6) and np.allclose(np.linalg.eigvals(c.F), np.linalg.eigvals(np.asarray(ref.F)), rtol=1e-6)
