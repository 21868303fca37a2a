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
BUILD = os.path.join(ROOT, "tests", "cpu_math", "_build")


@pytest.fixture(scope="module")
def emul():
    os.makedirs(BUILD, exist_ok=True)
    so = os.path.join(BUILD, "libemul.so")
    src = os.path.join(ROOT, "tests", "cpu_math", "emul.cpp")
    hdr = os.path.join(ROOT, "parallel-gps_amd", "csrc", "pgps_math.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I", os.path.dirname(hdr), src, "-o", so],
                       check=True)
    return ctypes.CDLL(so)


def run_emul(lib, ssm, ys, Lc, W, dtype):
    P0, Fs, Qs, H, R = ssm
    N, d = Fs.shape[0], Fs.shape[1]
    c = lambda a: np.ascontiguousarray(a, dtype=dtype)
    P0, Fs, Qs, H, ys = c(P0), c(Fs), c(Qs), c(H).reshape(-1), c(ys)
    fms, sms = np.empty((N, d), dtype), np.empty((N, d), dtype)
    fPs, sPs = np.empty((N, d, d), dtype), np.empty((N, d, d), dtype)
    ll = ctypes.c_double()
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
