"""Exercises the CPU builds of the native code under AddressSanitizer + UndefinedBehaviorSanitizer.

Run by tests/test_sanitizers.py as a child process with LD_PRELOAD=libasan and PGPS_SAN=1 (which makes oracle/c_oracle.py
and tests/test_cpu_math.py build and load their `_build_san` variants).  Covers oracle/kalman_seq.c, oracle/kalman_par.c,
tests/cpu_math/emul.cpp (= csrc/pgps_math.h, csrc/pgps_dual.h on the host) and csrc/pgps_seq_host.cpp; every result is
also compared with the numpy oracle, so the sanitizers watch real work.  CPU only: the GPU build is never sanitized."""
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "parallel-gps_amd")):
    sys.path.insert(0, p)
assert os.environ.get("PGPS_SAN") == "1"

from oracle import c_oracle, np_oracle as O                      # noqa: E402
from tests.conftest import make_times, relerr, sample_series     # noqa: E402
from tests import test_cpu_math as tcm                           # noqa: E402


def kernels():
    from pssgp.kernels import Matern32, Matern52, RBF, Periodic, SquaredExponential
    return [Matern32(1., 0.7), Matern32(1., 1.) + Matern52(1., 0.7), RBF(1., 0.8, order=6, balancing_iter=10),
            Periodic(SquaredExponential(1., 1.), period=1., order=1) * Matern32(1., 1.) + Matern52(1., 1.)]


def main():
    # 1. the C oracles (sequential; chunked scan over threads)
    for i, k in enumerate(kernels()):
        for n in (1, 2, 131, 300):
            ssm = O.get_ssm(k.get_sde(), make_times(n, seed=i), 0.1)
            y = sample_series(ssm, seed=i, nan_frac=0.2 if n > 2 else 0.0)
            of, oP, oll = O.kf(ssm, y, True)
            os_, osP = O.kfs(ssm, y)
            for dtype, tol in ((np.float64, 1e-9), (np.float32, 5e-3)):
                f, P, s, sP, ll = c_oracle.kfs(ssm, y, dtype)
                assert relerr(f, of) < tol and relerr(sP, osP) < tol and abs(ll - oll) < tol * max(1.0, abs(oll)), (i, n, dtype)
            if ssm[1].shape[1] <= 16:
                for threads in (1, 3):
                    f, P, s, sP, ll = c_oracle.par_kfs(ssm, y, threads)
                    assert relerr(s, os_) < 1e-8 and abs(ll - oll) < 1e-8 * max(1.0, abs(oll)), (i, n, threads)
    print("oracle C files: ok")

    # 2. the device algebra on the host (emul.cpp over pgps_math.h / pgps_dual.h)
    os.makedirs(tcm.BUILD, exist_ok=True)
    so = os.path.join(tcm.BUILD, "libemul.so")
    src = os.path.join(ROOT, "tests", "cpu_math", "emul.cpp")
    inc = os.path.join(ROOT, "parallel-gps_amd", "csrc")
    subprocess.run(["g++"] + tcm.SAN_FLAGS + ["-std=c++17", "-shared", "-fPIC", "-I", inc, src, "-o", so], check=True)
    emul = ctypes.CDLL(so)
    for i, k in enumerate(kernels()[:3]):
        ssm = O.get_ssm(k.get_sde(), make_times(301, seed=10 + i), 0.1)
        y = sample_series(ssm, seed=10 + i, nan_frac=0.2)
        fms, fPs, ll = O.pkf(ssm, y, True)
        sms, sPs = O.pks(ssm, fms, fPs)
        for Lc, W in ((5, 8), (1, 64), (16, 4)):
            e = tcm.run_emul(emul, ssm, y, Lc, W, np.float64)
            assert relerr(e[0], fms) < 1e-10 and relerr(e[3], sPs) < 1e-10 and abs(e[4] - ll) < 1e-10 * abs(ll)
        e = tcm.run_emul(emul, ssm, y, 8, 16, np.float32)
        assert relerr(e[2], sms) < 5e-3
    print("device algebra on the host: ok")

    # 3. the product's sequential mode (csrc/pgps_seq_host.cpp), as its own sanitized library
    bdir = os.path.join(ROOT, "parallel-gps_amd", "csrc", "build_san")
    os.makedirs(bdir, exist_ok=True)
    so = os.path.join(bdir, "libpgps_seq_san.so")
    subprocess.run(["g++"] + tcm.SAN_FLAGS + ["-std=c++17", "-shared", "-fPIC", os.path.join(inc, "pgps_seq_host.cpp"), "-o", so],
                   check=True)
    lib = ctypes.CDLL(so)
    P, L, I = ctypes.c_void_p, ctypes.c_long, ctypes.c_int
    for suf, real in (("f64", ctypes.c_double), ("f32", ctypes.c_float)):
        getattr(lib, f"pgps_seq_kf_{suf}").argtypes = [L, I, P, P, P, P, real, P, P, P, P, P, P]
        getattr(lib, f"pgps_seq_ks_{suf}").argtypes = [L, I, P, P, P, P, P, P, P]
    ptr = lambda a: None if a is None else a.ctypes.data_as(P)
    for i, k in enumerate(kernels()):
        ssm = O.get_ssm(k.get_sde(), make_times(200, seed=20 + i), 0.1)
        y = sample_series(ssm, seed=20 + i, nan_frac=0.2)
        of, oP, oll = O.kf(ssm, y, True)
        os_, osP = O.kfs(ssm, y)
        for dtype, suf, real, tol in ((np.float64, "f64", ctypes.c_double, 1e-10), (np.float32, "f32", ctypes.c_float, 5e-3)):
            c = lambda a: np.ascontiguousarray(a, dtype=dtype)
            P0, Fs, Qs, H, ys = c(ssm[0]), c(ssm[1]), c(ssm[2]), c(ssm[3]).reshape(-1), c(y)
            n, d = Fs.shape[0], Fs.shape[1]
            fms, fPs, mps, Pps = np.empty((n, d), dtype), np.empty((n, d, d), dtype), np.empty((n, d), dtype), np.empty((n, d, d), dtype)
            sms, sPs = np.empty((n, d), dtype), np.empty((n, d, d), dtype)
            ll = ctypes.c_double(0.0)
            rc = getattr(lib, f"pgps_seq_kf_{suf}")(n, d, ptr(P0), ptr(Fs), ptr(Qs), ptr(H), real(0.1), ptr(ys), ptr(fms), ptr(fPs),
                                                    ctypes.cast(ctypes.byref(ll), P), ptr(mps), ptr(Pps))
            assert rc == 0
            rc = getattr(lib, f"pgps_seq_ks_{suf}")(n, d, ptr(Fs), ptr(fms), ptr(fPs), ptr(mps), ptr(Pps), ptr(sms), ptr(sPs))
            assert rc == 0
            assert relerr(fms, of) < tol and relerr(sms, os_) < tol and relerr(sPs, osP) < tol
            assert abs(ll.value - oll) < tol * abs(oll)
            # the error paths too: null pointers and bad sizes come back as codes
            assert getattr(lib, f"pgps_seq_kf_{suf}")(0, d, ptr(P0), ptr(Fs), ptr(Qs), ptr(H), real(0.1), ptr(ys), ptr(fms), ptr(fPs),
                                                      None, None, None) != 0
    print("sequential host mode: ok")
    print("SAN OK")


if __name__ == "__main__":
    main()
