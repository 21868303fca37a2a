"""ctypes loader of oracle/_build/liboracle_seq.so (CPU ORACLE -- TEST INFRASTRUCTURE ONLY).

`kfs(lgssm, ys)` runs the plain-C restatement of pssgp/kalman/sequential.py (kalman_seq.c) and
returns (fms, fPs, sms, sPs, ll).  Used by tests/ at sizes numpy loops cannot reach and by
bench.py's cpu_baseline leg.  `par_kfs(lgssm, ys, nthreads)` is the same pass as a chunked scan on all host cores
(kalman_par.c, OpenMP, fp64): bench.py's cpu_baseline_all_cores.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SAN = os.environ.get("PGPS_SAN") == "1"            # ASan + UBSan build (tests/test_sanitizers.py)
_BUILD = "_build_san" if _SAN else "_build"
_SO = os.path.join(_HERE, _BUILD, "liboracle_seq.so")
_lib = None


def build():
    subprocess.run(["make", "-s", "-C", _HERE] + (["SAN=1"] if _SAN else []), check=True)
    return _SO


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "kalman_seq.c")):
            build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def kfs(lgssm, ys, dtype=np.float64):
    lib = load()
    P0, Fs, Qs, H, R = lgssm
    dtype = np.dtype(dtype)
    suf, real = ("f64", ctypes.c_double) if dtype == np.float64 else ("f32", ctypes.c_float)
    c = lambda a: np.ascontiguousarray(a, dtype=dtype)
    P0, Fs, Qs, H, ys = c(P0), c(Fs), c(Qs), c(H).reshape(-1), c(ys).reshape(-1)
    N, d = Fs.shape[0], Fs.shape[1]
    fms, sms = np.empty((N, d), dtype), np.empty((N, d), dtype)
    fPs, sPs = np.empty((N, d, d), dtype), np.empty((N, d, d), dtype)
    ll = ctypes.c_double(0.0)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    fn = getattr(lib, f"oracle_kfs_{suf}")
    fn.argtypes = [ctypes.c_long, ctypes.c_int] + [ctypes.c_void_p] * 4 + [real] + [ctypes.c_void_p] * 6
    rc = fn(N, d, p(P0), p(Fs), p(Qs), p(H), real(float(np.asarray(R).reshape(()))), p(ys), p(fms), p(fPs),
            p(sms), p(sPs), ctypes.cast(ctypes.byref(ll), ctypes.c_void_p))
    if rc != 0:
        raise RuntimeError(f"oracle_kfs failed with code {rc}")
    return fms, fPs, sms, sPs, ll.value


_SO_PAR = os.path.join(_HERE, _BUILD, "liboracle_par.so")
_lib_par = None


def load_par():
    global _lib_par
    if _lib_par is None:
        if not os.path.exists(_SO_PAR) or os.path.getmtime(_SO_PAR) < os.path.getmtime(os.path.join(_HERE, "kalman_par.c")):
            build()
        _lib_par = ctypes.CDLL(_SO_PAR)
    return _lib_par


def par_max_threads():
    return int(load_par().oracle_par_max_threads())


def par_kfs(lgssm, ys, nthreads=0):
    """(fms, fPs, sms, sPs, ll) by the chunked scan of kalman_par.c on `nthreads` cores (0 = all)."""
    lib = load_par()
    P0, Fs, Qs, H, R = lgssm
    c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    P0, Fs, Qs, H, ys = c(P0), c(Fs), c(Qs), c(H).reshape(-1), c(ys).reshape(-1)
    N, d = Fs.shape[0], Fs.shape[1]
    fms, sms = np.empty((N, d)), np.empty((N, d))
    fPs, sPs = np.empty((N, d, d)), np.empty((N, d, d))
    ll = ctypes.c_double(0.0)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    fn = lib.oracle_par_kfs_f64
    fn.argtypes = [ctypes.c_long, ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_double] + [ctypes.c_void_p] * 6 + \
        [ctypes.c_int]
    rc = fn(N, d, p(P0), p(Fs), p(Qs), p(H), float(np.asarray(R).reshape(())), p(ys), p(fms), p(fPs), p(sms), p(sPs),
            ctypes.cast(ctypes.byref(ll), ctypes.c_void_p), int(nthreads))
    if rc != 0:
        raise RuntimeError(f"oracle_par_kfs failed with code {rc}")
    return fms, fPs, sms, sPs, ll.value
