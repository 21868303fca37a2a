"""ctypes loader of oracle/_build/liboracle_seq.so (CPU ORACLE -- TEST INFRASTRUCTURE ONLY).

`kfs(lgssm, ys)` runs the plain-C restatement of pssgp/kalman/sequential.py (kalman_seq.c) and
returns (fms, fPs, sms, sPs, ll).  Used by tests/ at sizes numpy loops cannot reach and by
bench.py's cpu_baseline leg.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_seq.so")
_lib = None


def build():
    subprocess.run(["make", "-s", "-C", _HERE], check=True)
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
