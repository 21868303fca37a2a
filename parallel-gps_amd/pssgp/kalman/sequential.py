"""Sequential Kalman filter / RTS smoother: the reference's `kf / ks / kfs`
(pssgp/kalman/sequential.py:11-73), the `parallel=False` mode of StateSpaceGP.

Runs on the host like the reference's (`tf.scan` on `/cpu:0`), through the C++ twins in
libpgps.so (`pgps_seq_kf_* / pgps_seq_ks_*`, csrc/pgps_seq_host.cpp).
"""
import ctypes

import numpy as np

from .. import _backend

__all__ = ["kf", "ks", "kfs"]


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _declare(lib):
    if getattr(lib, "_pgps_seq_declared", False):
        return
    P, L, I = ctypes.c_void_p, ctypes.c_long, ctypes.c_int
    for suf, real in (("f64", ctypes.c_double), ("f32", ctypes.c_float)):
        getattr(lib, f"pgps_seq_kf_{suf}").argtypes = [L, I, P, P, P, P, real, P, P, P, P, P, P]
        getattr(lib, f"pgps_seq_ks_{suf}").argtypes = [L, I, P, P, P, P, P, P, P]
    lib._pgps_seq_declared = True


def kf(lgssm, observations, return_loglikelihood=False, return_predicted=False):
    lib = _backend.load_library()
    _declare(lib)
    dtype = _backend._dtype_of(lgssm)
    suf, real = _backend._suffix(dtype)
    P0, Fs, Qs, H, R, N, d = _backend._unpack_lgssm(lgssm, dtype)
    ys = _backend._prep(observations, dtype, (-1,))
    fms, fPs = np.empty((N, d), dtype), np.empty((N, d, d), dtype)
    mps = np.empty((N, d), dtype) if return_predicted else None
    Pps = np.empty((N, d, d), dtype) if return_predicted else None
    ll = ctypes.c_double(0.0)
    code = getattr(lib, f"pgps_seq_kf_{suf}")(N, d, _ptr(P0), _ptr(Fs), _ptr(Qs), _ptr(H), real(R), _ptr(ys),
                                              _ptr(fms), _ptr(fPs), ctypes.cast(ctypes.byref(ll), ctypes.c_void_p),
                                              _ptr(mps), _ptr(Pps))
    _backend.check(None, code, "pgps_seq_kf")
    out = (fms, fPs)
    if return_loglikelihood:
        out += (np.asarray(ll.value, dtype=dtype),)
    if return_predicted:
        out += (mps, Pps)
    return out


def ks(lgssm, ms, Ps, mps, Pps):
    lib = _backend.load_library()
    _declare(lib)
    dtype = _backend._dtype_of(lgssm)
    suf, _ = _backend._suffix(dtype)
    Fs = _backend._prep(lgssm[1], dtype)
    N, d = Fs.shape[0], Fs.shape[1]
    ms, mps = _backend._prep(ms, dtype, (N, d)), _backend._prep(mps, dtype, (N, d))
    Ps, Pps = _backend._prep(Ps, dtype, (N, d, d)), _backend._prep(Pps, dtype, (N, d, d))
    sms, sPs = np.empty((N, d), dtype), np.empty((N, d, d), dtype)
    code = getattr(lib, f"pgps_seq_ks_{suf}")(N, d, _ptr(Fs), _ptr(ms), _ptr(Ps), _ptr(mps), _ptr(Pps),
                                              _ptr(sms), _ptr(sPs))
    _backend.check(None, code, "pgps_seq_ks")
    return sms, sPs


def kfs(model, observations):
    fms, fPs, mps, Pps = kf(model, observations, return_predicted=True)
    return ks(model, fms, fPs, mps, Pps)
