"""Parallel-in-time Kalman filter and RTS smoother: the reference's `pkf / pks / pkfs`
(pssgp/kalman/parallel.py:121-201) as calls into the HIP library.

Signatures are the reference's; arrays are numpy (host) in and out.  `max_parallel` is
accepted and ignored: the reference needs it to bound the depth of tfp's recursion
(parallel.py:127,188), the HIP scan handles any length.
"""
from .. import _backend

__all__ = ["pkf", "pks", "pkfs"]


def pkf(lgssm, observations, return_loglikelihood=False, max_parallel=10000):
    """Filtered means (N, d) and covariances (N, d, d) [+ log-likelihood] (parallel.py:121-152)."""
    del max_parallel
    return _backend.pkf(lgssm, observations, return_loglikelihood)


def pks(lgssm, ms, Ps, max_parallel=10000):
    """Smoothed means and covariances from filtered ones (parallel.py:187-196)."""
    del max_parallel
    return _backend.pks(lgssm, ms, Ps)


def pkfs(model, observations, max_parallel=10000):
    """Filter then smoother, one fused three-launch pass on the GPU (parallel.py:199-201)."""
    del max_parallel
    return _backend.pkfs(model, observations)
