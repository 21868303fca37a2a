"""State-space form of the periodic (exp-sine-squared) kernel.

Reference: pssgp/kernels/periodic.py:18-81.  k(r) = s2 exp(-2 sin^2(pi r / p) / l'^2) with
l' = 2 l (GPflow's Periodic(SE) has exp(-sin^2/(2 l^2)); hence the factor 2,
periodic.py:57) is expanded in cosines: k(r) = sum_j q2_j cos(j w0 r).  Each harmonic is a
2-state undamped oscillator: F = blockdiag(j w0 [[0,-1],[1,0]]), Q = 0,
P_inf = diag(q2_j) (x) I2, H = [1 0 1 0 ...].  State dim 2 (order + 1).
"""
import math

import numpy as np
from scipy.special import comb, factorial

from .base import ContinuousDiscreteModel, Kernel, SDEKernelMixin, _pairwise_dist, get_lssm_spec
from .math_utils import kron


class SquaredExponential(Kernel):
    """Base kernel handle for `Periodic` (stands in for gpflow.kernels.SquaredExponential)."""

    def __init__(self, variance=1.0, lengthscales=1.0):
        self.variance = float(variance)
        self.lengthscales = float(lengthscales)

    def K(self, X, X2=None):
        r = _pairwise_dist(X, X2) / self.lengthscales
        return self.variance * np.exp(-0.5 * r * r)


_OFFLINE_MEMO = {}


def _get_offline_coeffs(N):
    """(memoised by order: the tables hold no hyper-parameter -- an optimiser's or sampler's step re-reads them)"""
    got = _OFFLINE_MEMO.get(N)
    if got is None:
        got = _OFFLINE_MEMO[N] = _offline_coeffs(N)
        for a in got:
            a.setflags(write=False)
    return got


def _offline_coeffs(N):
    """Hyper-parameter-free tables (periodic.py:18-38): b[k, j] = 2 C(k, (k-j)/2) for
    j <= k with k - j even (halved at j = 0), K[k, j] = k, div_facto_K = 1 / K!."""
    idx = np.arange(N + 1)
    K = np.repeat(idx[:, None], N + 1, axis=1)
    J = K.T
    valid = (J <= K) & ((K - J) % 2 == 0)
    half = np.where(valid, (K - J) // 2, 0)
    b = np.where(valid, 2.0 * comb(K, half), 0.0)
    b[:, 0] *= 0.5
    return b, K, 1.0 / factorial(K)


class Periodic(SDEKernelMixin, Kernel):
    def __init__(self, base_kernel, period=1.0, **kwargs):
        assert isinstance(base_kernel, SquaredExponential), \
            "Only SquaredExponential is supported at the moment"
        self._order = kwargs.pop('order', 6)
        self.base_kernel = base_kernel
        self.period = float(period)
        SDEKernelMixin.__init__(self, **kwargs)

    def K(self, X, X2=None):
        s = np.sin(math.pi * _pairwise_dist(X, X2) / self.period) / self.base_kernel.lengthscales
        return self.base_kernel.variance * np.exp(-0.5 * s * s)

    def get_spec(self, T):
        return get_lssm_spec(2 * (self._order + 1), T)

    def get_sde(self):
        N = self._order
        w0 = 2.0 * math.pi / self.period
        ell = self.base_kernel.lengthscales * 2.0
        b, K, inv_fact = _get_offline_coeffs(N)
        terms = b * ell ** (-2.0 * K) * inv_fact * math.exp(-ell ** (-2.0)) * 2.0 ** (-K) \
            * self.base_kernel.variance
        q2 = np.sum(terms, axis=0)
        rot = np.array([[0.0, -w0], [w0, 0.0]])
        F = kron(np.diag(np.arange(N + 1, dtype=np.float64)), rot)
        dim = 2 * (N + 1)
        Pinf = kron(np.diag(q2), np.eye(2))
        H = kron(np.ones((1, N + 1)), np.array([[1.0, 0.0]]))
        return ContinuousDiscreteModel(Pinf, F, np.eye(dim), H, np.zeros((dim, dim)))
