"""State-space approximation of the squared-exponential kernel.

Reference: pssgp/kernels/rbf.py:14-101.  The SE spectral density
S(w) = s2 l sqrt(2 pi) exp(-l^2 w^2 / 2) is replaced by sqrt(2 pi) / P(w) with
P(w) = sum_{k<=order} (w^2/2)^k / k!  (Taylor of exp(w^2/2)); P(w) as a polynomial in
s = i w is spectrally factorised, the stable half gives a companion-form F of
dimension `order`, which is then rescaled by the lengthscale, balanced, and P_inf comes
from the Lyapunov equation.
"""
import math

import numpy as np

from .. import config as pssgp_config
from .base import ContinuousDiscreteModel, Kernel, SDEKernelMixin, _pairwise_dist, get_lssm_spec
from .math_utils import balance_ss, solve_lyap_vec


def _get_unscaled_rbf_sde(order=6):
    """Unit-lengthscale, unit-variance (F, L, H, q) of dimension `order` (rbf.py:14-61)."""
    # P(w) written in s = i w:  w^(2k) = (-1)^k s^(2k);  highest power first for np.roots
    in_s = np.zeros(2 * order + 1)
    for k in range(order + 1):
        in_s[2 * (order - k)] = (-1.0) ** k * 0.5 ** k / math.factorial(k)
    q = math.sqrt(2.0 * math.pi)                       # sqrt(2 pi) / P(0), P(0) = 1
    roots = np.roots(in_s)
    stable = np.real(np.poly(roots[np.real(roots) < 0]))   # monic, degree = order
    gain = stable[-1] / stable[0]                      # numerator that makes H F^-1 L unit-DC
    stable = stable / stable[0]
    n = stable.size - 1
    F = np.diag(np.ones(n - 1), k=1)
    F[-1, :] = -stable[:0:-1]
    L = np.zeros((n, 1))
    L[-1, 0] = 1.0
    H = np.zeros((1, n))
    H[0, 0] = gain
    return F, L, H, q


class RBF(SDEKernelMixin, Kernel):
    """`order` = state dimension (default 3); `balancing_iter` sweeps of balancing."""

    def __init__(self, variance=1.0, lengthscales=1.0, **kwargs):
        self._order = kwargs.pop('order', 3)
        self._balancing_iter = kwargs.pop('balancing_iter', pssgp_config.NUMBER_OF_BALANCING_STEPS)
        self.variance = float(variance)
        self.lengthscales = float(lengthscales)
        SDEKernelMixin.__init__(self, **kwargs)

    def K(self, X, X2=None):
        r = _pairwise_dist(X, X2) / self.lengthscales
        return self.variance * np.exp(-0.5 * r * r)

    def get_spec(self, T):
        return get_lssm_spec(self._order, T)

    def get_sde(self):
        F, L, H, q = _get_unscaled_rbf_sde(self._order)
        dim = F.shape[0]
        ell = self.lengthscales
        # time rescaling t -> t / ell of a companion form (rbf.py:90-95)
        F[-1, :] = F[-1, :] / ell ** np.arange(dim, 0, -1)
        H = H / ell ** dim
        Q = np.array([[self.variance * ell * q]])
        Fb, Lb, Hb, Qb = balance_ss(F, L, H, Q, n_iter=self._balancing_iter)
        Pinf = solve_lyap_vec(Fb, Lb, Qb)
        return ContinuousDiscreteModel(Pinf, Fb, Lb, Hb, np.reshape(Qb, (1, 1)))
