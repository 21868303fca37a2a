"""Matern-nu SDE in companion form (reference: pssgp/kernels/matern/common.py:10-52).

For nu = d - 1/2 the spectral density factorises as (lambda + i w)^-d, giving
  F = shift matrix with last row  -C(d,k) lambda^(d-k), k = 0..d-1      (common.py:10-18)
  L = e_d,  H = e_1^T
  q = (2 lambda)^(2d-1) sigma^2 ((d-1)!)^2 / (2d-2)!                    (common.py:21-23)
with lambda = sqrt(2d-1) / lengthscale                                   (common.py:45).
"""
import math

import numpy as np


def get_matern_sde(variance, lengthscales, d):
    lam = math.sqrt(2 * d - 1) / float(lengthscales)
    F = np.diag(np.ones(d - 1), k=1)
    for k in range(d):
        F[d - 1, k] -= math.comb(d, k) * lam ** (d - k)
    L = np.zeros((d, 1))
    L[d - 1, 0] = 1.0
    H = np.zeros((1, d))
    H[0, 0] = 1.0
    q = (2.0 * lam) ** (2 * d - 1) * float(variance) * math.factorial(d - 1) ** 2 \
        / math.factorial(2 * d - 2)
    return F, L, H, np.array([[q]])


class MaternFamily:
    """What the three half-integer Matern kernels share, written once: nu = p + 1/2 has state dimension p + 1, covariance
    sigma^2 exp(-r) poly_p(r) with r = sqrt(2 nu) |t - t'| / l, the companion-form SDE above, and a stationary covariance
    that is either known in closed form (p = 0, 1) or comes out of the Lyapunov equation of the balanced model (p = 2, as
    the reference does it).  Subclasses set `state_dim` and, for the balanced ones, `balanced = True`."""
    state_dim = None
    balanced = False
    # poly_p(r) coefficients by ascending power of r
    _POLY = {1: (1.0,), 2: (1.0, 1.0), 3: (1.0, 1.0, 1.0 / 3.0)}

    def _init_matern(self, variance, lengthscales, kwargs):
        if self.balanced:
            from ... import config as pssgp_config
            self._balancing_iter = kwargs.pop('balancing_iter', pssgp_config.NUMBER_OF_BALANCING_STEPS)
        self.variance = float(variance)
        self.lengthscales = float(lengthscales)

    def _matern_K(self, dist):
        r = math.sqrt(2 * self.state_dim - 1) * dist / self.lengthscales
        poly = sum(c * r ** k for k, c in enumerate(self._POLY[self.state_dim]))
        return self.variance * poly * np.exp(-r)

    def _matern_sde(self):
        """(Pinf, F, L, H, Q) in the layout of ContinuousDiscreteModel."""
        p1 = self.state_dim
        F, L, H, Q = get_matern_sde(self.variance, self.lengthscales, p1)
        if self.balanced:
            from ..math_utils import balance_ss, solve_lyap_vec
            F, L, H, Q = balance_ss(F, L, H, Q, n_iter=self._balancing_iter)
            return solve_lyap_vec(F, L, Q), F, L, H, Q
        lam = math.sqrt(2 * p1 - 1) / self.lengthscales
        # derivatives of a stationary process of this family are uncorrelated with it at equal times up to p = 1:
        # Pinf = sigma^2 diag(1, lam^2, ...)
        return np.diag([self.variance * lam ** (2 * i) for i in range(p1)]), F, L, H, Q
