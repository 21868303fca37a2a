"""Matern-5/2, state dim 3; balanced, P_inf from the Lyapunov solve
(reference: matern/matern52.py:10-25)."""
import math

import numpy as np

from ... import config as pssgp_config
from ..base import ContinuousDiscreteModel, Kernel, SDEKernelMixin, _pairwise_dist, get_lssm_spec
from ..math_utils import balance_ss, solve_lyap_vec
from .common import get_matern_sde


class Matern52(SDEKernelMixin, Kernel):
    def __init__(self, variance=1.0, lengthscales=1.0, **kwargs):
        self._balancing_iter = kwargs.pop('balancing_iter', pssgp_config.NUMBER_OF_BALANCING_STEPS)
        self.variance = float(variance)
        self.lengthscales = float(lengthscales)
        SDEKernelMixin.__init__(self, **kwargs)

    def K(self, X, X2=None):
        r = math.sqrt(5.0) * _pairwise_dist(X, X2) / self.lengthscales
        return self.variance * (1.0 + r + r * r / 3.0) * np.exp(-r)

    def get_spec(self, T):
        return get_lssm_spec(3, T)

    def get_sde(self):
        F, L, H, q = get_matern_sde(self.variance, self.lengthscales, 3)
        Fb, Lb, Hb, Qb = balance_ss(F, L, H, q, n_iter=self._balancing_iter)
        Pinf = solve_lyap_vec(Fb, Lb, Qb)
        return ContinuousDiscreteModel(Pinf, Fb, Lb, Hb, Qb)
