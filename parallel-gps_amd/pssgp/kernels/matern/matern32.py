"""Matern-3/2, state dim 2; closed-form P_inf = diag(s2, lambda^2 s2)
(reference: matern/matern32.py:10-28)."""
import math

import numpy as np

from ..base import ContinuousDiscreteModel, Kernel, SDEKernelMixin, _pairwise_dist, get_lssm_spec
from .common import get_matern_sde


class Matern32(SDEKernelMixin, Kernel):
    def __init__(self, variance=1.0, lengthscales=1.0, **kwargs):
        self.variance = float(variance)
        self.lengthscales = float(lengthscales)
        SDEKernelMixin.__init__(self, **kwargs)

    def K(self, X, X2=None):
        r = math.sqrt(3.0) * _pairwise_dist(X, X2) / self.lengthscales
        return self.variance * (1.0 + r) * np.exp(-r)

    def get_spec(self, T):
        return get_lssm_spec(2, T)

    def get_sde(self):
        F, L, H, Q = get_matern_sde(self.variance, self.lengthscales, 2)
        lam = math.sqrt(3.0) / self.lengthscales
        P_infty = np.diag([self.variance, lam ** 2 * self.variance])
        return ContinuousDiscreteModel(P_infty, F, L, H, Q)
