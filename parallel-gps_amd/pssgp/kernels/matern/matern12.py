"""Matern-1/2 (Ornstein-Uhlenbeck), state dim 1 (reference: matern/matern12.py:8-23)."""
import numpy as np

from ..base import ContinuousDiscreteModel, Kernel, SDEKernelMixin, _pairwise_dist, get_lssm_spec
from .common import get_matern_sde


class Matern12(SDEKernelMixin, Kernel):
    def __init__(self, variance=1.0, lengthscales=1.0, **kwargs):
        self.variance = float(variance)
        self.lengthscales = float(lengthscales)
        SDEKernelMixin.__init__(self, **kwargs)

    def K(self, X, X2=None):
        return self.variance * np.exp(-_pairwise_dist(X, X2) / self.lengthscales)

    def get_spec(self, T):
        return get_lssm_spec(1, T)

    def get_sde(self):
        F, L, H, Q = get_matern_sde(self.variance, self.lengthscales, 1)
        return ContinuousDiscreteModel(np.array([[self.variance]]), F, L, H, Q)
