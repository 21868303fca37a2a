"""Matern-1/2 (Ornstein-Uhlenbeck), state dimension 1: the p = 0 member of MaternFamily (common.py).\nReference semantics: pssgp/kernels/matern/matern12.py:8-23."""
from ..base import ContinuousDiscreteModel, Kernel, SDEKernelMixin, _pairwise_dist, get_lssm_spec
from .common import MaternFamily


class Matern12(MaternFamily, SDEKernelMixin, Kernel):
    state_dim = 1

    def __init__(self, variance=1.0, lengthscales=1.0, **kwargs):
        self._init_matern(variance, lengthscales, kwargs)
        SDEKernelMixin.__init__(self, **kwargs)

    def K(self, X, X2=None):
        return self._matern_K(_pairwise_dist(X, X2))

    def get_spec(self, T):
        return get_lssm_spec(self.state_dim, T)

    def get_sde(self):
        return ContinuousDiscreteModel(*self._matern_sde())
