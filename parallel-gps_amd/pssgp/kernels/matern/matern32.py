"""Matern-3/2, state dimension 2, Pinf = diag(s2, lam^2 s2) in closed form: the p = 1 member of MaternFamily (common.py).\nReference semantics: pssgp/kernels/matern/matern32.py:10-28."""
from ..base import ContinuousDiscreteModel, Kernel, SDEKernelMixin, _pairwise_dist, get_lssm_spec
from .common import MaternFamily


class Matern32(MaternFamily, SDEKernelMixin, Kernel):
    state_dim = 2

    def __init__(self, variance=1.0, lengthscales=1.0, **kwargs):
        self._init_matern(variance, lengthscales, kwargs)
        SDEKernelMixin.__init__(self, **kwargs)

    def K(self, X, X2=None):
        return self._matern_K(_pairwise_dist(X, X2))

    def get_spec(self, T):
        return get_lssm_spec(self.state_dim, T)

    def get_sde(self):
        return ContinuousDiscreteModel(*self._matern_sde())
