"""The names the reference's experiment scripts share (pssgp/experiments/common.py:21-71): the model / covariance /
sampler enumerations and the two factories.  `ModelEnum.GP` (GPflow's dense GPR) has no counterpart on this backend:
the dense GP exists only as the test oracle; asking for it raises.  All three samplers are implemented
(experiments/toy.py: hmc, mala_chain, nuts_chain; the real-data drivers pick one by --mcmc)."""
import enum

from ..kernels import Matern12, Matern32, Matern52, Periodic, RBF, SquaredExponential
from ..model import StateSpaceGP


class MCMC(enum.Enum):
    HMC = "HMC"
    MALA = "MALA"
    NUTS = "NUTS"


class ModelEnum(enum.Enum):
    GP = "GP"
    SSGP = "SSGP"
    PSSGP = "PSSGP"


class CovarianceEnum(enum.Enum):
    Matern12 = "Matern12"
    Matern32 = "Matern32"
    Matern52 = "Matern52"
    RBF = "RBF"
    QP = "QP"


_SIMPLE = {CovarianceEnum.Matern12: Matern12, CovarianceEnum.Matern32: Matern32, CovarianceEnum.Matern52: Matern52,
           CovarianceEnum.RBF: RBF}


def get_simple_covariance_function(covariance_enum, **kwargs):
    """Kernel by name; QP = Periodic over a squared-exponential base kernel that takes `variance` / `lengthscales`,
    the remaining keywords (period, order) go to Periodic (common.py:44-57)."""
    cov = CovarianceEnum(covariance_enum)
    if cov in _SIMPLE:
        return _SIMPLE[cov](**kwargs)
    kwargs = dict(kwargs)
    base = SquaredExponential(kwargs.pop("variance", 1.), kwargs.pop("lengthscales", 1.))
    return Periodic(base, **kwargs)


def get_model(model_enum, data, noise_variance, covariance_function, max_parallel=10000):
    """SSGP = sequential Kalman path on the host, PSSGP = parallel scans on the GPU (common.py:60-71)."""
    model = ModelEnum(model_enum)
    if model is ModelEnum.SSGP:
        return StateSpaceGP(data, covariance_function, noise_variance, parallel=False)
    if model is ModelEnum.PSSGP:
        return StateSpaceGP(data, covariance_function, noise_variance, parallel=True, max_parallel=max_parallel)
    raise NotImplementedError("ModelEnum.GP is GPflow's dense GPR; this backend has the state-space models only "
                              "(the dense GP lives in oracle/np_oracle.py as a test oracle)")
