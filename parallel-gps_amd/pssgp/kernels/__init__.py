from .matern import Matern12, Matern32, Matern52  # noqa: F401
from .periodic import Periodic, SquaredExponential  # noqa: F401
from .rbf import RBF  # noqa: F401
