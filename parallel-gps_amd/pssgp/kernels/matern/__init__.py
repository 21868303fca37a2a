from .matern12 import Matern12  # noqa: F401
from .matern32 import Matern32  # noqa: F401
from .matern52 import Matern52  # noqa: F401
