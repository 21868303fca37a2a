from .base import LGSSM  # noqa: F401
