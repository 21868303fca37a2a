"""The data contract between kernels and filters (pssgp/kalman/base.py:3)."""
from collections import namedtuple

LGSSM = namedtuple("LGSSM", ["P0", "Fs", "Qs", "H", "R"])
