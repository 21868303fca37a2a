"""Global knobs, mirroring pssgp/config.py:6-16 of the reference.

`NUMBER_OF_BALANCING_STEPS` is read by the composite kernels (sum / product) and is the
default `balancing_iter` of Matern52 and RBF.  `default_float` replaces
`gpflow.config.default_float()` (the reference selects fp64/fp32 through GPflow,
pssgp/experiments/toy_models/speed_and_stability.py:68).
"""
import numpy as np

NUMBER_OF_BALANCING_STEPS = 10

_DEFAULT_FLOAT = np.float64


def set_number_balancing_steps(n_balancing_steps):
    """Set the default number of balancing sweeps (pssgp/config.py:9-16)."""
    global NUMBER_OF_BALANCING_STEPS
    NUMBER_OF_BALANCING_STEPS = int(n_balancing_steps)


def default_float():
    return _DEFAULT_FLOAT


def set_default_float(dtype):
    """np.float64 (default) or np.float32: the dtype the HIP scan computes in."""
    global _DEFAULT_FLOAT
    dtype = np.dtype(dtype).type
    if dtype not in (np.float32, np.float64):
        raise ValueError("default float must be float32 or float64")
    _DEFAULT_FLOAT = dtype
