"""The reference's import path for `rmse` (`from pssgp.misc_utils import rmse`, pssgp/misc_utils.py:10-15; used by
experiments/toy_models/speed_and_stability.py).  Plotting (`error_shade`) is outside this package: it needs matplotlib,
which the product does not depend on."""
from .experiments.toy import rmse

__all__ = ["rmse", "error_shade"]


def error_shade(t, m, cov, **kwargs):
    """Mean +- 1.96 standard deviations as a shaded band on the current matplotlib axes (pssgp/misc_utils.py:18-28)."""
    try:
        import matplotlib.pyplot as plt
    except ImportError as e:                # noqa: F841
        raise ImportError("pssgp.misc_utils.error_shade draws with matplotlib, which is not installed") from e
    import numpy as np
    t, m, cov = np.asarray(t).reshape(-1), np.asarray(m).reshape(-1), np.asarray(cov).reshape(-1)
    half = 1.96 * np.sqrt(np.maximum(cov, 0.0))
    return plt.fill_between(t, m - half, m + half, **kwargs)
