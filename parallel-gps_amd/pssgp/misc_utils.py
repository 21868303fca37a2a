"""pssgp/misc_utils.py of the reference: rmse (10-15) and the 95 % band plot helper (18-28), on numpy arrays."""
import numpy as np


def rmse(x1, x2):
    """Root mean square error of two arrays of the same number of elements."""
    a, b = np.asarray(x1, np.float64).reshape(-1), np.asarray(x2, np.float64).reshape(-1)
    return float(np.sqrt(np.mean(np.square(a - b))))


def error_shade(t, m, cov, **kwargs):
    """Shade mean +- 1.96 standard deviations on the current matplotlib axes (needs matplotlib)."""
    import matplotlib.pyplot as plt
    t, m, cov = (np.asarray(a, np.float64).reshape(-1) for a in (t, m, cov))
    half = 1.96 * np.sqrt(cov)
    return plt.fill_between(t, m - half, m + half, **kwargs)
