"""A few StateSpaceGP calls under `rocprofv3 --marker-trace`: the roctx ranges libpgps opens (named after the reference's
tf.name_scope of the same work: parallel_filter, merge_sorted, make_model) show up in the marker summary."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "parallel-gps_amd"))
from pssgp.kernels import Matern32, RBF      # noqa: E402
from pssgp.model import StateSpaceGP         # noqa: E402

rng = np.random.default_rng(0)
t = np.cumsum(0.05 * rng.uniform(0.5, 1.5, 50000))
y = np.sin(t) + 0.3 * rng.standard_normal(t.size)
tq = np.sort(rng.uniform(t[0], t[-1], 5000))
for kern in (Matern32(1.0, 1.0), RBF(1.0, 1.0, order=6, balancing_iter=10)):
    m = StateSpaceGP((t[:, None], y[:, None]), kern, noise_variance=0.1, parallel=True)
    for _ in range(3):
        ll = float(m.maximum_log_likelihood_objective())
        mean, var = m.predict_f(tq[:, None])
    print(type(kern).__name__, ll, float(mean[0, 0]), float(var[0, 0]))
