import os, sys, time, json
import numpy as np
sys.path.insert(0, "/root/repo/parallel-gps_amd")
from pssgp.model import StateSpaceGP
from pssgp.experiments.real_data import co2_covariance
N = 1 << 17
rng = np.random.default_rng(0)
t = np.cumsum(rng.uniform(0.5, 1.5, N)) * (1.0 / 52.0)
y = 0.3 * np.sin(2 * np.pi * t) + 0.01 * t + 0.05 * rng.standard_normal(N)
gp = StateSpaceGP((t[:, None], y[:, None]), co2_covariance(3), 0.05, parallel=True)
for _ in range(3):
    gp.maximum_log_likelihood_objective()
t0 = time.perf_counter(); sde = gp.kernel.get_sde(); print("get_sde ms", (time.perf_counter() - t0) * 1e3)
import cProfile, pstats
cProfile.run("gp.kernel.get_sde()", "/tmp/prof.out")
pstats.Stats("/tmp/prof.out").sort_stats("cumtime").print_stats(12)
