"""Runs the real-data drivers (pssgp/experiments/real_data.py) at the reference's sizes on SYNTHETIC files written in the
reference's formats (its data files are not part of this repository): sunspot MAP fit + 96 000-point predict_f, a short
CO2 HMC at quasi-periodic order 2 (d = 14) and a shorter one at the reference's order 3 (d = 18).  Usage: python tools/real_data_demo.py"""
import json
import os
import pathlib
import sys
import tempfile

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "parallel-gps_amd"))
from pssgp.experiments import real_data as RD  # noqa: E402
from tests.test_experiments import _write_co2, _write_sunspots  # noqa: E402

import numpy as np  # noqa: E402

with tempfile.TemporaryDirectory() as d:
    p = pathlib.Path(d)
    _write_sunspots(p, n=3300, seed=0)
    print(json.dumps(RD.sunspot_map(d, n_training=3200)))
    # a CO2-like record of the reference's length: 2400 weekly + 900 monthly points
    rng = np.random.default_rng(1)
    tw = 1974.4 + np.arange(2400) / 52.18
    tm = 1958.2 + np.arange(900) / 12.0
    co2 = lambda t: 315.0 + 1.3 * (t - 1958.0) + 0.012 * (t - 1958.0) ** 2 + 3.0 * np.sin(2 * np.pi * t) + 0.3 * rng.standard_normal(t.shape)
    with open(p / "co2_weekly_mlo.txt", "w") as f:
        for t, v in zip(tw, co2(tw)):
            f.write(f"{int(t)} 1 1 {t:.4f} {v:.2f} 7 0.0 0.0 0.0\n")
    with open(p / "co2_mm_mlo.txt", "w") as f:
        for t, v in zip(tm, co2(tm)):
            f.write(f"{int(t)} 1 {t:.4f} {v:.2f} {v:.2f} 30 0.1 0.1\n")
    print(json.dumps(RD.co2_hmc(d, n_training=3192, qp_order=2, n_samples=40, n_burnin=20, step_size=0.002)))
    # the reference's own order (co2/mcmc.py:42-65): d = 18, one device evaluation per finite-difference point
    print(json.dumps(RD.co2_hmc(d, n_training=3192, qp_order=3, n_samples=8, n_burnin=4, step_size=0.002)))
