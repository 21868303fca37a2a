import sys, time, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/parallel-gps_amd")
from pssgp import _backend as B
from pssgp.kernels import Matern32, Matern52, Periodic, SquaredExponential, RBF
from oracle import np_oracle as O
k = Periodic(SquaredExponential(1., 1.), period=1., order=1) * Matern32(1., 1.) + Matern52(1., 1.)
sde = k.get_sde()
for name, t in (("regular 2^20", np.linspace(0.0, 4000.0, 1 << 20)), ("jittered 2^20", np.cumsum(0.004 * np.random.default_rng(0).uniform(0.5, 1.5, 1 << 20))),
                ("regular 5000", np.linspace(0.0, 20.0, 5000))):
    y = np.sin(t) + 0.1 * np.random.default_rng(1).standard_normal(t.size)
    B.lti_ll(sde.F, sde.P0, sde.H, 0.1, t, y)
    t0 = time.perf_counter()
    for _ in range(5): ll = B.lti_ll(sde.F, sde.P0, sde.H, 0.1, t, y)
    dt_ = (time.perf_counter() - t0) / 5 * 1e3
    Fs, Qs = B.discretise(sde.F, sde.P0, t, 0.0)
    ssm = O.get_ssm(sde, t[:3000], 0.1)
    e = max(np.max(np.abs(Fs[:3000] - ssm[1])), np.max(np.abs(Qs[:3000] - ssm[2])))
    print(f"{name:15s} lti_ll {dt_:7.3f} ms  ll {ll:.6f}  discretisation vs oracle (first 3000 steps) max abs err {e:.2e}", flush=True)
