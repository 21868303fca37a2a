"""Where does one derivative direction per model (launch_grad_pack) stop paying against all directions in one dual
number?  Matern-3/2 (d = 2) and Matern-1/2 (d = 1), wall-clock per log_likelihood_and_grad call."""
import sys, time, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "parallel-gps_amd"))
from pssgp import _backend as B
from pssgp.kernels import Matern12, Matern32
from pssgp.model import StateSpaceGP
ctx = B.get_context()
for K in (Matern32, Matern12):
    for log2n in (12, 15, 16, 17, 18, 20):
        n = 1 << log2n
        rng = np.random.default_rng(0)
        t = np.cumsum(0.01 * rng.uniform(0.5, 1.5, n)); y = np.sin(t) + 0.3 * rng.standard_normal(n)
        gp = StateSpaceGP((t[:, None], y[:, None]), K(1., 0.5), noise_variance=0.1, parallel=True)
        out = []
        for lim in (0, 1 << 30):
            ctx.set_grad_pack(lim)
            for _ in range(5): gp.log_likelihood_and_grad()
            t0 = time.perf_counter()
            for _ in range(30): gp.log_likelihood_and_grad()
            out.append((time.perf_counter() - t0) / 30 * 1e6)
        ctx.set_grad_pack(-1)
        print(f"{K.__name__} N=2^{log2n}: all directions in one dual {out[0]:8.1f} us   one direction per model {out[1]:8.1f} us", flush=True)
