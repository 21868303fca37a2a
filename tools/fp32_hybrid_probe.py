import os, sys, numpy as np
ROOT=os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "parallel-gps_amd"))
from oracle import c_oracle as C
from pssgp import _backend as B
from pssgp.kernels import Matern32, Matern52, RBF
from tests.conftest import relerr, sample_series_fast
ctx = B.get_context()
for kname, k in (("rbf6", RBF(1.,1.,order=6,balancing_iter=10)), ("m32", Matern32(1.,1.)), ("m52", Matern52(1.,1.)), ("rbf8", RBF(1.,1.,order=8,balancing_iter=10))):
    sde = k.get_sde()
    for n in (32768, 1<<20):
        t = np.linspace(0.0, 4.0, n)
        Fs, Qs = B.discretise(sde.F, sde.P0, t, 0.0)
        ssm = (sde.P0, Fs, Qs, np.asarray(sde.H).reshape(1,-1), np.array([[0.1]]))
        y = sample_series_fast(ssm, seed=n % 89, nan_frac=0.1)
        cf, cP, cs, csP, cll = C.kfs(ssm, y, np.float64)
        ssm32 = tuple(np.asarray(a, np.float32) for a in ssm)
        ctx.set_f32_policy(1)
        fms, fPs, ll = B.pkf(ssm32, y.astype(np.float32), return_loglikelihood=True)
        ctx.set_f32_policy(2)
        sms, sPs = B.pks(ssm32, fms, fPs)
        ctx.set_f32_policy(0)
        print(f"{kname} N={n}: float32 filter fm {relerr(fms, cf):.1e} fP {relerr(fPs, cP):.1e} ll {abs(float(ll)-cll)/abs(cll):.1e} | fp64 smoother on them: sm {relerr(sms, cs):.1e} sP {relerr(sPs, csP):.1e}", flush=True)
