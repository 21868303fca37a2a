"""fp32 error of the filter / smoother on the reference's benchmark grid (np.linspace(0, 4, N)), per kernel family,
against the fp64 C oracle on the same model: max-norm relative errors of the moments and of the log-likelihood, and the
condition numbers that explain them.  Diagnostic (GPU box): python tools/fp32_grid_errors.py [kernel ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "parallel-gps_amd"))
from oracle import c_oracle as C                          # noqa: E402  (the checker, not the product)
from pssgp import _backend as B                           # noqa: E402
from pssgp.kernels import Matern32, Matern52, RBF         # noqa: E402
from tests.conftest import relerr, sample_series_fast     # noqa: E402

KERNELS = {"m32": lambda: Matern32(1., 1.), "m52": lambda: Matern52(1., 1.),
           "rbf6": lambda: RBF(1., 1., order=6, balancing_iter=10), "rbf8": lambda: RBF(1., 1., order=8, balancing_iter=10)}


def main():
    names = sys.argv[1:] or ["m32", "m52", "rbf6"]
    ctx = B.get_context()
    print("kernel      N  family   fms       fPs       sms       sPs       ll       cond(P_pred) at N/2   cond(P0)")
    for name in names:
        sde = KERNELS[name]().get_sde()
        d = np.asarray(sde.F).shape[0]
        for n in (4096, 32768, 1 << 20):
            t = np.linspace(0.0, 4.0, n)
            Fs, Qs = B.discretise(sde.F, sde.P0, t, 0.0)
            ssm = (sde.P0, Fs, Qs, np.asarray(sde.H).reshape(1, -1), np.array([[0.1]]))
            y = sample_series_fast(ssm, seed=n % 89, nan_frac=0.1)
            cf, cP, cs, csP, cll = C.kfs(ssm, y, np.float64)
            k = n // 2
            Pp = Fs[k] @ cP[k - 1] @ Fs[k].T + Qs[k]
            conds = (np.linalg.cond(Pp), np.linalg.cond(np.asarray(sde.P0)))
            qf, qP, qs, qsP, qll = C.kfs(ssm, y, np.float32)         # the sequential filter / RTS smoother in float32
            print(f"{name:6s} {n:7d}  seq32  {relerr(qf, cf):.2e}  {relerr(qP, cP):.2e}  {relerr(qs, cs):.2e}  "
                  f"{relerr(qsP, csP):.2e}  {abs(qll - cll) / abs(cll):.2e}   (the C oracle's float32 build)", flush=True)
            # "f32": float32 arithmetic whatever the grid (pgps_set_f32_policy 1: what rounds 1-3 did), by kernel family;
            # "auto": the default since round 4 -- dense grids are computed in fp64 on the float32 arrays
            fams = [(0, "f32"), (-1, "auto")] + ([(1, "lane")] if d <= 6 else []) + [(3, "rows")] + ([(4, "quad")] if 5 <= d <= 8 else [])
            for fam, label in fams:
                ctx.set_family(max(fam, 0))
                ctx.set_f32_policy(0 if fam < 0 else 1)
                try:
                    ssm32 = tuple(np.asarray(a, np.float32) for a in ssm)
                    sms, sPs, fms, fPs, ll = B.pkfs(ssm32, np.asarray(y, np.float32), return_filtered=True, return_loglikelihood=True)
                    print(f"{name:6s} {n:7d}  {label:5s}  {relerr(fms, cf):.2e}  {relerr(fPs, cP):.2e}  {relerr(sms, cs):.2e}  "
                          f"{relerr(sPs, csP):.2e}  {abs(float(ll) - cll) / abs(cll):.2e}   {conds[0]:.2e}   {conds[1]:.2e}", flush=True)
                except B.PgpsError as e:
                    print(f"{name:6s} {n:7d}  {label:5s}  {e}", flush=True)
                finally:
                    ctx.set_family(0)
                    ctx.set_f32_policy(0)


if __name__ == "__main__":
    main()
