import os, sys, numpy as np
sys.path.insert(0, "parallel-gps_amd"); sys.path.insert(0, ".")
from oracle import np_oracle as O
from tests.conftest import make_times, relerr, sample_series
from tests.test_gpu_fuzz import _random_model, _ssm
from pssgp import _backend as B
def run(ctx, ssm, y, chunk):
    B._contexts[0] = ctx
    ctx.set_chunk(chunk)
    sms, sPs, fms, fPs, ll = B.pkfs(ssm, y, return_filtered=True, return_loglikelihood=True)
    return dict(fms=fms, fPs=fPs, sms=sms, sPs=sPs, ll=float(ll))
mask = sys.argv[1] if len(sys.argv) > 1 else "15"
dt = np.float32 if len(sys.argv) > 2 else np.float64
os.environ["PGPS_WC_ROWS2"] = mask
default = B.Context(0)
os.environ["PGPS_WC_ROWS2"] = "0"
tiles = B.Context(0)
del os.environ["PGPS_WC_ROWS2"]
print("mask", mask)
for d in (24, 32):
    rng = np.random.default_rng(4200 + d)
    F, P, H = _random_model(rng, d)
    for n, chunk in [(3, 2), (64, 16)]:
        t = make_times(n, seed=7 * d + n)
        ssm = _ssm(F, P, H, t, 0.2)
        y = sample_series(ssm, seed=n, nan_frac=0.0)
        ssm_ = tuple(np.asarray(q, dt) for q in ssm); y_ = y.astype(dt)
        a = run(default, ssm_, y_, chunk); b = run(tiles, ssm_, y_, chunk)
        of, oP, oll = O.kf(ssm, y, True); os_, osP = O.kfs(ssm, y)
        per = [relerr(a["fms"][k], of[k]) for k in range(min(n, 6))]
        perb = [relerr(b["fms"][k], of[k]) for k in range(min(n, 6))]
        print(f"d={d} n={n} chunk={chunk} rows2 fms/step {['%.1e' % e for e in per]} tiles {['%.1e' % e for e in perb]} "
              f"sms {relerr(a['sms'], os_):.1e}/{relerr(b['sms'], os_):.1e} ll {abs(a['ll']-oll):.1e}/{abs(b['ll']-oll):.1e}")
