import sys, time, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/parallel-gps_amd")
from pssgp.kernels import Matern32
from pssgp.model import StateSpaceGP
from pssgp import _backend as B
rng = np.random.default_rng(0)
n, k = 4096, 1024
t = np.sort(rng.uniform(0, 40, n)); y = np.sin(t) + 0.3 * rng.standard_normal(n)
tq = np.sort(rng.uniform(0, 40, k))[:, None]
gp = StateSpaceGP((t[:, None], y[:, None]), Matern32(1., 0.5), noise_variance=0.1, parallel=True)
ctx = B.get_context()
def bench(fn, reps=200):
    for _ in range(20): fn()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t0) / reps * 1e6
for thr in (-1, 0, 2048, 8192, 65536):
    ctx.set_one_launch(thr)
    gp._ll_memo = None
    a = bench(gp.maximum_log_likelihood_objective)
    def pf():
        gp._ll_memo = None
        return gp.predict_f(tq)
    b = bench(pf)
    c = bench(gp.log_likelihood_and_grad)
    print(f"one_launch={thr:6d}: ll {a:6.1f}  predict_f {b:6.1f}  ll+grad {c:6.1f} us", flush=True)
ctx.set_one_launch(-1)
for limit in (0, -1):
    ctx.set_grad_pack(limit)
    print(f"grad_pack={limit:3d}: ll+grad {bench(gp.log_likelihood_and_grad):6.1f} us", flush=True)
ctx.set_grad_pack(-1)
# host-side share: the same calls with the device work removed is not possible; time the pure-Python part instead
ser = gp._device_series()
fused, _ = gp._device_forms()
packed = gp._packed_fused(fused)
print("python: _device_forms+_packed_fused (memo hit)", bench(lambda: gp._packed_fused(gp._device_forms()[0])), "us")
print("python: ser.gp_ll only", bench(lambda: ser.gp_ll(packed, 0.1)), "us")
ser.set_queries(tq.reshape(-1))
print("python: ser.gp_predict only", bench(lambda: ser.gp_predict(packed, 0.1)), "us")
print("python: set_queries (same grid)", bench(lambda: ser.set_queries(tq.reshape(-1))), "us")

# RBF order 6 (general-LTI path): same setting vs a new setting every call
from pssgp.kernels import RBF
n = 1000
t = np.sort(rng.uniform(0, 10, n)); y = np.sin(t) + 0.3 * rng.standard_normal(n)
gp = StateSpaceGP((t[:, None], y[:, None]), RBF(1., 0.5, order=6, balancing_iter=5), noise_variance=0.1, parallel=True)
state = {"i": 0}
def fresh(fn):
    def run():
        state["i"] += 1
        gp.kernel.lengthscales = 0.5 * (1.0 + 1e-6 * state["i"])
        return fn()
    return run
print(f"RBF6 N={n}: same setting  ll {bench(gp.maximum_log_likelihood_objective, 100):7.1f}   ll+grad {bench(gp.log_likelihood_and_grad, 50):7.1f} us")
print(f"RBF6 N={n}: new setting   ll {bench(fresh(gp.maximum_log_likelihood_objective), 100):7.1f}   ll+grad {bench(fresh(gp.log_likelihood_and_grad), 50):7.1f} us")
gp._rbf_ref = None
ll_scaled = float(gp.maximum_log_likelihood_objective())
sde = gp.kernel.get_sde()
ll_direct = B.lti_ll(sde.F, sde.P0, sde.H, gp.noise_variance, t, y)
print("RBF6 ll through the scaled realisation vs get_sde's:", ll_scaled, ll_direct, abs(ll_scaled - ll_direct) / abs(ll_direct))
