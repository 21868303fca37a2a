"""Resident one-launch filter + smoother (csrc/pgps_resident.hip.h) against the three-launch path and the numpy oracle,
array form (pkfs) and fused form (gp), whole and ragged lengths, with missing observations; then timings of both roads
on device-resident arrays and the resident kernel's phase stamps.   python tools/res_check.py [--time]"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "parallel-gps_amd")]
from oracle import np_oracle as O           # noqa: E402
from pssgp import _backend as B             # noqa: E402
from pssgp.kernels import Matern32          # noqa: E402


def series(n, seed=0, nan_frac=0.2):
    rng = np.random.default_rng(seed)
    t = np.cumsum(0.05 * rng.uniform(0.5, 1.5, n))
    y = np.sin(t) + 0.3 * rng.standard_normal(n)
    y[rng.uniform(size=n) < nan_frac] = np.nan
    return t, y


def relerr(a, b):
    return float(np.max(np.abs(a - b)) / max(1.0, float(np.max(np.abs(b)))))


def main():
    ctx = B.get_context(0)
    kern = Matern32(variance=1.0, lengthscales=1.0)
    sde = kern.get_sde()
    form = B.nilpotent_form(sde.F)
    worst = 0.0
    for n in (1, 3, 64, 1000, 4096, 4097, 5000, 3 * 4096 + 17, 65536, 100000, 1 << 17, (1 << 20) - 5, 1 << 20):
        t, y = series(n, seed=n)
        ssm = tuple(np.asarray(a, np.float64) for a in O.get_ssm(sde, t, 0.1))
        ctx.set_resident(0)
        ref = B.pkfs(ssm, y, return_filtered=True, return_loglikelihood=True)
        gref = B.gp(form, sde.P0, sde.H.reshape(-1), 0.1, t, y, want_filtered=True, want_smoothed=True)
        ctx.set_resident(1)
        got = B.pkfs(ssm, y, return_filtered=True, return_loglikelihood=True)
        ggot = B.gp(form, sde.P0, sde.H.reshape(-1), 0.1, t, y, want_filtered=True, want_smoothed=True)
        st = ctx.status()
        errs = [relerr(g, r) for g, r in zip(got[:4], ref[:4])] + [abs(float(got[4]) - float(ref[4])) / max(1.0, abs(float(ref[4])))]
        gerrs = [relerr(ggot[k], gref[k]) for k in ("sms", "sPs", "fms", "fPs")] + [abs(float(ggot["ll"]) - float(gref["ll"])) / max(1.0, abs(float(gref["ll"])))]
        line = f"N={n:8d} status={st} array vs 3-launch: " + " ".join(f"{e:.1e}" for e in errs) + " | fused: " + " ".join(f"{e:.1e}" for e in gerrs)
        if n <= 5000:
            fm_o, fP_o, ll_o = O.kf(ssm, y, True)
            sm_o, sP_o = O.kfs(ssm, y)
            oerrs = [relerr(got[0], sm_o), relerr(got[1], sP_o), relerr(got[2], fm_o), relerr(got[3], fP_o), abs(float(got[4]) - ll_o) / max(1.0, abs(ll_o))]
            line += " | array vs oracle: " + " ".join(f"{e:.1e}" for e in oerrs)
            errs += oerrs
        print(line, flush=True)
        worst = max(worst, *errs, *gerrs)
        assert st == 0
    print("worst", worst)
    assert worst < 1e-9, worst
    if "--time" in sys.argv:
        timing(ctx, sde, form)


def timing(ctx, sde, form):
    n = 1 << 20
    t, y = series(n, seed=1)
    ssm = tuple(np.asarray(a, np.float64) for a in O.get_ssm(sde, t, 0.1))
    P0, Fs, Qs, H, R = ssm
    arrs = dict(P0=P0.reshape(-1), Fs=Fs.reshape(-1), Qs=Qs.reshape(-1), H=np.asarray(H, np.float64).reshape(-1), ys=y, ts=t)
    dev = {}
    for k, v in arrs.items():
        v = np.ascontiguousarray(v, np.float64)
        dev[k] = ctx.malloc(v.nbytes)
        ctx.h2d(dev[k], v)
    for k, sz in (("fms", 2), ("fPs", 4), ("sms", 2), ("sPs", 4)):
        dev[k] = ctx.malloc(n * sz * 8)
    dev["ll"] = ctx.malloc(8)
    lam, N1, N2 = form
    N1 = np.ascontiguousarray(N1, np.float64); N2 = np.ascontiguousarray(N2, np.float64)
    Pinf = np.ascontiguousarray(sde.P0, np.float64); Hh = np.ascontiguousarray(sde.H.reshape(-1), np.float64)
    from ctypes import c_long, c_int, c_double

    def run_array():
        ctx.call("pgps_pkfs_dev_f64", c_long(n), c_int(2), dev["P0"], dev["Fs"], dev["Qs"], dev["H"], c_double(float(np.asarray(R).reshape(-1)[0])), dev["ys"],
                 dev["fms"], dev["fPs"], dev["sms"], dev["sPs"], dev["ll"])

    def run_fused():
        ctx.call("pgps_gp_dev_f64", c_long(n), c_int(2), c_double(lam), B._ptr(N1), B._ptr(N2), B._ptr(Pinf), B._ptr(Hh), c_double(0.1),
                 dev["ts"], c_double(0.0), dev["ys"], dev["fms"], dev["fPs"], dev["sms"], dev["sPs"], dev["ll"])

    for name, fn in (("array", run_array), ("fused", run_fused)):
        for mode in (0, 1, 0, 1):
            ctx.set_resident(mode)
            for _ in range(20):
                fn()
            ctx.synchronize()
            t0 = time.perf_counter()
            reps = 200
            for _ in range(reps):
                fn()
            ctx.synchronize()
            dt = (time.perf_counter() - t0) / reps
            print(f"{name} resident={mode}: {dt * 1e6:.1f} us per pass", flush=True)
    for name, fn in (("array", run_array), ("fused", run_fused)):
        ctx.set_resident(2)
        for _ in range(5):
            fn()
        ctx.synchronize()
        st = ctx.resident_stamps()
        d = np.diff(st[:, :10], axis=1)
        names = ["load+reduce", "block scan", "publish+barrier1", "fold+apply", "kalman pass", "ll+suffix scan", "publish+barrier2",
                 "fold+apply", "rts pass"]
        print(f"{name}: phase stamps (cycles; median / max over {st.shape[0]} workgroups), total median {np.median(st[:, 9] - st[:, 0]):.0f}"
              f" first-start..last-end {st[:, 9].max() - st[:, 0].min()}")
        for i, nm in enumerate(names):
            print(f"    {nm:18s} {np.median(d[:, i]):9.0f} {d[:, i].max():9.0f}")
    ctx.set_resident(-1)


if __name__ == "__main__":
    main()
