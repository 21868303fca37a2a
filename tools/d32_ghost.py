#!/usr/bin/env python3
"""The round-3 'ghost': at the first two-rows commit (edfddd1) a float64 d = 32 reduce kernel was wrong by 1e-1 and the
error vanished with an unrelated change of its loads.  Run against a library built from that commit
(PGPS_LIB=/path/libpgps_edfddd1.so python tools/d32_ghost.py) and against the current one: random stable models of
d = 24..32, pkfs against the numpy oracle at several series lengths and chunk lengths, and -- PGPS_WC_ROWS2=0 in a second
context -- the LDS-tile kernels on the same inputs.  Prints the worst relative error per case."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "parallel-gps_amd"))
from oracle import np_oracle as O
from pssgp import _backend as B
from tests.conftest import make_times, relerr, sample_series
from tests.test_gpu_fuzz import _random_model, _ssm


def where(got, want, name):
    """Which time steps and which state components carry the error: the steps with any entry off by > 1e-8 of the array's
    scale, as runs, and the components wrong at the first such step."""
    got = np.asarray(got, float).reshape(got.shape[0], -1); want = np.asarray(want, float).reshape(got.shape)
    bad = np.abs(got - want) > 1e-8 * np.abs(want).max()
    steps = np.flatnonzero(bad.any(axis=1))
    if steps.size == 0:
        print(f"      {name}: no entry off by more than 1e-8 of the scale", flush=True)
        return
    cut = np.flatnonzero(np.diff(steps) > 1)
    runs = [(int(a), int(b)) for a, b in zip(np.r_[steps[0], steps[cut + 1]], np.r_[steps[cut], steps[-1]])]
    comps = np.flatnonzero(bad[steps[0]])
    print(f"      {name}: {steps.size} of {got.shape[0]} steps wrong, runs {runs[:6]}{' ...' if len(runs) > 6 else ''}; at step {steps[0]}: "
          f"{comps.size} of {got.shape[1]} entries, first {comps[:12].tolist()}", flush=True)


def run(ctxname, dims, cases):
    for d in dims:
        rng = np.random.default_rng(4200 + d)
        F, P, H = _random_model(rng, d)
        for n, chunk in cases:
            t = make_times(n, seed=7 * d + n)
            ssm = _ssm(F, P, H, t, 0.2)
            y = sample_series(ssm, seed=n, nan_frac=0.2)
            B.get_context().set_chunk(chunk)
            try:
                sms, sPs, fms, fPs, ll = B.pkfs(ssm, y, return_filtered=True, return_loglikelihood=True)
            except Exception as e:          # noqa: BLE001
                print(f"{ctxname} d={d} n={n} chunk={chunk}: {e}", flush=True)
                continue
            of, oP, oll = O.kf(ssm, y, True)
            os_, osP = O.kfs(ssm, y)
            errs = (relerr(fms, of), relerr(fPs, oP), relerr(sms, os_), relerr(sPs, osP), abs(float(ll) - oll) / abs(oll))
            flag = "  <-- WRONG" if max(errs) > 1e-6 else ""
            print(f"{ctxname} d={d} n={n} chunk={chunk}: fm {errs[0]:.1e} fP {errs[1]:.1e} sm {errs[2]:.1e} sP {errs[3]:.1e} ll {errs[4]:.1e}{flag}",
                  flush=True)
            if flag:
                where(fms, of, "fm")
                where(fPs.reshape(n, -1), oP.reshape(n, -1), "fP")
                where(sms, os_, "sm")
    B.get_context().set_chunk(0)


if __name__ == "__main__":
    print("library:", B._LIB_PATH, flush=True)
    dims = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [24, 27, 29, 31, 32]
    cases = [(33, 16), (700, 7), (1500, 0), (2600, 33), (1 << 13, 0), (1 << 15, 0)]
    run("two-rows", dims, cases)
