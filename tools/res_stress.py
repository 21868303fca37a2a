#!/usr/bin/env python3
"""Stress of the resident launch's hand-offs (GPU box): random series lengths, steps per lane (8 / 16 / automatic), missing
stretches and -- the point -- time grids that are dense in some stretches (a filter that REMEMBERS across a workgroup there:
general fold behind the grid-wide wait) and sparse in others (carry from the neighbour's total alone), so that both roads and
both kinds of wait meet inside one launch.  Every output of the array form and of the fused form against the three launches
with the shortcut switched off; back-to-back launches on changing inputs.  Usage: python tools/res_stress.py [cases] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "parallel-gps_amd"))
from pssgp import _backend as B
from pssgp.kernels import Matern32


def relerr(a, b):
    a = np.asarray(a, float); b = np.asarray(b, float)
    return float(np.max(np.abs(a - b)) / max(1.0, float(np.max(np.abs(b)))))


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    ctx = B.get_context()
    worst = 0.0
    for c in range(cases):
        n = int(rng.choice([5000, 20000, 1 << 16, (1 << 17) + 333, 1 << 18, (1 << 19) - 7, 1 << 19, (1 << 19) + 4097, 1 << 20, (1 << 20) - 4095]))
        ls = float(rng.choice([0.3, 1.0, 5.0]))
        sde = Matern32(variance=1.0, lengthscales=ls).get_sde()
        dt = 0.05 * rng.uniform(0.5, 1.5, n)
        # dense stretches: a few windows of 3000 .. 30000 steps with steps a million times smaller
        for _ in range(int(rng.integers(0, 5))):
            a = int(rng.integers(0, n)); w = int(rng.integers(3000, 30000))
            dt[a:a + w] *= 1e-6
        t = np.cumsum(dt)
        Fs, Qs = B.discretise(sde.F, sde.P0, t, 0.0)
        y = np.sin(0.7 * t) + 0.3 * rng.standard_normal(n)
        for _ in range(int(rng.integers(0, 4))):
            a = int(rng.integers(0, n)); w = int(rng.integers(1, 9000))
            y[a:a + w] = np.nan
        ssm = (sde.P0, Fs, Qs, sde.H, np.array([[0.1]]))
        chunk = int(rng.choice([0, 8, 16])) if n <= (1 << 19) else int(rng.choice([0, 16]))
        ctx.set_resident(0); ctx.set_shortcut(0)
        want = B.pkfs(ssm, y, return_filtered=True, return_loglikelihood=True)
        ctx.set_resident(1); ctx.set_shortcut(1); ctx.set_chunk(chunk)
        try:
            fam = ctx.get_family(n, 2)
            got = B.pkfs(ssm, y, return_filtered=True, return_loglikelihood=True)
            got2 = B.pkfs(ssm, y, return_filtered=True, return_loglikelihood=True)          # back to back, other epoch
        finally:
            ctx.set_chunk(0)
        st = ctx.status()
        e = max(relerr(g, w) for g, w in zip(got, want))
        same = all(np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True) for a, b in zip(got, got2))
        worst = max(worst, e)
        print(f"case {c:3d} n {n:8d} ls {ls:4.1f} chunk {chunk:2d} family {fam:2d} status {st}  worst rel err {e:.2e}  repeat identical {same}", flush=True)
        assert st == 0 and e < 1e-9 and same, "MISMATCH"
    ctx.set_resident(-1); ctx.set_shortcut(1)
    print(f"{cases} cases, worst {worst:.2e}")


if __name__ == "__main__":
    main()
