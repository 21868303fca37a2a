#!/usr/bin/env python3
"""One-off extended fuzz campaign on the GPU box (not part of the test-suite): tests/test_gpu_fuzz.py with many more
seeds, plus random models of state dimension 17..32 (dense and block-diagonal) through the general-LTI entry points
and random sharded series (1..6 ranks, ragged boundaries, every kernel family) through the segment protocol.
Usage: python tools/fuzz_campaign.py [first_seed] [n_seeds] [only: comma-separated case names]"""
import os
import sys
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "parallel-gps_amd"))
from oracle import np_oracle as O  # noqa: E402
from tests import test_gpu_fuzz as T  # noqa: E402
from tests.conftest import make_times, sample_series  # noqa: E402


def large_d_case(seed):
    from pssgp import _backend as B
    rng = np.random.default_rng(5000 + seed)
    d = int(rng.integers(17, 33))
    if rng.random() < 0.5:                                   # block-diagonal: sizes 2..16
        sizes, left = [], d
        while left > 0:
            s = int(min(left, rng.integers(2, 17)))
            if left - s == 1:
                s += 1 if s < 16 else -1
            sizes.append(s)
            left -= s
        F, P = np.zeros((d, d)), np.zeros((d, d))
        o = 0
        for s in sizes:
            Fb, Pb, _ = T._random_model(rng, s)
            F[o:o + s, o:o + s], P[o:o + s, o:o + s] = Fb, Pb
            o += s
        H = rng.standard_normal((1, d))
        if rng.random() < 0.5:                               # hide the blocks behind a permutation
            p = rng.permutation(d)
            F, P, H = F[np.ix_(p, p)], P[np.ix_(p, p)], H[:, p]
        kind = f"blocks{sizes}"
    else:
        F, P, H = T._random_model(rng, d)
        kind = "dense"
    n = int(rng.choice([1, 2, 7, 33, 100, 700, 1500]))
    t = make_times(n, seed=seed)
    ssm = T._ssm(F, P, H, t, 0.2)
    y = sample_series(ssm, seed=seed, nan_frac=float(rng.choice([0.0, 0.3])) if n > 3 else 0.0)
    oll = O.kf(ssm, y, True)[2]
    tag = f"seed={seed} d={d} n={n} {kind}"
    ll = B.lti_ll(F, P, H.reshape(-1), 0.2, t, y)
    assert abs(ll - oll) <= 1e-8 * abs(oll) + 1e-12, (tag, ll, oll)
    k = int(rng.choice([1, 7, 90]))
    tq = np.sort(rng.uniform(0.0, t[-1] + 0.3, k))
    mean, var, _ = B.lti_predict(F, P, H.reshape(-1), 0.2, t, y, tq)
    all_t, all_y, flags = O.merge_sorted(t, tq, (y, np.full(tq.shape, np.nan)),
                                         (np.zeros(t.shape, bool), np.ones(tq.shape, bool)))
    ms, Ps = O.kfs(T._ssm(F, P, H, all_t, 0.2), all_y)
    h = H.reshape(-1)
    assert np.max(np.abs(mean - ms[flags] @ h)) < 1e-7 * max(1.0, float(np.max(np.abs(ms)))), tag
    assert np.max(np.abs(var - np.einsum("i,nij,j->n", h, Ps[flags], h))) < 1e-7 * max(1.0, float(np.max(np.abs(Ps)))), tag
    return tag


def two_rows_case(seed):
    """The array entry points at d = 17..32 (two-rows level-1 kernels, csrc/pgps_rc2.hip.h): random dimension, length, chunk
    length, missing fraction and precision against the numpy oracle -- filter + smoother + log-likelihood and the filter alone."""
    from pssgp import _backend as B
    from tests.conftest import relerr
    rng = np.random.default_rng(9000 + seed)
    d = int(rng.integers(17, 33))
    F, P, H = T._random_model(rng, d)
    n = int(rng.choice([1, 2, 3, 17, 33, 64, 65, 257, 700, 1500, 2600, 5000]))
    chunk = int(rng.choice([0, 1, 2, 3, 5, 16, 33, 64]))
    f32 = bool(rng.random() < 0.3)
    t = make_times(n, seed=seed)
    ssm = T._ssm(F, P, H, t, 0.2)
    y = sample_series(ssm, seed=seed, nan_frac=float(rng.choice([0.0, 0.2, 0.6])) if n > 3 else 0.0)
    tag = f"seed={seed} d={d} n={n} chunk={chunk} {'f32' if f32 else 'f64'}"
    of, oP, oll = O.kf(ssm, y, True)
    os_, osP = O.kfs(ssm, y)
    ctx = B.get_context()
    ctx.set_chunk(chunk)
    try:
        arg = tuple(np.asarray(a, np.float32) for a in ssm) if f32 else ssm
        yy = y.astype(np.float32) if f32 else y
        tol = 3e-3 if f32 else 1e-7
        sms, sPs, fms, fPs, ll = B.pkfs(arg, yy, return_filtered=True, return_loglikelihood=True)
        assert relerr(fms, of) < tol and relerr(fPs, oP) < tol, tag
        assert relerr(sms, os_) < tol and relerr(sPs, osP) < tol, tag
        assert abs(float(ll) - oll) <= tol * abs(oll) + 1e-8, (tag, float(ll), oll)
        fms, fPs, ll = B.pkf(arg, yy, return_loglikelihood=True)
        assert relerr(fms, of) < tol and relerr(fPs, oP) < tol and abs(float(ll) - oll) <= tol * abs(oll) + 1e-8, tag
    finally:
        ctx.set_chunk(0)
    return tag


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    only = set(sys.argv[3].split(",")) if len(sys.argv) > 3 else None       # e.g. "two_rows,large_d,segments"
    bad = 0
    for seed in range(first, first + count):
        for name, fn in (("rc", T.test_random_models_sizes_and_chains), ("rc_fp32", T.test_random_models_fp32_row_cooperative),
                         ("quad_fp32", T.random_quad_case), ("large_d", large_d_case), ("two_rows", two_rows_case),
                         ("segments", T.random_segments_case), ("adjoint", T.test_random_adjoint_statistics),
                         ("fused_adjoint", T.test_random_fused_path_adjoint_statistics)):
            if only and name not in only:
                continue
            try:
                fn(seed)
            except Exception:
                bad += 1
                print(f"FAIL {name} seed {seed}", flush=True)
                traceback.print_exc()
        if seed % 5 == 0:
            print(f"seed {seed} done, failures so far {bad}", flush=True)
    print(f"campaign finished: seeds {first}..{first + count - 1}, failures {bad}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
