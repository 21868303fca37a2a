#!/usr/bin/env python3
"""Structured inputs for the round-3 'ghost' (tools/d32_ghost.py reproduces it with a library built from commit edfddd1):
models whose chain totals are known in closed form, so that a wrong entry of a filtered covariance names the entry of the
chain total (A, C, J, b, eta) that was corrupted, and the part of the Kalman step that produced it.

    identity   F = I, Q = 0, every observation missing:   fm = 0, fP = P0 exactly at every step (A = I, C = J = 0)
    diagonal   F = diag(f_i), Q = 0, missing:             fP = diag-scaled P0 (the F A and F C F^T products only)
    dense      F random, Q = 0, missing:                  fP = (prod F) P0 (prod F)^T
    noise      F = I, Q = q I, missing:                   fP = P0 + k q I   (the copy of Q into C)
    observed   F = I, Q = 0, observations through H = e_0 (the rank-one updates and J, eta)
PGPS_LIB=... python tools/d32_probe.py [d] [n] [chunk]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "parallel-gps_amd"))
from oracle import np_oracle as O
from pssgp import _backend as B


def report(name, got, want, n, d):
    got = np.asarray(got, float).reshape(n, -1); want = np.asarray(want, float).reshape(n, -1)
    scale = np.abs(want).max() + 1e-300
    bad = np.abs(got - want) > 1e-9 * scale
    steps = np.flatnonzero(bad.any(axis=1))
    if steps.size == 0:
        print(f"    {name}: exact to 1e-9 at all {n} steps", flush=True)
        return
    k = steps[0]
    ent = np.flatnonzero(bad[k])
    if got.shape[1] == d * d:
        rows = sorted(set((ent // d).tolist())); cols = sorted(set((ent % d).tolist()))
        print(f"    {name}: {steps.size} steps wrong, first {k}, last {steps[-1]}; at step {k}: {ent.size} entries, rows {rows} cols {cols}; "
              f"worst |err| {np.abs(got[k] - want[k]).max():.3e} (scale {scale:.2e})", flush=True)
        worst = ent[np.argsort(-np.abs(got[k] - want[k])[ent])[:6]]
        print("        " + "  ".join(f"({e // d},{e % d}): got {got[k, e]:+.4e} want {want[k, e]:+.4e}" for e in worst), flush=True)
        if name == "fP":            # the map of wrong entries: row = the lane that owns it, column = its register
            m = bad[k].reshape(d, d); z = (got[k].reshape(d, d) == 0.0) & m
            for r in range(d):
                print("        " + "".join("0" if z[r, c] else ("x" if m[r, c] else ".") for c in range(d)), flush=True)
    else:
        print(f"    {name}: {steps.size} steps wrong, first {k}, last {steps[-1]}; at step {k}: entries {ent.tolist()}; "
              f"worst |err| {np.abs(got[k] - want[k]).max():.3e} (scale {scale:.2e})", flush=True)


def main():
    d = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 96
    chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    print("library:", B._LIB_PATH, "d", d, "n", n, "chunk", chunk, flush=True)
    rng = np.random.default_rng(5)
    L = rng.standard_normal((d, d)) * 0.3
    P0 = L @ L.T + np.diag(np.arange(1, d + 1, dtype=float))
    eye = np.eye(d)
    e0 = np.zeros((1, d)); e0[0, 0] = 1.0
    hr = rng.standard_normal((1, d))
    nan = np.full(n, np.nan)
    yobs = rng.standard_normal(n)
    fdiag = np.stack([np.diag(1.0 - 0.01 * (np.arange(d) + 1) / d * (1 + (k % 3))) for k in range(n)])
    M = rng.standard_normal((d, d)) * 0.1
    fdense = np.stack([eye + M * (0.5 + 0.1 * (k % 5)) for k in range(n)])
    zero = np.zeros((n, d, d))
    fdiag[0] = eye; fdense[0] = eye     # the first element takes the prior as it is (reference parallel.py:23-29): keep the
                                        # sequential oracle on the same footing
    cases = [
        ("identity", np.broadcast_to(eye, (n, d, d)).copy(), zero, e0, nan),
        ("diagonal", fdiag, zero, e0, nan),
        ("dense", fdense, zero, e0, nan),
        ("noise", np.broadcast_to(eye, (n, d, d)).copy(), np.broadcast_to(0.25 * eye, (n, d, d)).copy(), e0, nan),
        ("observed e0", np.broadcast_to(eye, (n, d, d)).copy(), zero, e0, yobs),
        ("observed H random", np.broadcast_to(eye, (n, d, d)).copy(), zero, hr, yobs),
        ("dense observed", fdense, np.broadcast_to(0.05 * eye, (n, d, d)).copy(), hr, yobs),
    ]
    B.get_context().set_chunk(chunk)
    for name, Fs, Qs, H, y in cases:
        ssm = (P0, Fs, Qs, H, np.array([[0.3]]))
        print(f"{name}:", flush=True)
        try:
            sms, sPs, fms, fPs, ll = B.pkfs(ssm, y, return_filtered=True, return_loglikelihood=True)
        except Exception as e:              # noqa: BLE001
            print("    ", e, flush=True)
            continue
        of, oP, oll = O.kf(ssm, y, True)
        report("fm", fms, of, n, d)
        report("fP", fPs, oP, n, d)
        try:
            os_, osP = O.kfs(ssm, y)
            report("sm", sms, os_, n, d)
            report("sP", sPs, osP, n, d)
        except np.linalg.LinAlgError as e:
            print("    smoother oracle:", e, flush=True)
        print(f"    ll {float(ll):.12g} oracle {oll:.12g}", flush=True)
    B.get_context().set_chunk(0)


def fit():
    """One generic step after the first element (n = 4, chunk = 2): the chain total's C is F1 P0 F1^T + Q1; which terms of
    which product are missing?  Candidates are built on the host and compared with what the device returns at step 2."""
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    rng = np.random.default_rng(11)
    L = rng.standard_normal((d, d)) * 0.3
    P0 = L @ L.T + np.diag(np.arange(1, d + 1, dtype=float))
    n = 4
    Fs = np.stack([np.eye(d)] + [np.eye(d) * 0.9 + rng.standard_normal((d, d)) * 0.2 for _ in range(n - 1)])
    Qs = np.stack([np.zeros((d, d))] + [0.1 * np.eye(d)] * (n - 1))
    H = np.zeros((1, d)); H[0, 0] = 1.0
    y = np.full(n, np.nan)
    B.get_context().set_chunk(2)
    ssm = (P0, Fs, Qs, H, np.array([[0.3]]))
    sms, sPs, fms, fPs, ll = B.pkfs(ssm, y, return_filtered=True, return_loglikelihood=True)
    B.get_context().set_chunk(0)
    got1 = fPs[1]                               # inside chain 0 (apply kernel only)
    F1, F2 = Fs[1], Fs[2]
    true1 = F1 @ P0 @ F1.T + Qs[1]
    print("step 1 (inside the first chain): max |err|", np.abs(got1 - true1).max(), flush=True)
    got2 = fPs[2]
    # step 2 = F2 C F2^T + Q2 with C the (possibly wrong) total of chain 0: recover C
    F2i = np.linalg.inv(F2)
    C = F2i @ (got2 - Qs[2]) @ F2i.T
    print("recovered chain total C against F1 P0 F1^T + Q1: max |err|", np.abs(C - true1).max(), flush=True)
    out = os.path.join(ROOT, "gpurun_out", "probe_fit.npz")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    np.savez(out, C=C, F1=F1, F2=F2, P0=P0, Q1=Qs[1], got2=got2, fPs=fPs, sPs=sPs)
    sym = lambda x: 0.5 * (x + x.T)
    Bset = [5, 6, 7]
    keep = np.array([k not in Bset for k in range(d)])
    cands = {}
    for name, S in (("k in {5,6,7}", Bset), ("k in {5}", [5]), ("k in {6}", [6]), ("k in {7}", [7])):
        kp = np.array([k not in S for k in range(d)])
        Fk = F1 * kp[None, :]
        cands[f"first product without terms {name}"] = sym((Fk @ P0) @ F1.T + Qs[1])
        cands[f"second product without terms {name}"] = sym((F1 @ P0) @ Fk.T + Qs[1])
        cands[f"both products without terms {name}"] = sym((Fk @ P0) @ Fk.T + Qs[1])
        x = F1 @ P0 @ F1.T + Qs[1]
        xr = x * kp[:, None]
        cands[f"rows {name} of the result zero before symmetrising"] = sym(xr)
        xr = (F1 @ P0 @ F1.T) * kp[:, None] + Qs[1]
        cands[f"rows {name} of the product zero (Q kept)"] = sym(xr)
    for name, c in cands.items():
        print(f"    {np.abs(C - c).max():10.3e}   {name}", flush=True)
    m = np.abs(C - true1) > 1e-9 * np.abs(true1).max()
    for r in range(d):
        print("        " + "".join("x" if m[r, c] else "." for c in range(d)), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "fit":
        fit()
    else:
        main()
