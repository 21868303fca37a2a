"""
CPU ORACLE -- TEST INFRASTRUCTURE ONLY.

A plain-numpy (fp64) restatement of the one hot path of EEA-sensors/parallel-gps
(`pssgp`): LTI discretisation, sequential Kalman filter / RTS smoother, the
parallel (associative-scan) filter / smoother, and the dense-GP ground truth the
reference's own tests compare against.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this module -- and only as the checker.  Nothing under `parallel-gps_amd/`
imports it; the product path fails loudly if the HIP library is missing.

Parity pin.  The reference (TensorFlow 2.6 / TFP 0.13 /
GPflow 2.2.1) cannot be imported in the build container and holds no stored
vectors for this path.  What its tests *do* pin is
  (i)   state-space log-likelihood == dense GPR log-marginal-likelihood
        (tests/test_gp_vs_kfs.py:45-73, 1e-6 for Matern / sum / product),
  (ii)  predict_f mean/var == dense GPR posterior (tests/test_gp_vs_kfs.py:80-99),
  (iii) known-answer SDE constants (tests/test_rbf.py:27-39,
        tests/test_periodic.py:32-50).
  (iv)  the parameters its own three models learn in notebooks/PSSGP101.ipynb cell 13
        on notebooks/data/regression_1D.csv (7.96569 / 0.212416 / 0.00575949) -- the
        reference-held numeric pin, reproduced in tests/test_reference_pin.py.
`tests/test_oracle.py` re-runs (i)-(iii) against this file: the dense GP below is
independent of every state-space routine, so agreement of `kf/ks` and `pkf/pks`
with it to 1e-9 is the pin.  Third-party arithmetic restated from its published
algorithm: `tfp.math.scan_associative` (tensorflow_probability==0.13.0,
requirements.txt:98) -> `scan_associative` below (odd/even recursive doubling);
`tf.linalg.expm` -> `scipy.linalg.expm` (both Higham Pade scaling-and-squaring).

Every function cites the reference file:line it follows (paths relative to
/root/reference).
"""
import math

import numpy as np
import scipy.linalg as sla

LOG2PI = math.log(2.0 * math.pi)


# --------------------------------------------------------------------------------------
# O1: dense GP ground truth  (GPflow 2.2.1 kernel formulas; GPR log marginal likelihood)
# --------------------------------------------------------------------------------------
def dense_K(spec, x1, x2):
    """Dense covariance for a kernel *spec* (nested tuples).

    spec := ("matern12"|"matern32"|"matern52"|"rbf", variance, lengthscale)
          | ("periodic", variance, lengthscale, period)      # Periodic(SquaredExponential)
          | ("sum", [spec, ...]) | ("prod", [spec, ...])
    GPflow 2.2.1 stationaries.py formulas; these are what the reference's tests use as
    ground truth through gpflow.models.GPR (tests/test_gp_vs_kfs.py:47-50).
    """
    kind = spec[0]
    x1 = np.asarray(x1, dtype=np.float64).reshape(-1)
    x2 = np.asarray(x2, dtype=np.float64).reshape(-1)
    if kind == "sum":
        return sum(dense_K(s, x1, x2) for s in spec[1])
    if kind == "prod":
        out = np.ones((x1.size, x2.size))
        for s in spec[1]:
            out = out * dense_K(s, x1, x2)
        return out
    r = np.abs(x1[:, None] - x2[None, :])
    if kind == "periodic":
        _, var, ell, period = spec
        s = np.sin(np.pi * r / period) / ell
        return var * np.exp(-0.5 * s * s)
    _, var, ell = spec
    r = r / ell
    if kind == "matern12":
        return var * np.exp(-r)
    if kind == "matern32":
        s3 = math.sqrt(3.0)
        return var * (1.0 + s3 * r) * np.exp(-s3 * r)
    if kind == "matern52":
        s5 = math.sqrt(5.0)
        return var * (1.0 + s5 * r + 5.0 / 3.0 * r * r) * np.exp(-s5 * r)
    if kind == "rbf":
        return var * np.exp(-0.5 * r * r)
    raise ValueError(kind)


def dense_gp(spec, t, y, noise_variance, t_query=None):
    """GPR log marginal likelihood and posterior (zero mean function).

    ll = log N(y | 0, K + r I); mean = K*^T (K + r I)^-1 y; var = k** - diag(K*^T (K+rI)^-1 K*).
    Ground truth in tests/test_gp_vs_kfs.py:47-51,82-86.
    """
    t = np.asarray(t, dtype=np.float64).reshape(-1)
    y = np.asarray(y, dtype=np.float64).reshape(-1)
    n = t.size
    K = dense_K(spec, t, t) + noise_variance * np.eye(n)
    L = np.linalg.cholesky(K)
    alpha = sla.solve_triangular(L, y, lower=True)
    ll = -0.5 * float(alpha @ alpha) - float(np.sum(np.log(np.diag(L)))) - 0.5 * n * LOG2PI
    if t_query is None:
        return ll
    tq = np.asarray(t_query, dtype=np.float64).reshape(-1)
    Ks = dense_K(spec, t, tq)
    A = sla.solve_triangular(L, Ks, lower=True)
    mean = A.T @ alpha
    var = np.diag(dense_K(spec, tq, tq)) - np.sum(A * A, axis=0)
    return ll, mean, var


# --------------------------------------------------------------------------------------
# LTI discretisation  (pssgp/kernels/base.py:29-47)
# --------------------------------------------------------------------------------------
def get_ssm(sde, ts, R, t0=0.0):
    """(P0, F, L, H, Q) continuous model -> LGSSM (P0, Fs, Qs, H, R).

    Follows pssgp/kernels/base.py:29-47 literally: dts from [t0; ts] (34-35),
    Fs = expm(dt F) (36), Phi = [[F, L Q L^T],[0, -F^T]] (39-42),
    AB = expm(dt Phi) @ [0; I] (44-45), Qs = AB[:, :n] @ Fs^T (46).
    """
    P0, F, L, H, Q = (np.asarray(a, dtype=np.float64) for a in sde)
    n = F.shape[0]
    ts = np.asarray(ts, dtype=np.float64).reshape(-1)
    dts = np.diff(np.concatenate([[float(t0)], ts]))
    LQL = L @ np.atleast_2d(Q) @ L.T
    Phi = np.block([[F, LQL], [np.zeros((n, n)), -F.T]])
    Fs = np.empty((ts.size, n, n))
    Qs = np.empty((ts.size, n, n))
    sel = np.concatenate([np.zeros((n, n)), np.eye(n)], axis=0)
    cache = {}
    for k, dt in enumerate(dts):
        hit = cache.get(dt)
        if hit is None:
            Fk = sla.expm(dt * F)
            AB = sla.expm(dt * Phi) @ sel
            hit = (Fk, AB[:n, :] @ Fk.T)
            if len(cache) < 4096:
                cache[dt] = hit
        Fs[k], Qs[k] = hit
    return P0, Fs, Qs, H.reshape(1, n), np.asarray(R, dtype=np.float64).reshape(1, 1)


# --------------------------------------------------------------------------------------
# O2: sequential Kalman filter and RTS smoother  (pssgp/kalman/sequential.py)
# --------------------------------------------------------------------------------------
def kf(lgssm, observations, return_loglikelihood=False, return_predicted=False):
    """pssgp/kalman/sequential.py:11-47.  m0 = 0 (14); predict + symmetrise (19-21);
    scalar-innovation update unless y is NaN (23-38); symmetrise (39)."""
    P0, Fs, Qs, H, R = lgssm
    ys = np.asarray(observations, dtype=np.float64).reshape(-1)
    N, d = Fs.shape[0], Fs.shape[1]
    h = H.reshape(d)
    r = float(np.asarray(R).reshape(()))
    m = np.zeros(d)
    P = np.array(P0, dtype=np.float64)
    fms, fPs = np.empty((N, d)), np.empty((N, d, d))
    mps, Pps = np.empty((N, d)), np.empty((N, d, d))
    ell = 0.0
    for k in range(N):
        F, Q, y = Fs[k], Qs[k], ys[k]
        mp = F @ m
        Pp = F @ P @ F.T + Q
        Pp = 0.5 * (Pp + Pp.T)
        if not np.isnan(y):
            S = float(h @ Pp @ h) + r
            yp = float(h @ mp)
            ell += -0.5 * (LOG2PI + math.log(S) + (y - yp) ** 2 / S)
            Kt = (h @ Pp) / S                       # cholesky_solve(chol(S), H P)  (29)
            m = mp + Kt * (y - yp)
            P = Pp - np.outer(Kt, Kt) * S
        else:
            m, P = mp, Pp
        P = 0.5 * (P + P.T)
        fms[k], fPs[k], mps[k], Pps[k] = m, P, mp, Pp
    out = (fms, fPs)
    if return_loglikelihood:
        out += (ell,)
    if return_predicted:
        out += (mps, Pps)
    return out


def ks(lgssm, ms, Ps, mps, Pps):
    """pssgp/kalman/sequential.py:50-68 (reverse scan; Ct = Pp^-1 F P by Cholesky 57-58)."""
    _, Fs, Qs, *_ = lgssm
    N, d = ms.shape
    sms, sPs = np.empty_like(ms), np.empty_like(Ps)
    sms[-1], sPs[-1] = ms[-1], Ps[-1]
    for k in range(N - 2, -1, -1):
        F = Fs[k + 1]
        Ct = sla.cho_solve(sla.cho_factor(Pps[k + 1], lower=True), F @ Ps[k])
        sm = ms[k] + Ct.T @ (sms[k + 1] - mps[k + 1])
        sP = Ps[k] + Ct.T @ (sPs[k + 1] - Pps[k + 1]) @ Ct
        sms[k], sPs[k] = sm, 0.5 * (sP + sP.T)
    return sms, sPs


def kfs(lgssm, observations):
    """pssgp/kalman/sequential.py:71-73."""
    fms, fPs, mps, Pps = kf(lgssm, observations, return_predicted=True)
    return ks(lgssm, fms, fPs, mps, Pps)


# --------------------------------------------------------------------------------------
# O3: parallel filter / smoother  (pssgp/kalman/parallel.py), batched numpy
# --------------------------------------------------------------------------------------
def _T(X):
    return np.swapaxes(X, -1, -2)


def _mv(A, x):
    return np.einsum("...ij,...j->...i", A, x)


def first_filtering_element(m0, P0, F, Q, H, R, y):
    """pssgp/kalman/parallel.py:13-43."""
    d = F.shape[0]
    if np.isnan(y):
        return np.zeros((d, d)), m0.copy(), P0.copy(), np.zeros((d, d)), np.zeros(d)
    h = H.reshape(d)
    r = float(np.asarray(R).reshape(()))
    S1 = float(h @ P0 @ h) + r
    K1t = (h @ P0) / S1                                                        # (26)
    A = np.zeros((d, d))
    b = m0 + K1t * (y - float(h @ m0))                                         # (29)
    C = P0 - np.outer(K1t, K1t) * S1                                           # (30)
    S = float(h @ Q @ h) + r                                                   # (32)
    HF = h @ F
    eta = HF * (y / S)                                                         # (35-37)
    J = np.outer(HF, HF) / S                                                   # (38)
    return A, b, C, J, eta


def generic_filtering_elements(Fs, Qs, H, R, ys):
    """Batched pssgp/kalman/parallel.py:46-72 and the NaN select of 86-95."""
    N, d = Fs.shape[0], Fs.shape[1]
    h = H.reshape(d)
    r = float(np.asarray(R).reshape(()))
    nan = np.isnan(ys)
    y0 = np.where(nan, 0.0, ys)
    S = np.einsum("i,nij,j->n", h, Qs, h) + r                                   # (57)
    HQ = np.einsum("i,nij->nj", h, Qs)
    Kt = HQ / S[:, None]                                                        # (60)
    KH = Kt[:, :, None] * h[None, None, :]                                      # Kt^T H
    A = Fs - KH @ Fs                                                            # (61)
    b = Kt * y0[:, None]                                                        # (62)
    C = Qs - KH @ Qs                                                            # (63)
    HF = np.einsum("i,nij->nj", h, Fs)                                          # (65)
    eta = HF * (y0 / S)[:, None]                                                # (66-68)
    J = HF[:, :, None] * HF[:, None, :] / S[:, None, None]                      # (70)
    m3, m2 = nan[:, None, None], nan[:, None]
    A = np.where(m3, Fs, A)                                                     # (46-53)
    b = np.where(m2, 0.0, b)
    C = np.where(m3, Qs, C)
    J = np.where(m3, 0.0, J)
    eta = np.where(m2, 0.0, eta)
    return A, b, C, J, eta


def make_associative_filtering_elements(m0, P0, Fs, Qs, H, R, ys):
    """pssgp/kalman/parallel.py:83-97 (row 0 overwritten by the first element)."""
    elems = generic_filtering_elements(Fs, Qs, H, R, ys)
    first = first_filtering_element(m0, P0, Fs[0], Qs[0], H, R, ys[0])
    for arr, f in zip(elems, first):
        arr[0] = f
    return elems


def filtering_operator(e1, e2):
    """pssgp/kalman/parallel.py:100-118 (two solves, symmetrisation of C and J)."""
    A1, b1, C1, J1, eta1 = e1
    A2, b2, C2, J2, eta2 = e2
    d = A1.shape[-1]
    I = np.eye(d)
    temp = np.linalg.solve(_T(I + C1 @ J2), _T(A2))          # solve(M, A2^T, adjoint=True)
    tT = _T(temp)
    A = tT @ A1
    b = _mv(tT, b1 + _mv(C1, eta2)) + b2
    C = tT @ (C1 @ _T(A2)) + C2
    temp = np.linalg.solve(_T(I + J2 @ C1), A1)
    tT = _T(temp)
    eta = _mv(tT, eta2 - _mv(J2, b1)) + eta1
    J = tT @ (J2 @ A1) + J1
    C = 0.5 * (C + _T(C))
    J = 0.5 * (J + _T(J))
    return A, b, C, J, eta


def smoothing_operator(e1, e2):
    """pssgp/kalman/parallel.py:176-184."""
    E1, g1, L1 = e1
    E2, g2, L2 = e2
    return E2 @ E1, _mv(E2, g1) + g2, E2 @ L1 @ _T(E2) + L2


def scan_associative(op, elems, bracketing="tree"):
    """Inclusive prefix combine under `op`.

    bracketing="tree": the odd/even recursive doubling of tfp.math.scan_associative
    (tensorflow_probability==0.13.0; call sites pssgp/kalman/parallel.py:131-133,193-195):
    combine adjacent pairs, recurse on the n/2 results (these are the prefixes at odd
    positions), then one more combine gives the even positions.
    bracketing="sequential": plain left fold (same value up to round-off by associativity).
    """
    n = elems[0].shape[0]
    if bracketing == "sequential":
        out = [np.empty_like(e) for e in elems]
        acc = tuple(e[0:1] for e in elems)
        for o, a in zip(out, acc):
            o[0] = a[0]
        for k in range(1, n):
            acc = op(acc, tuple(e[k:k + 1] for e in elems))
            for o, a in zip(out, acc):
                o[k] = a[0]
        return tuple(out)
    if n < 2:
        return tuple(e.copy() for e in elems)
    reduced = op(tuple(e[0:-1:2] for e in elems), tuple(e[1::2] for e in elems))
    odd = scan_associative(op, reduced, "tree")
    if n % 2 == 0:
        even = op(tuple(o[:-1] for o in odd), tuple(e[2::2] for e in elems))
    else:
        even = op(odd, tuple(e[2::2] for e in elems))
    out = []
    for e, ev, od in zip(elems, even, odd):
        res = np.empty_like(e)
        res[0] = e[0]
        res[2::2] = ev
        res[1::2] = od
        out.append(res)
    return tuple(out)


def pkf(lgssm, observations, return_loglikelihood=False, bracketing="tree"):
    """pssgp/kalman/parallel.py:121-152."""
    P0, Fs, Qs, H, R = lgssm
    ys = np.asarray(observations, dtype=np.float64).reshape(-1)
    d = P0.shape[0]
    h = H.reshape(d)
    r = float(np.asarray(R).reshape(()))
    m0 = np.zeros(d)
    elems = make_associative_filtering_elements(m0, np.asarray(P0, float), Fs, Qs, H, R, ys)
    final = scan_associative(filtering_operator, elems, bracketing)
    fms, fPs = final[1], final[2]
    if not return_loglikelihood:
        return fms, fPs
    pm = np.concatenate([m0[None], fms[:-1]], axis=0)                           # (136)
    pP = np.concatenate([np.asarray(P0, float)[None], fPs[:-1]], axis=0)        # (137)
    mp = _mv(Fs, pm)                                                            # (138)
    Pp = Fs @ pP @ _T(Fs) + Qs                                                  # (139)
    mu = mp @ h                                                                 # (140)
    s2 = np.einsum("i,nij,j->n", h, Pp, h) + r                                  # (141)
    lp = -0.5 * (LOG2PI + np.log(s2) + (ys - mu) ** 2 / s2)                     # (143-145)
    lp = np.where(np.isnan(lp), 0.0, lp)                                        # (147-149)
    return fms, fPs, float(np.sum(lp))


def make_associative_smoothing_elements(Fs, Qs, fms, fPs):
    """pssgp/kalman/parallel.py:155-173."""
    F, Q, m, P = Fs[1:], Qs[1:], fms[:-1], fPs[:-1]
    Pp = F @ P @ _T(F) + Q                                                      # (160)
    # E = (Pp^-1 F P)^T via Cholesky (161-162)
    E = np.empty_like(P)
    for k in range(P.shape[0]):
        E[k] = sla.cho_solve(sla.cho_factor(Pp[k], lower=True), F[k] @ P[k]).T
    g = m - _mv(E @ F, m)                                                       # (163)
    L = P - E @ Pp @ _T(E)                                                      # (164)
    L = 0.5 * (L + _T(L))                                                       # (165)
    d = fms.shape[1]
    E = np.concatenate([E, np.zeros((1, d, d))], axis=0)                        # (155-156)
    g = np.concatenate([g, fms[-1:]], axis=0)
    L = np.concatenate([L, fPs[-1:]], axis=0)
    return E, g, L


def pks(lgssm, ms, Ps, bracketing="tree"):
    """pssgp/kalman/parallel.py:187-196 (scan over time-reversed elements)."""
    _, Fs, Qs, *_ = lgssm
    elems = make_associative_smoothing_elements(Fs, Qs, ms, Ps)
    rev = tuple(e[::-1].copy() for e in elems)
    final = scan_associative(smoothing_operator, rev, bracketing)
    return final[1][::-1].copy(), final[2][::-1].copy()


def pkfs(lgssm, observations, bracketing="tree"):
    """pssgp/kalman/parallel.py:199-201."""
    fms, fPs = pkf(lgssm, observations, False, bracketing)
    return pks(lgssm, fms, fPs, bracketing)


# --------------------------------------------------------------------------------------
# Model-level restatement  (pssgp/model.py)
# --------------------------------------------------------------------------------------
def merge_sorted(a, b, *pairs):
    """pssgp/model.py:15-55.  `a` and `b` sorted 1-D; the shorter array is scattered into
    the longer one at arange + searchsorted(longer, shorter) (side='left', model.py:43)."""
    a = np.asarray(a)
    b = np.asarray(b)
    if a.shape[0] < b.shape[0]:
        a, b = b, a
        pairs = tuple((j, i) for i, j in pairs)
    b_idx = np.arange(b.shape[0]) + np.searchsorted(a, b, side="left")
    n = a.shape[0] + b.shape[0]
    is_a = np.ones(n, dtype=bool)
    is_a[b_idx] = False

    def inner(u, v):
        u, v = np.asarray(u), np.asarray(v)
        c = np.empty((n,) + u.shape[1:], dtype=np.result_type(u, v))
        c[b_idx] = v
        c[is_a] = u
        return c

    return (inner(a, b),) + tuple(inner(i, j) for i, j in pairs)


def ssgp_log_likelihood(sde, t, y, noise_variance, parallel=True):
    """pssgp/model.py:113-117."""
    ssm = get_ssm(sde, t, noise_variance)
    if parallel:
        return pkf(ssm, y, True)[2]
    return kf(ssm, y, True)[2]


def ssgp_predict_f(sde, t, y, noise_variance, t_query, parallel=True):
    """pssgp/model.py:92-111: merge, NaN-mark queries, smooth, mask, project through H."""
    t = np.asarray(t, float).reshape(-1)
    y = np.asarray(y, float).reshape(-1)
    tq = np.asarray(t_query, float).reshape(-1)
    all_t, all_y, flags = merge_sorted(t, tq, (y, np.full(tq.shape, np.nan)),
                                       (np.zeros(t.shape, bool), np.ones(tq.shape, bool)))
    ssm = get_ssm(sde, all_t, noise_variance)
    sms, sPs = pkfs(ssm, all_y) if parallel else kfs(ssm, all_y)
    h = ssm[3].reshape(-1)
    sm, sP = sms[flags], sPs[flags]
    return sm @ h, np.einsum("i,nij,j->n", h, sP, h)
