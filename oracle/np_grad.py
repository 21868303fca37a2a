"""
CPU ORACLE -- TEST INFRASTRUCTURE ONLY (see np_oracle.py's header: only tests/, smoke() and bench.py's cpu_baseline leg
may import this; nothing under parallel-gps_amd/ does).

Gradient of the state-space log-likelihood (pssgp/kalman/parallel.py:135-151 = sequential.py:11-47) with respect to
the MODEL (F, Pinf, H, R) of an LTI kernel, by a plain sequential reverse sweep over the Kalman filter -- the checker of
the device's adjoint pass.  The reference gets these numbers from TensorFlow autodiff
(tests/test_gp_vs_kfs.py:53-78); here they are restated as explicit recursions and pinned by finite differences of
np_oracle's own likelihood and by the dense GP's gradient (tests/test_grad_host.py).

Model of a step (m, P filtered at k-1; P_{-1} = Pinf, m_{-1} = 0; Q_k = Pinf - F_k Pinf F_k^T):
    mp = F_k m            Pp = Pinf + F_k (P - Pinf) F_k^T        F_k = expm(dt_k F)
    s = H Pp H^T + R      r = y - H mp        ll += -1/2 log(2 pi s) - 1/2 r^2 / s
    K = Pp H^T / s        m' = mp + K r       P' = Pp - K s K^T               (NaN y: m' = mp, P' = Pp, no term)
Reverse sweep with a = d ll / d m', B = d ll / d P' (symmetric):
    sbar = -(a.K) r / s + K^T B K - 1/(2 s) + r^2 / (2 s^2)        rbar = a.K - r / s
    ubar = a r / s - 2 B K + sbar H^T         mpbar = a - rbar H^T        Ppbar = B + sym(ubar H)
    a <- F_k^T mpbar                          B <- F_k^T Ppbar F_k
The statistics returned: ll and
    Abar = sum_k dt_k [mpbar mp^T + 2 Ppbar (Pp - Pinf)]   ( = sum_k dt_k Fbar_k F_k^T: contracts with any dF that
                                                              commutes with F, for which d F_k = dt_k dF F_k)
    Ubar = sum_k ubar_k        (d ll / d Pinf = sym(Ubar H): the P-adjoints telescope)
    Hbar = sum_k sbar u + Pp ubar - rbar mp         (u = Pp H^T)
    Rbar = sum_k sbar
so that  d ll / d theta = <Abar, dF> + Ubar^T dPinf H^T + Hbar . dH + Rbar dR.
"""
import math

import numpy as np
import scipy.linalg as sla

LOG2PI = math.log(2.0 * math.pi)


def ll_grad_stats(F, Pinf, H, R, ts, ys, t0=0.0):
    F = np.asarray(F, np.float64)
    Pinf = np.asarray(Pinf, np.float64)
    h = np.asarray(H, np.float64).reshape(-1)
    R = float(R)
    ts = np.asarray(ts, np.float64).reshape(-1)
    ys = np.asarray(ys, np.float64).reshape(-1)
    n, d = ts.size, F.shape[0]
    dts = np.diff(np.concatenate([[t0], ts]))
    Fk = np.stack([sla.expm(dt * F) for dt in dts])
    m, P = np.zeros(d), Pinf.copy()
    ll = 0.0
    keep = []
    for k in range(n):
        mp = Fk[k] @ m
        Pp = Pinf + Fk[k] @ (P - Pinf) @ Fk[k].T
        Pp = 0.5 * (Pp + Pp.T)
        u = Pp @ h
        s = float(h @ u) + R
        obs = not math.isnan(ys[k])
        r = ys[k] - float(h @ mp) if obs else 0.0
        if obs:
            ll += -0.5 * (LOG2PI + math.log(s)) - 0.5 * r * r / s
            K = u / s
            m, P = mp + K * r, Pp - np.outer(K, K) * s
        else:
            K = np.zeros(d)
            m, P = mp, Pp
        keep.append((mp, Pp, u, s, r, K, obs))
    a, B = np.zeros(d), np.zeros((d, d))
    Abar, Ubar, Hbar, Rbar = np.zeros((d, d)), np.zeros(d), np.zeros(d), 0.0
    for k in range(n - 1, -1, -1):
        mp, Pp, u, s, r, K, obs = keep[k]
        if obs:
            aK = float(a @ K)
            BK = B @ K
            sbar = -aK * r / s + float(K @ BK) - 0.5 / s + 0.5 * r * r / (s * s)
            rbar = aK - r / s
            ubar = a * (r / s) - 2.0 * BK + sbar * h
        else:
            sbar, rbar, ubar = 0.0, 0.0, np.zeros(d)
        mpbar = a - rbar * h
        Ppbar = B + 0.5 * (np.outer(ubar, h) + np.outer(h, ubar))
        Abar += dts[k] * (np.outer(mpbar, mp) + 2.0 * Ppbar @ (Pp - Pinf))
        Ubar += ubar
        Hbar += sbar * u + Pp @ ubar - rbar * mp
        Rbar += sbar
        a = Fk[k].T @ mpbar
        B = Fk[k].T @ Ppbar @ Fk[k]
        B = 0.5 * (B + B.T)
    return ll, Abar, Ubar, Hbar, Rbar


def contract(stats, H, grads, dR=None):
    """Gradient with respect to the kernel's parameters (grads: [(dF, dPinf, dH)], pssgp.kernels.sde_grads) followed by the
    observation noise (d ll / d R)."""
    _, Abar, Ubar, Hbar, Rbar = stats
    h = np.asarray(H, np.float64).reshape(-1)
    g = [float(np.sum(Abar * dF) + Ubar @ np.asarray(dP) @ h + Hbar @ np.asarray(dH).reshape(-1)) for dF, dP, dH in grads]
    return np.array(g + [Rbar])


def ll_only(F, Pinf, H, R, ts, ys, t0=0.0):
    """The same filter without the reverse sweep (for difference quotients in the tests)."""
    F = np.asarray(F, np.float64)
    Pinf = np.asarray(Pinf, np.float64)
    h = np.asarray(H, np.float64).reshape(-1)
    ts = np.asarray(ts, np.float64).reshape(-1)
    ys = np.asarray(ys, np.float64).reshape(-1)
    d = F.shape[0]
    dts = np.diff(np.concatenate([[t0], ts]))
    m, P = np.zeros(d), Pinf.copy()
    ll = 0.0
    for k in range(ts.size):
        Fk = sla.expm(dts[k] * F)
        mp = Fk @ m
        Pp = Pinf + Fk @ (P - Pinf) @ Fk.T
        u = Pp @ h
        s = float(h @ u) + float(R)
        if math.isnan(ys[k]):
            m, P = mp, Pp
            continue
        r = ys[k] - float(h @ mp)
        ll += -0.5 * (LOG2PI + math.log(s)) - 0.5 * r * r / s
        K = u / s
        m, P = mp + K * r, Pp - np.outer(K, K) * s
    return ll
