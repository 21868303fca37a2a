"""CPU checks of the algebraic rearrangements the row-cooperative kernels (csrc/pgps_rc.hip.h) rely on, against the
oracle's restatement of the reference operators (pssgp/kalman/parallel.py:100-118, 159-184).  No GPU needed."""
import numpy as np
import pytest

from oracle import np_oracle as O


def _psd(rng, d, rank=None):
    a = rng.standard_normal((d, rank or d))
    return a @ a.T / d


def _filt_elem(rng, d, first=False):
    A = np.zeros((d, d)) if first else rng.standard_normal((d, d)) * 0.4
    return A, rng.standard_normal(d), _psd(rng, d), _psd(rng, d, rank=max(1, d // 2)), rng.standard_normal(d)


@pytest.mark.parametrize("d", [2, 5, 11, 16])
def test_single_elimination_form_of_the_filtering_operator(d):
    """rc_ks_filter: M = I + C1 J2, ONE solve Nm = M^-1 C1 (symmetric), then G = M^-1 A1 = A1 - Nm J2 A1,
    w = M^-1 (b1 + C1 eta2) = b1 + Nm (eta2 - J2 b1); outputs as in filt_combine."""
    rng = np.random.default_rng(d)
    for first in (False, True):
        A1, b1, C1, J1, e1 = _filt_elem(rng, d, first)
        A2, b2, C2, J2, e2 = _filt_elem(rng, d)
        M = np.eye(d) + C1 @ J2
        Nm = np.linalg.solve(M, C1)
        assert np.max(np.abs(Nm - Nm.T)) < 1e-12 * max(1.0, np.max(np.abs(Nm)))
        z = e2 - J2 @ b1
        W = J2 @ A1
        G = A1 - Nm @ W
        A = A2 @ G
        b = A2 @ (b1 + Nm @ z) + b2
        C = A2 @ Nm @ A2.T + C2
        eta = G.T @ z + e1
        J = G.T @ W + J1
        # the oracle's operator works on batched tuples (n, ...)
        e1t = tuple(x[None] for x in (A1, b1, C1, J1, e1))
        e2t = tuple(x[None] for x in (A2, b2, C2, J2, e2))
        Ao, bo, Co, Jo, eo = (x[0] for x in O.filtering_operator(e1t, e2t))
        for got, want in ((A, Ao), (b, bo), (0.5 * (C + C.T), Co), (0.5 * (J + J.T), Jo), (eta, eo)):
            assert np.max(np.abs(got - want)) < 1e-10 * max(1.0, np.max(np.abs(want)))


@pytest.mark.parametrize("d", [3, 11])
def test_stored_smoothing_elements_reproduce_rts(d):
    """rc_apply1 / rc_smooth1: the element (E, g, L) with E^T = Pp^-1 (F P), g = m - E mp, L = P - E (F P) and the
    backward recursion sm = E sm' + g, sP = E sP' E^T + L equal the RTS smoother; beyond the end of the series F = 0,
    Q = I gives the reference's last element (0, m, P)."""
    rng = np.random.default_rng(7 + d)
    n = 40
    F = rng.standard_normal((n, d, d)) * 0.5
    Q = np.stack([_psd(rng, d) + 0.1 * np.eye(d) for _ in range(n)])
    P0 = _psd(rng, d) + 0.5 * np.eye(d)
    H = rng.standard_normal((1, d))
    y = rng.standard_normal(n)
    y[::6] = np.nan
    ssm = (P0, F, Q, H, np.array([[0.3]]))
    fms, fPs, _ = O.kf(ssm, y, True)
    sms_o, sPs_o = O.kfs(ssm, y)
    E, g, L = np.zeros((n, d, d)), np.zeros((n, d)), np.zeros((n, d, d))
    for k in range(n):
        Fn, Qn = (F[k + 1], Q[k + 1]) if k + 1 < n else (np.zeros((d, d)), np.eye(d))
        FP = Fn @ fPs[k]
        Pp = FP @ Fn.T + Qn
        E[k] = np.linalg.solve(Pp, FP).T
        g[k] = fms[k] - E[k] @ (Fn @ fms[k])
        L[k] = fPs[k] - E[k] @ FP
    assert np.all(E[-1] == 0) and np.allclose(g[-1], fms[-1]) and np.allclose(L[-1], fPs[-1])
    sm, sP = np.zeros(d), np.zeros((d, d))
    for k in range(n - 1, -1, -1):
        sm = E[k] @ sm + g[k]
        sP = E[k] @ sP @ E[k].T + L[k]
        assert np.max(np.abs(sm - sms_o[k])) < 1e-10 and np.max(np.abs(sP - sPs_o[k])) < 1e-10


def test_first_element_as_an_extension_of_the_prior():
    """rc_reduce1's chain 0: (0, 0, P0, 0, 0) extended by (F = I, Q = 0, y0) is the reference's first element
    (parallel.py:13-43 with m0 = 0): A = 0, b = K y, C = P0 - K S K^T."""
    rng = np.random.default_rng(3)
    d = 6
    P0 = _psd(rng, d) + 0.2 * np.eye(d)
    H = rng.standard_normal((1, d))
    R, y0 = 0.4, 0.7
    A, b, C = np.zeros((d, d)), np.zeros(d), P0.copy()
    Fi, Qi = np.eye(d), np.zeros((d, d))
    Ap, bp, Cp = Fi @ A, Fi @ b, Fi @ C @ Fi.T + Qi
    u, v = Cp @ H[0], Ap.T @ H[0]
    S = H[0] @ u + R
    A1, b1, C1 = Ap - np.outer(u, v) / S, bp + u * (y0 - H[0] @ bp) / S, Cp - np.outer(u, u) / S
    Ao, bo, Co, _, _ = O.first_filtering_element(np.zeros(d), P0, np.eye(d), np.zeros((d, d)), H, np.array([[R]]), y0)
    assert np.max(np.abs(A1)) == 0.0 and np.allclose(b1, bo, atol=1e-13) and np.allclose(C1, Co, atol=1e-13)
