"""Host-side numerics for building a kernel's LTI SDE (once per call, O(d^6) at most).

balance_ss      -- diagonal-similarity balancing, arXiv 1401.5766
                   (reference: pssgp/kernels/math_utils.py:10-81; numba loop 10-29)
solve_lyap_vec  -- stationary covariance by the vectorised Lyapunov system
                   (reference: pssgp/kernels/math_utils.py:84-120)
"""
import numpy as np

try:                                    # optional: scipy's Bartels-Stewart solver for the larger state dimensions
    from scipy.linalg import solve_continuous_lyapunov as _lyap
except Exception:                       # pragma: no cover
    _lyap = None


def kron(A, B):
    """np.kron for the small matrices of an SDE, as one broadcast product (np.kron spends ~35 us of Python per call on
    them, nine calls per composite get_sde()); same entries, same values."""
    A = np.atleast_2d(np.asarray(A, np.float64))
    B = np.atleast_2d(np.asarray(B, np.float64))
    return (A[:, None, :, None] * B[None, :, None, :]).reshape(A.shape[0] * B.shape[0], A.shape[1] * B.shape[1])


def _native_balancing_diagonal(F, n_iter):
    """The same sweep in libpgps' host code (pgps_host_balance_f64; the reference compiles this loop with numba):
    None when the library is not built."""
    try:
        import ctypes
        from .. import _backend
        lib = _backend.load_library()
        fn = lib.pgps_host_balance_f64
    except Exception:
        return None
    Fc = np.ascontiguousarray(F, dtype=np.float64)
    scale = np.empty(Fc.shape[0], np.float64)
    with np.errstate(all="ignore"):
        code = fn(ctypes.c_int(Fc.shape[0]), Fc.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(int(n_iter)),
                  scale.ctypes.data_as(ctypes.c_void_p))
    return scale if code == 0 else None


def _balancing_diagonal(F, n_iter):
    """Sweep `n_iter` times over the states; each visit equalises the off-diagonal
    column and row 2-norms of the progressively rescaled matrix.  Returns the
    accumulated diagonal scaling d (math_utils.py:10-29: the norms are taken on the
    working copy, which is rescaled in place)."""
    native = _native_balancing_diagonal(F, n_iter)
    if native is not None:
        return native
    W = np.array(F, dtype=np.float64, copy=True)
    dim = W.shape[0]
    scale = np.ones(dim)
    # the diagonal takes no part: W[i, i] is multiplied and divided by the same factor and is left out of both norms --
    # zeroing it turns the masked sums into plain dot products (this loop is the host cost of every get_sde())
    np.fill_diagonal(W, 0.0)
    for _ in range(int(n_iter)):
        for i in range(dim):
            ci, ri = W[:, i], W[i, :]
            f = (ri.dot(ri) / ci.dot(ci)) ** 0.25
            scale[i] *= f
            ci *= f
            ri /= f
    return scale


def balance_ss(F, L, H, q, n_iter=5, return_scaling=False):
    """Balance (F, L, H, q) for numerical stability; returns (F, L, H, q).

    F <- D^-1 F D, L <- D^-1 L, H <- H D, then L and H are normalised to unit
    max-abs with q absorbing both squared factors (math_utils.py:68-81).
    return_scaling: also (d, h_max) -- the balanced state is  x_b = h_max D^-1 x  (L's normalisation is absorbed by q and
    leaves the state alone), so a stationary covariance known before balancing is known after it: balanced_covariance().
    """
    F = np.asarray(F, dtype=np.float64)
    L = np.asarray(L, dtype=np.float64)
    H = np.asarray(H, dtype=np.float64)
    q = np.asarray(q, dtype=np.float64)
    d = _balancing_diagonal(F, n_iter)
    Fb = F * d[None, :] / d[:, None]
    Lb = L / d[:, None]
    Hb = H * d[None, :]
    l_max = np.max(np.abs(Lb))
    Lb = Lb / l_max
    q = (l_max ** 2) * q
    h_max = np.max(np.abs(Hb))
    Hb = Hb / h_max
    q = (h_max ** 2) * q
    if return_scaling:
        return Fb, Lb, Hb, q, (d, h_max)
    return Fb, Lb, Hb, q


def balanced_covariance(P, scaling, Fb, Lb, qb, tol=1e-9):
    """Stationary covariance of the BALANCED model from the one of the model before balancing, P_b = h_max^2 D^-1 P D^-1
    -- what sums and products of kernels know in closed form (block diagonal, Kronecker product of the parts') and the
    reference obtains from a Lyapunov solve of the balanced system (base.py:130-183, 186-244): the same matrix, without
    the two Schur decompositions that were 45 % of a composite kernel's get_sde().  Checked against the Lyapunov
    equation; None (the caller then solves it) if the residual is not at rounding level."""
    d, h_max = scaling
    Pb = (h_max * h_max) * np.asarray(P, np.float64) / np.outer(d, d)
    Pb = 0.5 * (Pb + Pb.T)
    G = Lb @ np.atleast_2d(qb) @ Lb.T
    res = Fb @ Pb + Pb @ Fb.T + G
    scale = max(float(np.max(np.abs(G))), float(np.max(np.abs(Fb))) * float(np.max(np.abs(Pb))), 1e-300)
    if not np.all(np.isfinite(res)) or float(np.max(np.abs(res))) > tol * scale:
        return None
    return Pb


def solve_lyap_vec(F, L, Q):
    """P solving F P + P F^T + L Q L^T = 0 through (I (x) F + F (x) I) vec(P) = vec(L Q L^T)
    and a final -1/2 (P + P^T) (math_utils.py:106-120)."""
    F = np.asarray(F, dtype=np.float64)
    L = np.asarray(L, dtype=np.float64)
    Q = np.atleast_2d(np.asarray(Q, dtype=np.float64))
    dim = F.shape[0]
    C = L @ Q @ L.T
    if dim > 8 and _lyap is not None:
        # same equation by Bartels-Stewart (O(d^3) instead of the O(d^6) dense Kronecker solve: 0.3 ms against 3 ms
        # at d = 18, the host cost of every hyper-parameter setting of an MCMC run); kept only if its residual is at
        # rounding level, otherwise the reference's vectorised system below
        P = _lyap(F, -C)
        P = P + _lyap(F, -(F @ P + P @ F.T + C))             # one step of refinement on the residual
        res = F @ P + P @ F.T + C
        if np.all(np.isfinite(P)) and np.max(np.abs(res)) <= 1e-12 * max(1.0, float(np.max(np.abs(C))),
                                                                          float(np.max(np.abs(F))) * float(np.max(np.abs(P)))):
            return 0.5 * (P + P.T)
    # I (x) F + F (x) I, written as two broadcast products (the same entries in the same order of addition as np.kron's,
    # at a quarter of its cost for these d <= 8 matrices: every hyper-parameter setting of a chain pays for this)
    eye = np.eye(dim)
    big = (eye[:, None, :, None] * F[None, :, None, :] + F[:, None, :, None] * eye[None, :, None, :]).reshape(dim * dim, dim * dim)
    P = np.linalg.solve(big, C.reshape(-1)).reshape(dim, dim)
    return -0.5 * (P + P.T)
