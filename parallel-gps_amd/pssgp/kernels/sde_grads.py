"""The SDE of a kernel together with its partial derivatives with respect to the hyper-parameters, in a FROZEN state
basis -- the host half of the adjoint gradient of the log-likelihood.

What the reference obtains from TensorFlow autodiff through `get_sde()` (pssgp/kernels/*.py) and the scan
(tests/test_gp_vs_kfs.py:53-78) is split here: the device returns the adjoints of the model, i.e. of
(F, Pinf, H, R) -- parameter-independent "sufficient statistics" of one filter pass and one backward pass -- and this
module supplies d(F, Pinf, H)/d(theta_p).

The log-likelihood does not depend on the basis of the state space: (F, Pinf, H) and (S^-1 F S, S^-1 Pinf S^-T, H S)
give the same value for every invertible S.  So the basis changes that `get_sde()` makes as a function of the
hyper-parameters (balancing, math_utils.balance_ss; the powers of the lengthscale in a companion form) can be FROZEN
at the current setting when differentiating.  In such a basis every parameter of every kernel of the reference acts on
the drift as a SCALING OF TIME of one factor:

  Matern-nu, RBF   k(r) = s2 k1(r / l):             dF/dl = -F / l,   dPinf/dl = 0,        dPinf/ds2 = Pinf / s2
  Periodic         k(r) = s2 kp(r / p; l):          dF/dp = -F / p,   dPinf/d(l, s2) = d diag(q_j^2) (x) I2
  sum              block diagonal:                   derivatives of a part sit in its block
  product          F = F1 (+) F2 (Kronecker sum), Pinf = P1 (x) P2, H = H1 (x) H2: product rule

and H never moves except through those rules.  In particular every dF commutes with F (scalings of commuting Kronecker
/ block factors), which is what lets the device contract the adjoint of the transition matrices with
d expm(dt F) = dt dF expm(dt F)  once and for all:  sum_k <Fbar_k, dF_k> = <sum_k dt_k Fbar_k F_k^T, dF>  (no Frechet
derivative of the matrix exponential).

`sde_with_grads(kernel)` returns (sde, grads) with grads = [(dF, dPinf, dH), ...] in the order of
`StateSpaceGP.trainable_parameters()` (without the observation noise), in the basis of `sde`, which is the model
`kernel.get_sde()` builds (same operations in the same order).
"""
import math

import numpy as np

from .. import config as pssgp_config
from .base import ContinuousDiscreteModel, SDEProduct, SDESum, block_diag
from .math_utils import balance_ss, balanced_covariance, kron, solve_lyap_vec


def is_scalar_parameter(v):
    """A hyper-parameter value: a Python or numpy real number or a 0-d array (not a bool, not None)."""
    if v is None or isinstance(v, (bool, np.bool_)):
        return False
    if isinstance(v, (int, float, np.integer, np.floating)):
        return True
    return isinstance(v, np.ndarray) and v.ndim == 0 and v.dtype.kind in "fiu"


def leaf_parameters(kernel):
    """[(owner, attribute)] of a kernel's hyper-parameters: every leaf's variance, lengthscales and period, a Periodic
    kernel's own period before its base kernel's parameters, the parts of sums and products in order (the reference's
    gpflow Parameters in the order gpflow lists them)."""
    ps = []

    def walk(k):
        for sub in getattr(k, "kernels", ()):
            walk(sub)
        if getattr(k, "kernels", None):
            return
        ps.extend((k, a) for a in ("variance", "lengthscales", "period") if is_scalar_parameter(getattr(k, a, None)))
        if getattr(k, "base_kernel", None) is not None:
            walk(k.base_kernel)

    walk(kernel)
    return ps


def _rebalance(sde_u, grads_u, n_iter):
    """Balance an un-balanced composite model as SDESum / SDEProduct.get_sde() do (base.py:151-183, 222-244) and carry the
    derivatives into the balanced basis x_b = h_max D^-1 x (D, h_max frozen)."""
    Fb, Lb, Hb, Qb, scaling = balance_ss(sde_u.F, sde_u.L, sde_u.H, sde_u.Q, n_iter, return_scaling=True)
    Pinf = balanced_covariance(sde_u.P0, scaling, Fb, Lb, Qb)
    if Pinf is None:
        Pinf = solve_lyap_vec(Fb, Lb, Qb)
    d, h_max = scaling
    col_over_row = d[None, :] / d[:, None]
    inv_outer = (h_max * h_max) / np.outer(d, d)
    out = [(None if dF is None else dF * col_over_row, None if dP is None else dP * inv_outer,
            None if dH is None else dH * d[None, :] / h_max) for dF, dP, dH in grads_u]
    return ContinuousDiscreteModel(Pinf, Fb, Lb, Hb, Qb), out


def _periodic_dq2_dl(kernel):
    """d q2_j / d(base lengthscale) of Periodic.get_sde()'s cosine-series coefficients (periodic.py:53-81):
    q2_j = s2 sum_k b[k, j] l'^(-2k) exp(-l'^-2) 2^-k / k!,  l' = 2 l."""
    from .periodic import _get_offline_coeffs
    ell = float(kernel.base_kernel.lengthscales) * 2.0
    b, K, inv_fact = _get_offline_coeffs(kernel._order)
    base = b * ell ** (-2.0 * K) * inv_fact * math.exp(-ell ** (-2.0)) * 2.0 ** (-K)
    # d/dl' [l'^(-2k) exp(-l'^-2)] = l'^(-2k) exp(-l'^-2) (-2k / l' + 2 l'^-3);  dl'/dl = 2
    return float(kernel.base_kernel.variance) * 2.0 * np.sum(base * (-2.0 * K / ell + 2.0 * ell ** (-3.0)), axis=0)


def sde_with_grads(kernel):
    """(sde, [(dF, dPinf, dH) for every parameter of leaf_parameters(kernel)]) -- see the module docstring.  Raises
    NotImplementedError for a kernel it has no rule for (the caller then differentiates the likelihood by differences)."""
    sde, grads = _sde_with_grads(kernel)
    d = np.asarray(sde.F).shape[0]
    zF, zH = np.zeros((d, d)), np.zeros((1, d))
    return sde, [(zF if dF is None else dF, zF if dP is None else dP, zH if dH is None else dH) for dF, dP, dH in grads]


def _kron(a, b):
    return None if a is None else kron(a, b)


def _sde_with_grads(kernel):
    """sde_with_grads with the identically-zero derivatives kept as None through sums, products and balancing (a leaf's
    parameter moves one of F, Pinf, H: two thirds of the Kronecker products and embeddings of a composite kernel were
    products of zeros -- the CO2 kernel: 1.24 -> 0.8 ms per hyper-parameter setting)."""
    from .matern.common import MaternFamily
    from .periodic import Periodic
    from .rbf import RBF
    if isinstance(kernel, SDESum):
        parts = [_sde_with_grads(k) for k in kernel.kernels]
        sdes = [p[0] for p in parts]
        F = block_diag([s.F for s in sdes])
        L = block_diag([np.atleast_2d(s.L) for s in sdes])
        H = np.concatenate([np.atleast_2d(s.H) for s in sdes], axis=1)
        Q = block_diag([np.atleast_2d(s.Q) for s in sdes])
        P0 = block_diag([s.P0 for s in sdes])
        dim = F.shape[0]
        grads, lo = [], 0
        for s, g in parts:
            n = s.F.shape[0]
            for dF, dP, dH in g:
                eF = eP = eH = None
                if dF is not None:
                    eF = np.zeros((dim, dim)); eF[lo:lo + n, lo:lo + n] = dF
                if dP is not None:
                    eP = np.zeros((dim, dim)); eP[lo:lo + n, lo:lo + n] = dP
                if dH is not None:
                    eH = np.zeros((1, dim)); eH[:, lo:lo + n] = dH
                grads.append((eF, eP, eH))
            lo += n
        return _rebalance(ContinuousDiscreteModel(P0, F, L, H, Q), grads, pssgp_config.NUMBER_OF_BALANCING_STEPS)
    if isinstance(kernel, SDEProduct):
        parts = [_sde_with_grads(k) for k in kernel.kernels]
        acc, gacc = parts[0]
        for nxt, gnxt in parts[1:]:
            n1, n2 = acc.F.shape[0], nxt.F.shape[0]
            I1, I2 = np.eye(n1), np.eye(n2)
            H1, H2 = np.atleast_2d(acc.H), np.atleast_2d(nxt.H)
            g = [(_kron(dF, I2), _kron(dP, nxt.P0), _kron(dH, H2)) for dF, dP, dH in gacc]
            g += [(None if dF is None else kron(I1, dF), None if dP is None else kron(acc.P0, dP),
                   None if dH is None else kron(H1, dH)) for dF, dP, dH in gnxt]
            acc, gacc = SDEProduct._pair(acc, nxt), g
        return _rebalance(acc, gacc, pssgp_config.NUMBER_OF_BALANCING_STEPS)
    if getattr(kernel, "kernels", None):
        raise NotImplementedError(f"no derivative rule for the combination {type(kernel).__name__}")
    sde = kernel.get_sde()
    F, P0 = np.asarray(sde.F, np.float64), np.asarray(sde.P0, np.float64)
    H = np.atleast_2d(np.asarray(sde.H, np.float64))
    if isinstance(kernel, (MaternFamily, RBF)):
        by_name = {"variance": (None, P0 / float(kernel.variance), None),
                   "lengthscales": (-F / float(kernel.lengthscales), None, None)}
        return sde, [by_name[a] for _, a in leaf_parameters(kernel)]
    if isinstance(kernel, Periodic):
        dq2 = _periodic_dq2_dl(kernel)
        rules = {(id(kernel), "period"): (-F / float(kernel.period), None, None),
                 (id(kernel.base_kernel), "variance"): (None, P0 / float(kernel.base_kernel.variance), None),
                 (id(kernel.base_kernel), "lengthscales"): (None, kron(np.diag(dq2), np.eye(2)), None)}
        return sde, [rules[(id(o), a)] for o, a in leaf_parameters(kernel)]
    raise NotImplementedError(f"no derivative rule for {type(kernel).__name__}")
