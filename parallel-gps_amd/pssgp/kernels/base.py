"""State-space kernel base classes: the `get_sde / get_ssm / get_spec` contract, `+`, `*`.

Reference: pssgp/kernels/base.py (ContinuousDiscreteModel 15, get_lssm_spec 18-26,
_get_ssm 29-47, SDEKernelMixin 50-103, SDESum 130-183, SDEProduct 186-244).

The continuous model (P0, F, L, H, Q) is built on the host in numpy (tiny, once per call).
The batched discretisation  Fs[k] = expm(dt_k F),  Qs[k] = P0 - Fs[k] P0 Fs[k]^T  runs on
the GPU (`pgps_discretise_*`, csrc/pgps_discretise.hip).
"""
from collections import namedtuple

import numpy as np

from .. import config as pssgp_config
from ..kalman.base import LGSSM
from .math_utils import balance_ss, balanced_covariance, kron, solve_lyap_vec

ContinuousDiscreteModel = namedtuple("ContinuousDiscreteModel", ["P0", "F", "L", "H", "Q"])

ArraySpec = namedtuple("ArraySpec", ["shape", "dtype"])


def get_lssm_spec(dim, T):
    """Shapes/dtypes of an LGSSM with `T` steps (None = dynamic); base.py:18-26."""
    dtype = pssgp_config.default_float()
    return LGSSM(ArraySpec((dim, dim), dtype), ArraySpec((T, dim, dim), dtype),
                 ArraySpec((T, dim, dim), dtype), ArraySpec((1, dim), dtype),
                 ArraySpec((1, 1), dtype))


def _get_ssm(sde, ts, R, t0=0.):
    """LTI discretisation on the GPU (replaces base.py:29-47).

    dts = diff([t0; ts]); Fs = expm(dts F); Qs = P0 - Fs P0 Fs^T, the stationary form of
    the reference's matrix-fraction expression (equal to round-off because every kernel's
    P0 solves the Lyapunov equation; DESIGN.md "discretisation").
    """
    from .. import _backend
    dtype = pssgp_config.default_float()
    ts = np.ascontiguousarray(np.asarray(ts, dtype=dtype).reshape(-1))
    F = np.ascontiguousarray(sde.F, dtype=dtype)
    P0 = np.ascontiguousarray(sde.P0, dtype=dtype)
    LQL = np.asarray(sde.L, np.float64) @ np.atleast_2d(np.asarray(sde.Q, np.float64)) \
        @ np.asarray(sde.L, np.float64).T
    Fd, Pd = np.asarray(sde.F, np.float64), np.asarray(sde.P0, np.float64)
    resid = Fd @ Pd + Pd @ Fd.T + LQL
    scale = max(1.0, float(np.max(np.abs(LQL))), float(np.max(np.abs(Fd))) * float(np.max(np.abs(Pd))))
    if float(np.max(np.abs(resid))) > 1e-8 * scale:
        raise NotImplementedError(
            "P0 is not the stationary covariance of (F, L, Q): the GPU discretisation uses "
            "Qs = P0 - Fs P0 Fs^T, which needs F P0 + P0 F^T + L Q L^T = 0")
    Fs, Qs = _backend.discretise(F, P0, ts, float(t0))
    H = np.asarray(sde.H, dtype=dtype).reshape(1, -1)
    R = np.asarray(R, dtype=dtype).reshape(1, 1)
    return LGSSM(P0, Fs, Qs, H, R)


class Kernel:
    """Minimal stand-in for gpflow.kernels.Kernel: hyper-parameters are plain floats."""

    # Bumped by every assignment to an attribute of any kernel object: what StateSpaceGP memoises per hyper-parameter
    # setting is revalidated only when this has moved (an evaluation repeated at the same setting costs one comparison).
    _version = 0
    # ... and this one only by assignments that can change the STRUCTURE of a kernel (anything but a new value of a
    # hyper-parameter that already had one): the list of leaf parameters and the classes / orders of the nodes are
    # rebuilt only when it has moved -- a sampler's step assigns three floats and pays for three floats.
    _struct_version = 0
    _PARAMETERS = ("variance", "lengthscales", "period")

    def __setattr__(self, name, value):
        # hyper-parameters are kept as Python floats whatever they are assigned as (int, numpy scalar, 0-d array): the
        # memo keys, the closed-form derivative rules and `trainable_parameters()` then see one type
        if name in Kernel._PARAMETERS and value is not None and not isinstance(value, float):
            arr = np.asarray(value)
            if arr.ndim == 0 and arr.dtype.kind in "fiu":
                value = float(arr)
        Kernel._version += 1
        if not (name in Kernel._PARAMETERS and isinstance(value, float) and isinstance(self.__dict__.get(name), float)):
            Kernel._struct_version += 1
        object.__setattr__(self, name, value)

    def K(self, X, X2=None):
        raise NotImplementedError

    def K_diag(self, X):
        X = np.asarray(X, dtype=np.float64).reshape(-1)
        return np.array([self.K(x[None], x[None])[0, 0] for x in X])

    def __call__(self, X, X2=None, full_cov=True):
        return self.K(X, X2) if full_cov else self.K_diag(X)


def _pairwise_dist(X, X2):
    X = np.asarray(X, dtype=np.float64).reshape(-1)
    X2 = X if X2 is None else np.asarray(X2, dtype=np.float64).reshape(-1)
    return np.abs(X[:, None] - X2[None, :])


def _tree_ids(k):
    """The identities of a kernel's nodes: a part swapped IN PLACE (`k.kernels[0] = ...`, no attribute assignment, so no
    version bump) must not find the old tree's memoised SDE."""
    subs = getattr(k, "kernels", None)
    base = getattr(k, "base_kernel", None)
    if not subs and base is None:
        return id(k)
    return (id(k), tuple(_tree_ids(x) for x in subs) if subs else (), id(base) if base is not None else 0)


class SDEKernelMixin:
    """`get_sde()` -> continuous model; `get_ssm(ts, R, t0)` -> LGSSM (base.py:50-103)."""

    def __init__(self, t0=0., **_kwargs):
        self.t0 = t0

    def __init_subclass__(cls, **kwargs):
        """Every class's own `get_sde` is memoised per kernel object: the continuous model is rebuilt only after an
        attribute of SOME kernel has been assigned (`Kernel._version`; conservative: a composite's memo must fall with its
        children's parameters), the number of balancing sweeps or the default float has changed.  One kernel object shared by
        many short-lived models -- the reference's speed protocol, experiments/toy_models/speed_and_stability.py:56-87 --
        then pays for the polynomial roots, balancing sweeps and Lyapunov solve of its SDE once (RBF order 6: 0.25 ms).
        The returned model is shared: treat its arrays as read-only."""
        super().__init_subclass__(**kwargs)
        f = cls.__dict__.get("get_sde")
        if f is None or getattr(f, "_memoised", False):
            return

        def get_sde(self, _f=f):
            key = (Kernel._version, pssgp_config.NUMBER_OF_BALANCING_STEPS, pssgp_config.default_float(), _tree_ids(self))
            memo = self.__dict__.get("_sde_memo")
            if memo is not None and memo[0] == key:
                return memo[1]
            sde = _f(self)
            object.__setattr__(self, "_sde_memo", (key, sde))          # (not through Kernel.__setattr__: no version bump)
            return sde

        get_sde._memoised = True
        get_sde.__doc__ = f.__doc__
        cls.get_sde = get_sde

    def get_sde(self):
        raise NotImplementedError

    def get_ssm(self, ts, R, t0=0.):
        return _get_ssm(self.get_sde(), ts, R, t0)

    def get_spec(self, T):
        raise NotImplementedError

    def __add__(self, other):
        return SDESum([self, other])

    def __mul__(self, other):
        return SDEProduct([self, other])


def block_diag(arrs):
    """Block diagonal of (possibly non-square) 2-D arrays (base.py:113-127)."""
    rows = sum(a.shape[0] for a in arrs)
    cols = sum(a.shape[1] for a in arrs)
    out = np.zeros((rows, cols), dtype=np.float64)
    r = c = 0
    for a in arrs:
        out[r:r + a.shape[0], c:c + a.shape[1]] = a
        r += a.shape[0]
        c += a.shape[1]
    return out


class _Combination(SDEKernelMixin, Kernel):
    def __init__(self, kernels, name=None, **kwargs):
        if not all(isinstance(k, SDEKernelMixin) for k in kernels):
            raise TypeError("can only combine SDE Kernel instances")
        # flatten nested combinations of the same type, as gpflow.kernels.Combination does
        flat = []
        for k in kernels:
            flat.extend(k.kernels if type(k) is type(self) else [k])
        self.kernels = flat
        self.name = name
        SDEKernelMixin.__init__(self, **kwargs)

    def _dims(self, T):
        dims = []
        for kernel in self.kernels:
            spec = kernel.get_spec(T)
            if spec is None:
                return None
            dims.append(spec.P0.shape[-1])
        return dims


class SDESum(_Combination):
    """k1 + k2: block-diagonal SDE, then balance + Lyapunov (base.py:130-183)."""

    def K(self, X, X2=None):
        return sum(k.K(X, X2) for k in self.kernels)

    def get_spec(self, T):
        dims = self._dims(T)
        return None if dims is None else get_lssm_spec(int(sum(dims)), T)

    def get_sde(self):
        parts = [k.get_sde() for k in self.kernels]
        F = block_diag([p.F for p in parts])
        L = block_diag([np.atleast_2d(p.L) for p in parts])
        H = np.concatenate([np.atleast_2d(p.H) for p in parts], axis=1)
        Q = block_diag([np.atleast_2d(p.Q) for p in parts])
        Fb, Lb, Hb, Qb, scaling = balance_ss(F, L, H, Q, pssgp_config.NUMBER_OF_BALANCING_STEPS, return_scaling=True)
        # independent parts: the stationary covariance is block diagonal before balancing
        Pinf = balanced_covariance(block_diag([p.P0 for p in parts]), scaling, Fb, Lb, Qb)
        if Pinf is None:
            Pinf = solve_lyap_vec(Fb, Lb, Qb)
        return ContinuousDiscreteModel(Pinf, Fb, Lb, Hb, Qb)


class SDEProduct(_Combination):
    """k1 * k2: Kronecker-sum dynamics, Kronecker observation (base.py:186-244)."""

    def K(self, X, X2=None):
        out = None
        for k in self.kernels:
            out = k.K(X, X2) if out is None else out * k.K(X, X2)
        return out

    def get_spec(self, T):
        dims = self._dims(T)
        return None if dims is None else get_lssm_spec(int(np.prod(dims)), T)

    @staticmethod
    def _pair(s1, s2):
        """Unbalanced product of two SDEs (base.py:200-220,235-239): F = F1 (+) F2,
        diffusion = G1 (x) P2 + P1 (x) G2 with G = L Q L^T, H = H1 (x) H2."""
        n1, n2 = s1.F.shape[0], s2.F.shape[0]
        F = kron(s1.F, np.eye(n2)) + kron(np.eye(n1), s2.F)
        G1 = s1.L @ np.atleast_2d(s1.Q) @ s1.L.T
        G2 = s2.L @ np.atleast_2d(s2.Q) @ s2.L.T
        Q = kron(G1, s2.P0) + kron(s1.P0, G2)
        H = kron(np.atleast_2d(s1.H), np.atleast_2d(s2.H))
        P0 = kron(s1.P0, s2.P0)
        return ContinuousDiscreteModel(P0, F, np.eye(n1 * n2), H, Q)

    def get_sde(self):
        sdes = [k.get_sde() for k in self.kernels]
        acc = sdes[0]
        for nxt in sdes[1:]:
            acc = self._pair(acc, nxt)
        Fb, Lb, Hb, Qb, scaling = balance_ss(acc.F, acc.L, acc.H, acc.Q, pssgp_config.NUMBER_OF_BALANCING_STEPS,
                                             return_scaling=True)
        # a product of independent stationary processes: P0 = P1 (x) P2 before balancing (carried by _pair)
        Pinf = balanced_covariance(acc.P0, scaling, Fb, Lb, Qb)
        if Pinf is None:
            Pinf = solve_lyap_vec(Fb, Lb, Qb)
        return ContinuousDiscreteModel(Pinf, Fb, Lb, Hb, Qb)
