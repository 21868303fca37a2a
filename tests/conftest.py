import os
import sys

import numpy as np
import pytest

# PyTorch-ROCm bundles its own HIP/HSA runtime.  A process that uses both torch and libpgps must
# load torch FIRST so that libpgps' libamdhip64 dependency resolves to the copy already in the
# process (two HSA runtimes in one process = "No HIP GPUs are available").  Only the multi-GPU
# driver tests and bench.py use torch; the product library itself does not.
try:  # pragma: no cover
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "parallel-gps_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_present():
    try:
        from pssgp import _backend
        import ctypes
        lib = _backend.load_library()
        n = ctypes.c_int(0)
        lib.pgps_device_count(ctypes.byref(n))
        return n.value > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU (or without the built library) must fail loudly, not skip:
    # the GPU tests themselves raise from pssgp._backend in that case.
    pass


# ----------------------------------------------------------------------------------------------
# shared synthetic data: irregular times, latent drawn from the SSM prior
# ----------------------------------------------------------------------------------------------
def make_times(n, seed=0, delta=0.05):
    rng = np.random.default_rng(seed)
    return np.cumsum(delta * rng.uniform(0.5, 1.5, size=n))


def sample_series(ssm, seed=0, nan_frac=0.0):
    """Draw x_0 ~ N(0, P0), x_k = F_k x_{k-1} + N(0, Q_k), y_k = H x_k + N(0, R)."""
    P0, Fs, Qs, H, R = ssm
    rng = np.random.default_rng(seed)
    n, d = Fs.shape[0], Fs.shape[1]
    h = np.asarray(H, dtype=np.float64).reshape(d)
    r = float(np.asarray(R).reshape(()))

    def chol_psd(A):
        A = 0.5 * (np.asarray(A, dtype=np.float64) + np.asarray(A, dtype=np.float64).T)
        w, V = np.linalg.eigh(A)
        return V * np.sqrt(np.clip(w, 0.0, None))

    x = chol_psd(P0) @ rng.standard_normal(d)
    ys = np.empty(n)
    z = rng.standard_normal((n, d))
    e = rng.standard_normal(n)
    for k in range(n):
        x = np.asarray(Fs[k], dtype=np.float64) @ x + chol_psd(Qs[k]) @ z[k]
        ys[k] = h @ x + np.sqrt(r) * e[k]
    if nan_frac > 0:
        ys[rng.random(n) < nan_frac] = np.nan
    return ys


def sample_series_fast(ssm, seed=0, nan_frac=0.0):
    """Cheap stand-in for very long series: smooth signal + noise (values only need to be
    plausible observations; parity is judged against the oracle on the same numbers)."""
    P0, Fs, Qs, H, R = ssm
    rng = np.random.default_rng(seed)
    n = Fs.shape[0]
    t = np.arange(n) * 0.05
    ys = np.sin(0.7 * t) + 0.5 * np.sin(0.13 * t + 1.0) + np.sqrt(float(np.asarray(R).reshape(()))) * \
        rng.standard_normal(n)
    if nan_frac > 0:
        ys[rng.random(n) < nan_frac] = np.nan
    return ys


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))


@pytest.fixture(scope="session")
def kernel_zoo():
    """(name, kernel factory, oracle dense spec or None, tolerance vs dense GP) -- the seven
    covariance functions of the reference's equivalence test (tests/test_gp_vs_kfs.py:33-41),
    at state dims the lane-chunk kernels cover."""
    from pssgp.kernels import Matern12, Matern32, Matern52, RBF, Periodic, SquaredExponential
    return [
        ("matern12", lambda: Matern12(variance=1., lengthscales=0.5), ("matern12", 1., 0.5), 1e-6),
        ("matern32", lambda: Matern32(variance=1., lengthscales=0.5), ("matern32", 1., 0.5), 1e-6),
        ("matern52", lambda: Matern52(variance=1., lengthscales=0.5), ("matern52", 1., 0.5), 1e-6),
        ("rbf6", lambda: RBF(variance=1., lengthscales=0.5, order=6, balancing_iter=10), None, None),
        ("periodic2", lambda: Periodic(SquaredExponential(1., 0.5), period=0.5, order=2), None, None),
        ("m32+m52", lambda: Matern32(variance=1., lengthscales=0.5) + Matern52(variance=1., lengthscales=0.5),
         ("sum", [("matern32", 1., 0.5), ("matern52", 1., 0.5)]), 1e-6),
        ("m32*m52", lambda: Matern32(variance=1., lengthscales=0.5) * Matern52(variance=1., lengthscales=0.5),
         ("prod", [("matern32", 1., 0.5), ("matern52", 1., 0.5)]), 1e-6),
    ]
