"""GPU tests of the batched log-likelihood (pgps_gp_ll_batch_*): B hyper-parameter settings over one
series in one pair of launches -- the evaluation pattern of the reference's MCMC / grid-search
drivers (pssgp/experiments/toy_models/mcmc.py:55, co2/mcmc.py:76) at their N of 1e3..1e5."""
import numpy as np
import pytest

from oracle import np_oracle as O

pytestmark = pytest.mark.gpu


def _series(n, seed, nan_frac=0.0):
    rng = np.random.RandomState(seed)
    t = np.sort(rng.rand(n)) * (n / 50.0)
    y = np.sin(t) + 0.5 * rng.randn(n)
    if nan_frac:
        y[rng.rand(n) < nan_frac] = np.nan
    return t, y


@pytest.mark.parametrize("kname", ["m12", "m32", "m52"])
def test_batch_equals_dense_gp(kname):
    from pssgp.kernels import Matern12, Matern32, Matern52
    from pssgp.model import StateSpaceGP
    cls, spec = {"m12": (Matern12, "matern12"), "m32": (Matern32, "matern32"), "m52": (Matern52, "matern52")}[kname]
    t, y = _series(200, 1)
    rng = np.random.RandomState(2)
    thetas = np.exp(rng.uniform(-1.0, 1.0, (9, 3)))
    m = StateSpaceGP((t[:, None], y[:, None]), cls(1.0, 1.0), noise_variance=0.1, parallel=True)
    lls = m.log_likelihood_batch(thetas)
    want = np.array([O.dense_gp((spec, th[0], th[1]), t, y, th[2]) for th in thetas])
    np.testing.assert_allclose(lls, want, rtol=1e-8, atol=1e-8)
    assert m.kernel.variance == 1.0 and m.kernel.lengthscales == 1.0 and m.noise_variance == 0.1


@pytest.mark.parametrize("n,B", [(1, 3), (255, 5), (4096, 64), (30011, 33), (100000, 7)])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_batch_equals_single_evaluations(n, B, dtype):
    """Every ll of the batch equals the single-model fused call (different launch geometry, so equal
    to rounding: 1e-11 relative in fp64)."""
    from pssgp import _backend as Bk
    from pssgp.kernels import Matern32
    t, y = _series(n, n, nan_frac=0.1 if n > 1 else 0.0)
    rng = np.random.RandomState(B)
    thetas = np.exp(rng.uniform(-1.0, 1.0, (B, 3)))
    models = []
    for v, l, r in thetas:
        sde = Matern32(v, l).get_sde()
        models.append((Bk.nilpotent_form(sde.F), sde.P0, np.asarray(sde.H).reshape(-1), r))
    lls = Bk.gp_ll_batch(models, t.astype(dtype), y.astype(dtype))
    single = np.array([float(Bk.gp(f, P, H, r, t.astype(dtype), y.astype(dtype))["ll"]) for f, P, H, r in models])
    tol = 1e-11 if dtype == np.float64 else 1e-4
    np.testing.assert_allclose(lls, single, rtol=tol, atol=tol)


def test_batch_rejects_bad_models():
    from pssgp import _backend as Bk
    from pssgp.kernels import Matern32
    t, y = _series(100, 0)
    sde = Matern32(1.0, 1.0).get_sde()
    form = Bk.nilpotent_form(sde.F)
    with pytest.raises(Bk.PgpsError):
        Bk.gp_ll_batch([(form, sde.P0, np.asarray(sde.H).reshape(-1), -0.1)], t, y)     # negative noise variance
