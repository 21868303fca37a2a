"""GPU tests of predict_f on the device (pgps_gp_predict_*: merge of sorted times + missing-marking +
fused filter/smoother + projection at the query rows; pssgp/model.py:15-55,92-111 of the reference)."""
import numpy as np
import pytest

from oracle import np_oracle as O
from tests.conftest import relerr

pytestmark = pytest.mark.gpu


def _kernels():
    from pssgp.kernels import Matern12, Matern32, Matern52
    return {"m12": (Matern12, "matern12"), "m32": (Matern32, "matern32"), "m52": (Matern52, "matern52")}


def _data(n, k, seed):
    rng = np.random.RandomState(seed)
    t = np.sort(rng.rand(n)) * (n / 80.0)
    y = np.sin(3.0 * t) + 0.3 * rng.randn(n)
    tq = np.sort(rng.rand(k)) * (n / 80.0) * 1.1          # some queries beyond the last observation
    return t, y, tq


@pytest.mark.parametrize("kname", ["m12", "m32", "m52"])
@pytest.mark.parametrize("n,k", [(200, 50), (50, 200), (64, 64), (1, 5), (5, 1)])
def test_predict_equals_dense_gp(kname, n, k):
    """The reference's equivalence test (tests/test_gp_vs_kfs.py:80-99, 1e-6) through the on-device
    predict_f, both when the training series is the longer array and when the queries are."""
    from pssgp.model import StateSpaceGP
    cls, spec = _kernels()[kname]
    t, y, tq = _data(n, k, 7 * n + k)
    m = StateSpaceGP((t[:, None], y[:, None]), cls(1.2, 0.6), noise_variance=0.15, parallel=True)
    mean, var = m.predict_f(tq[:, None])
    assert mean.shape == (k, 1) and var.shape == (k, 1)
    _, mean_gp, var_gp = O.dense_gp((spec, 1.2, 0.6), t, y, 0.15, tq)
    np.testing.assert_allclose(mean[:, 0], mean_gp, atol=1e-8, rtol=1e-8)
    np.testing.assert_allclose(var[:, 0], var_gp, atol=1e-8, rtol=1e-8)


@pytest.mark.parametrize("n,k", [(300, 120), (120, 300), (200, 200)])
def test_predict_with_tied_times_matches_oracle_merge(n, k):
    """Queries AT training times and repeated query times: the merge order on ties (shorter array
    first) is the reference's, so results equal the oracle's merge + sequential smoother row by row."""
    from pssgp import _backend as B
    from pssgp.kernels import Matern32
    rng = np.random.RandomState(n + k)
    t = np.sort(np.round(rng.rand(n) * 20.0, 1))          # many exact ties, also inside each array
    y = np.cos(t) + 0.2 * rng.randn(n)
    tq = np.sort(np.round(rng.rand(k) * 22.0, 1))
    kern = Matern32(0.9, 1.5)
    sde = kern.get_sde()
    mean, var, ll = B.gp_predict(B.nilpotent_form(sde.F), sde.P0, sde.H, 0.2, t, y, tq)
    mean_o, var_o = O.ssgp_predict_f(sde, t, y, 0.2, tq, parallel=False)
    assert np.max(np.abs(mean - mean_o)) < 1e-9 and np.max(np.abs(var - var_o)) < 1e-9
    assert abs(ll - O.ssgp_log_likelihood(sde, t, y, 0.2, parallel=False)) < 1e-9 * abs(ll)


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-9), (np.float32, 2e-3)])
def test_predict_large_against_full_smoother(dtype, tol):
    """N = 300k training points, 100k queries: the projected outputs equal H sm, H sP H^T of the full
    smoothed posterior (pgps_gp_* on the host-merged series) at the query rows."""
    from pssgp import _backend as B
    from pssgp.kernels import Matern52
    n, k = 300_000, 100_000
    rng = np.random.RandomState(1)
    t = np.cumsum(0.05 * (0.5 + rng.rand(n)))
    y = np.sin(0.3 * t) + 0.3 * rng.randn(n)
    y[rng.rand(n) < 0.1] = np.nan
    tq = np.sort(rng.rand(k)) * t[-1]
    sde = Matern52(1.0, 2.0).get_sde()
    form = B.nilpotent_form(sde.F)
    mean, var, ll = B.gp_predict(form, sde.P0, sde.H, 0.1, t.astype(dtype), y.astype(dtype), tq.astype(dtype))
    all_t, all_y, flags = O.merge_sorted(t.astype(dtype), tq.astype(dtype), (y.astype(dtype), np.full(k, np.nan, dtype)),
                                         (np.zeros(n, bool), np.ones(k, bool)))
    res = B.gp(form, sde.P0, sde.H, 0.1, all_t, all_y, want_smoothed=True)
    h = np.asarray(sde.H, np.float64).reshape(-1)
    mean_f = res["sms"][flags].astype(np.float64) @ h
    var_f = np.einsum("i,nij,j->n", h, res["sPs"][flags].astype(np.float64), h)
    assert relerr(mean, mean_f) < tol and relerr(var, var_f) < tol
    assert abs(ll - float(res["ll"])) <= tol * abs(ll)
