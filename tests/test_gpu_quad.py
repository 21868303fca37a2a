"""GPU parity of the quad-cooperative level-1 kernels (csrc/pgps_qc.hip.h: four lanes own a chain of steps, lane q holds
columns 2q, 2q+1 of every operand, products are quad_perm broadcasts + v_pk_fma_f32; fp32, state dimensions 5..8 -- config
c3's RBF order 6) against the CPU oracle in fp64.  Forced with family 4; the chain totals travel through the row-
cooperative family's scans and segment protocol."""
import numpy as np
import pytest

from oracle import np_oracle as O
from tests.conftest import make_times, relerr, sample_series

pytestmark = pytest.mark.gpu
TOL32 = 1e-3        # north_star: 1e-3 relative in fp32


def _rbf(order, ls=0.7):
    from pssgp.kernels import RBF
    return RBF(variance=1., lengthscales=ls, order=order, balancing_iter=10)


def _oracle_all(ssm, y):
    fms, fPs, ll = O.kf(ssm, y, True)
    sms, sPs = O.kfs(ssm, y)
    return dict(fms=fms, fPs=fPs, sms=sms, sPs=sPs, ll=np.array([ll]))


def _gpu_all(ssm, y):
    from pssgp import _backend as B
    ssm_t = tuple(np.asarray(a, dtype=np.float32) for a in ssm)
    sms, sPs, fms, fPs, ll = B.pkfs(ssm_t, np.asarray(y, np.float32), return_filtered=True, return_loglikelihood=True)
    return dict(fms=fms, fPs=fPs, sms=sms, sPs=sPs, ll=np.array([float(ll)]))


def _check(got, want, tol):
    for name in want:
        e = relerr(np.asarray(got[name], np.float64), want[name])
        assert e < tol, f"{name}: rel err {e:.3e} >= {tol}"


@pytest.fixture
def quad_family():
    from pssgp import _backend as B
    ctx = B.get_context()
    ctx.set_family(4)
    yield ctx
    ctx.set_family(0)
    ctx.set_chunk(0)


@pytest.mark.parametrize("order", [5, 6, 7, 8])
def test_quad_kernels_match_oracle(quad_family, order):
    t = make_times(5000, seed=order)
    ssm = O.get_ssm(_rbf(order).get_sde(), t, 0.1)
    y = sample_series(ssm, seed=order, nan_frac=0.2)
    _check(_gpu_all(ssm, y), _oracle_all(ssm, y), TOL32)


@pytest.mark.parametrize("n,lw", [(1, 8), (2, 8), (7, 8), (9, 8), (127, 8), (129, 8), (1024, 64), (1025, 64), (2200, 7),
                                  (4097, 1), (70000, 16), (70001, 0)])
def test_quad_ragged_lengths(quad_family, n, lw):
    """Chain boundaries, partially filled waves (sixteen chains each), steps beyond the end of the series inside the last
    chain, the first step of the series, missing observations."""
    quad_family.set_chunk(lw)
    t = make_times(n, seed=n % 97)
    ssm = O.get_ssm(_rbf(6).get_sde(), t, 0.1)
    y = sample_series(ssm, seed=3, nan_frac=0.1 if n > 4 else 0.0)
    from oracle import c_oracle as C
    cf, cP, cs, csP, cll = C.kfs(ssm, y)
    _check(_gpu_all(ssm, y), dict(fms=cf, fPs=cP, sms=cs, sPs=csP, ll=np.array([cll])), TOL32)


def test_quad_filter_only(quad_family):
    from pssgp import _backend as B
    t = make_times(30000, seed=5)
    ssm = O.get_ssm(_rbf(6).get_sde(), t, 0.1)
    y = sample_series(ssm, seed=9, nan_frac=0.3)
    fms, fPs, ll = B.pkf(tuple(np.asarray(a, np.float32) for a in ssm), np.asarray(y, np.float32), return_loglikelihood=True)
    of, oP, oll = O.kf(ssm, y, True)
    assert relerr(np.asarray(fms, np.float64), of) < TOL32 and relerr(np.asarray(fPs, np.float64), oP) < TOL32
    assert abs(float(ll) - oll) < TOL32 * abs(oll)


def test_quad_all_missing_and_first_missing(quad_family):
    t = make_times(3000, seed=2)
    ssm = O.get_ssm(_rbf(6).get_sde(), t, 0.1)
    y = sample_series(ssm, seed=1, nan_frac=0.0)
    y1 = y.copy(); y1[0] = np.nan
    _check(_gpu_all(ssm, y1), _oracle_all(ssm, y1), TOL32)
    y2 = np.full_like(y, np.nan)
    got, want = _gpu_all(ssm, y2), _oracle_all(ssm, y2)
    for name in ("fms", "fPs", "sms", "sPs"):
        assert np.max(np.abs(np.asarray(got[name], np.float64) - want[name])) < 1e-3 * max(1.0, np.max(np.abs(want[name])))
    assert abs(got["ll"][0]) < 1e-6
