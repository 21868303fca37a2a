"""The device-resident series (pgps_series_*, include/pgps.h) and the one-launch kernel for short series (k_gp_one):
same results as the host entry points, which are pinned to the oracle elsewhere (tests/test_gpu_predict.py, test_gpu_grad.py)."""
import numpy as np
import pytest

from oracle import np_oracle as O

pytestmark = pytest.mark.gpu


def _data(n, seed=0):
    rng = np.random.default_rng(seed)
    t = np.sort(rng.uniform(0, 0.01 * n + 1, n))
    y = np.sin(t) + 0.3 * rng.standard_normal(n)
    return t, y


@pytest.mark.parametrize("kern", ["m12", "m32", "m52"])
@pytest.mark.parametrize("n", [1, 5, 300, 4097, 8192, 20000])
def test_series_matches_host_entry_points(kern, n):
    from pssgp import _backend as B
    from pssgp.kernels import Matern12, Matern32, Matern52
    k = {"m12": Matern12, "m32": Matern32, "m52": Matern52}[kern](1.3, 0.7)
    sde = k.get_sde()
    form = B.nilpotent_form(sde.F)
    t, y = _data(n, seed=n)
    tq = np.sort(np.random.default_rng(1).uniform(t[0] - 1, t[-1] + 1, max(3, n // 3)))
    ref_ll = float(B.gp(form, sde.P0, sde.H, 0.1, t, y)["ll"])
    ref_mean, ref_var, _ = B.gp_predict(form, sde.P0, sde.H, 0.1, t, y, tq)
    ser = B.Series(t, y)
    packed = B.Series.pack(form, sde.P0, sde.H)
    assert abs(ser.gp_ll(packed, 0.1) - ref_ll) <= 1e-12 * abs(ref_ll)
    ser.set_queries(tq)
    mean, var, ll = ser.gp_predict(packed, 0.1)
    assert np.max(np.abs(mean - ref_mean)) < 1e-11 and np.max(np.abs(var - ref_var)) < 1e-11
    assert abs(ll - ref_ll) <= 1e-12 * abs(ref_ll)
    # and against the oracle directly on a small case
    if n <= 300:
        om, ov = O.ssgp_predict_f(sde, t, y, 0.1, tq, parallel=False)
        assert np.max(np.abs(mean - om)) < 1e-8 and np.max(np.abs(var - ov)) < 1e-8
    ser.close()


@pytest.mark.parametrize("n", [1, 7, 255, 256, 257, 1000, 4096, 8192])
def test_one_launch_equals_three_launches(n):
    """k_gp_one (one workgroup, one launch) against the three-launch path on the same series: filtered and smoothed moments,
    log-likelihood, projected posterior."""
    from pssgp import _backend as B
    from pssgp.kernels import Matern32, Matern52
    ctx = B.get_context()
    t, y = _data(n, seed=3 + n)
    y[::7] = np.nan
    tq = np.sort(np.random.default_rng(2).uniform(t[0], t[-1] + 0.5, max(2, n // 4)))
    try:
        for k in (Matern32(1.0, 0.5), Matern52(0.8, 0.9)):
            sde = k.get_sde()
            form = B.nilpotent_form(sde.F)
            outs = []
            for mode in (0, -1):
                ctx.set_one_launch(mode)
                r = B.gp(form, sde.P0, sde.H, 0.1, t, y, want_smoothed=True)
                p = B.gp_predict(form, sde.P0, sde.H, 0.1, t, y, tq)
                outs.append((r, p))
            (r0, p0), (r1, p1) = outs
            for name in ("fms", "fPs", "sms", "sPs"):
                assert np.max(np.abs(r0[name] - r1[name])) <= 1e-10 * max(1.0, np.max(np.abs(r0[name]))), name
            assert abs(float(r0["ll"]) - float(r1["ll"])) <= 1e-11 * max(1.0, abs(float(r0["ll"])))
            assert np.max(np.abs(p0[0] - p1[0])) < 1e-10 and np.max(np.abs(p0[1] - p1[1])) < 1e-10
    finally:
        ctx.set_one_launch(-1)


def test_statespacegp_uses_the_resident_series():
    """StateSpaceGP on the device-resident series: objective, gradient and predict_f equal the host-staged calls; a
    changed hyper-parameter is picked up (the memo is keyed on the parameters)."""
    from pssgp import _backend as B
    from pssgp.kernels import Matern32
    from pssgp.model import StateSpaceGP
    t, y = _data(3000, seed=11)
    tq = np.linspace(t[0], t[-1], 700)
    gp = StateSpaceGP((t[:, None], y[:, None]), Matern32(1.0, 0.5), noise_variance=0.1, parallel=True)
    ll = float(gp.maximum_log_likelihood_objective())
    assert not getattr(gp, "_series", None)         # the first evaluation of a model: host-array entry points, no allocations
    gp.kernel.variance = 1.0                        # (drops the memoised objective)
    assert float(gp.maximum_log_likelihood_objective()) == pytest.approx(ll, rel=1e-12)
    assert gp._series and gp._series.N == 3000      # ... from the second on: resident
    sde = gp.kernel.get_sde()
    form = B.nilpotent_form(sde.F)
    assert abs(ll - float(B.gp(form, sde.P0, sde.H, 0.1, t, y)["ll"])) < 1e-10 * abs(ll)
    m, v = gp.predict_f(tq[:, None])
    rm, rv, _ = B.gp_predict(form, sde.P0, sde.H, 0.1, t, y, tq)
    assert np.max(np.abs(m[:, 0] - rm)) < 1e-11 and np.max(np.abs(v[:, 0] - rv)) < 1e-11
    l2, g = gp.log_likelihood_and_grad()
    rl, rg = B.gp_ll_grad(gp._grad_blocks(), t, y)
    assert abs(l2 - rl) < 1e-10 * abs(rl) and np.max(np.abs(g - rg)) < 1e-9 * max(1.0, np.max(np.abs(rg)))
    gp.kernel.lengthscales = 0.8
    ll3 = float(gp.maximum_log_likelihood_objective())
    sde = gp.kernel.get_sde()
    assert abs(ll3 - float(B.gp(B.nilpotent_form(sde.F), sde.P0, sde.H, 0.1, t, y)["ll"])) < 1e-10 * abs(ll3)
    assert abs(ll3 - ll) > 1e-3


@pytest.mark.parametrize("kname,n", [("rbf6", 700), ("rbf6", 20000), ("c5", 3000), ("periodic7", 900), ("co2_d18", 400)])
def test_series_lti_calls_match_host_entry_points(kname, n):
    """pgps_series_lti_* (any kernel's LTI model on the resident series) against pgps_lti_ll_* / _predict_* / _ll_batch_* on
    host arrays -- row-cooperative (d <= 16) and wave-cooperative (d = 18) paths -- and the model routes through them."""
    from pssgp import _backend as B
    from pssgp.kernels import Matern32, Matern52, Periodic, RBF, SquaredExponential
    from pssgp.model import StateSpaceGP
    k = {"rbf6": lambda: RBF(1.0, 0.8, order=6, balancing_iter=10),
         "c5": lambda: Periodic(SquaredExponential(1., 1.), period=1., order=1) * Matern32(1., 1.) + Matern52(1., 1.),
         "periodic7": lambda: Periodic(SquaredExponential(1., 0.5), period=0.5, order=7),
         "co2_d18": lambda: Periodic(SquaredExponential(1., 1.), period=1., order=3) * Matern32(1., 2.) + RBF(1., 1., order=4)}[kname]()
    sde = k.get_sde()
    d = np.asarray(sde.F).shape[0]
    t, y = _data(n, seed=n + d)
    y[::11] = np.nan
    tq = np.sort(np.random.default_rng(5).uniform(t[0] - 0.5, t[-1] + 0.5, max(3, n // 5)))
    ref_ll = B.lti_ll(sde.F, sde.P0, sde.H, 0.1, t, y)
    ref_mean, ref_var, _ = B.lti_predict(sde.F, sde.P0, sde.H, 0.1, t, y, tq)
    ser = B.Series(t, y)
    assert ser.has_lti
    assert abs(ser.lti_ll(sde.F, sde.P0, sde.H, 0.1) - ref_ll) <= 1e-12 * abs(ref_ll)
    ser.set_queries(tq)
    mean, var, ll = ser.lti_predict(sde.F, sde.P0, sde.H, 0.1)
    assert np.max(np.abs(mean - ref_mean)) < 1e-10 and np.max(np.abs(var - ref_var)) < 1e-10
    assert abs(ll - ref_ll) <= 1e-11 * abs(ref_ll)
    if d <= B.LTI_BATCH_DIM_MAX:
        models = [(sde.F, sde.P0 * s, sde.H, r) for s, r in ((1.0, 0.1), (1.3, 0.2), (0.7, 0.05))]
        assert np.max(np.abs(ser.lti_ll_batch(models) - B.lti_ll_batch(models, t, y))) <= 1e-11 * abs(ref_ll)
    ser.close()
    # the model: objective, prediction (and the objective again, handed over by the prediction), gradient
    m = StateSpaceGP((t[:, None], y[:, None]), k, noise_variance=0.1, parallel=True)
    assert abs(float(m.maximum_log_likelihood_objective()) - ref_ll) <= 1e-11 * abs(ref_ll)
    pm, pv = m.predict_f(tq[:, None])
    assert np.max(np.abs(pm[:, 0] - ref_mean)) < 1e-10 and np.max(np.abs(pv[:, 0] - ref_var)) < 1e-10
    assert abs(float(m.maximum_log_likelihood_objective()) - ref_ll) <= 1e-11 * abs(ref_ll)
    if d <= B.LTI_BATCH_DIM_MAX:
        ll_g, g = m.log_likelihood_and_grad()
        assert abs(ll_g - ref_ll) <= 1e-10 * abs(ref_ll) and np.all(np.isfinite(g))
