"""StateSpaceGP: GP regression through the state-space form of the kernel.

Mirrors pssgp/model.py:58-117 of the reference: same constructor, `predict_f`,
`maximum_log_likelihood_objective`, `training_loss`; `parallel=True` runs the HIP
associative-scan path (csrc/), `parallel=False` the sequential host recursion.  Data and
results are numpy arrays; hyper-parameters are plain floats (no GPflow `Parameter`s); the
gradient of the log-likelihood comes from `log_likelihood_and_grad` (forward-mode duals inside
the scan kernels) instead of TensorFlow autodiff.
"""
import numpy as np

from . import config
from .kernels.base import Kernel, _tree_ids
from .kernels.sde_grads import leaf_parameters
from .kalman.parallel import pkf, pkfs
from .kalman.sequential import kf, kfs


def _merge_sorted(a, b, *args):
    """Merge two sorted 1-D arrays and any number of (a_x, b_x) payload pairs.

    As pssgp/model.py:15-55: the shorter array is scattered into the longer at
    arange + searchsorted(longer, shorter) (left side: on ties the shorter array's
    point comes first).
    """
    a = np.asarray(a)
    b = np.asarray(b)
    assert a.ndim == b.ndim == 1
    if a.shape[0] < b.shape[0]:
        a, b = b, a
        args = tuple((j, i) for i, j in args)
    n_long, n_short = a.shape[0], b.shape[0]
    where_short = np.arange(n_short) + np.searchsorted(a, b, side="left")
    from_long = np.ones(n_long + n_short, dtype=bool)
    from_long[where_short] = False

    def weave(u, v):
        u, v = np.asarray(u), np.asarray(v)
        out = np.empty((n_long + n_short,) + u.shape[1:], dtype=np.result_type(u, v))
        out[from_long] = u
        out[where_short] = v
        return out

    return (weave(a, b),) + tuple(weave(i, j) for i, j in args)


_MATERN52_CHECKED = {}      # the same key -> the scaled unit form reproduced the kernel's own get_sde() (checked once per process)
_MATERN52_UNITS = {}        # (class, balancing sweeps, config sweeps) -> (Pinf, H, N, N^2 / 2) of the Matern-5/2 SDE at lam = 1, s2 = 1


class _StructureChanged(Exception):
    """_grad_rows_composite: the block partition of the drift is not the same at the perturbed parameters."""


def _public_evaluation(fn):
    """Marks the model's public evaluations (predict_f, maximum_log_likelihood_objective, log_likelihood_and_grad,
    log_likelihood_batch).  `_device_series()` counts THESE -- the outermost one of a call chain -- not its own calls: one
    evaluation may ask for the series several times on its way through the gradient methods and must get the same answer
    each time (round 4 counted calls: the first gradient of a Matern model got None for its fused adjoint, a series on
    the next internal call, and went down the dual-number road it had not chosen)."""
    import functools

    @functools.wraps(fn)
    def wrapper(self, *args, **kwargs):
        depth = getattr(self, "_eval_depth", 0)
        if depth == 0:
            self._n_eval = getattr(self, "_n_eval", 0) + 1
        self._eval_depth = depth + 1
        try:
            return fn(self, *args, **kwargs)
        finally:
            self._eval_depth = depth
    return wrapper


class StateSpaceGP:
    def __init__(self, data, kernel, noise_variance=1.0, parallel=False, max_parallel=10000):
        self.noise_variance = float(noise_variance)
        dtype = config.default_float()
        ts, ys = data
        ts = np.asarray(ts, dtype=dtype)
        ys = np.asarray(ys, dtype=dtype)
        if ts.ndim == 1:
            ts = ts[:, None]
        if ys.ndim == 1:
            ys = ys[:, None]
        if ys.shape[1] != 1:
            raise ValueError("only single-output observations are supported (pssgp/model.py:72)")
        self.kernel = kernel
        self._data = ts, ys
        self.num_latent_gps = ys.shape[-1]
        self.parallel = bool(parallel)
        self.max_parallel = max_parallel
        if not parallel:
            self._kf = lambda ssm, y: kf(ssm, y, return_loglikelihood=True, return_predicted=False)
            self._kfs = kfs
        else:
            self._kf = lambda ssm, y: pkf(ssm, y, return_loglikelihood=True, max_parallel=ts.shape[0])
            self._kfs = lambda ssm, y: pkfs(ssm, y, max_parallel=max_parallel)

    @property
    def data(self):
        return self._data

    @data.setter
    def data(self, value):
        """Assigning new data (the GPflow idiom `model.data = (X, Y)` the reference inherits) drops the device copy of
        the series and every memoised result; an IN-PLACE edit of the arrays is caught by the checksum of
        _device_series()."""
        ts, ys = value
        dtype = config.default_float()
        ts, ys = np.asarray(ts, dtype=dtype), np.asarray(ys, dtype=dtype)
        if ts.ndim == 1:
            ts = ts[:, None]
        if ys.ndim == 1:
            ys = ys[:, None]
        if ys.shape[1] != 1:
            raise ValueError("only single-output observations are supported (pssgp/model.py:72)")
        self._data = ts, ys
        self.invalidate_device_series()

    @staticmethod
    def _data_stamp(ts, ys):
        """Identity and shape of the arrays a device copy was made from, plus the bytes of their last entries
        (O(1) per evaluation: a whole-array checksum would cost more than a short series' likelihood).  An in-place edit
        that misses every sampled entry is not seen: call invalidate_device_series() after such an edit."""
        return (id(ts), id(ys), ts.shape, ys.shape, ts[-1:].tobytes(), ys[-1:].tobytes())

    def _param_key(self):
        """The kernel's hyper-parameters as a tuple (the memo key of _device_forms: an evaluation repeated at the same
        setting -- predict_f after the objective, a benchmark loop -- does not rebuild the SDE): the float value of
        every variance / lengthscales / period leaf WHATEVER its type (int, numpy scalar, 0-d array), and what else
        shapes the SDE (order, balancing sweeps, the class of every node).  The memo keeps a reference to the kernel it
        was made for and is compared by identity (`is`), so a new kernel object never finds an old one's forms."""
        quick = getattr(self, "_key_memo", None)
        tree = _tree_ids(self.kernel)   # (a part swapped in place -- `kernel.kernels[0] = ...` -- assigns no attribute)
        if (quick is not None and quick[0] == Kernel._version and quick[1] is self.kernel
                and quick[2] == config.NUMBER_OF_BALANCING_STEPS and quick[4] == tree):
            return quick[3]             # no kernel attribute has been assigned since the key was built
        version = Kernel._version
        struct = getattr(self, "_struct_memo", None)
        if struct is None or struct[0] != Kernel._struct_version or struct[1] is not self.kernel or struct[4] != tree:
            def shape_of(k):
                sub = tuple(shape_of(x) for x in getattr(k, "kernels", ()))
                base = getattr(k, "base_kernel", None)
                return (type(k).__name__, getattr(k, "_order", None), getattr(k, "_balancing_iter", None), sub,
                        None if base is None else type(base).__name__)

            struct = self._struct_memo = (Kernel._struct_version, self.kernel, leaf_parameters(self.kernel), shape_of(self.kernel), tree)
        vals = tuple([float(o.__dict__[n]) for o, n in struct[2]])
        key = (struct[3], config.NUMBER_OF_BALANCING_STEPS) + vals
        self._key_memo = (version, self.kernel, config.NUMBER_OF_BALANCING_STEPS, key, tree)
        return key

    def _device_series(self, force=False):
        """The training series resident on the device (pgps_series_*, fp64 fused path): created at the SECOND evaluation of
        the model (or with force=True) and kept for its life -- an optimiser or sampler loop then sends only the model's
        scalars per evaluation.  The first evaluation goes through the host-array entry points, whose staging buffers belong
        to the context: a model that is built, evaluated once and dropped -- the reference's speed protocol,
        experiments/toy_models/speed_and_stability.py:73-87 -- would otherwise pay a set of device and pinned-host
        allocations per call (measured: Matern-3/2, N = K = 4096, model + predict_f 470 us against 175 us).  None when the
        data are not sorted or not float64 (the host entry points take those), and at that first evaluation."""
        ser = getattr(self, "_series", None)
        ts, ys = self.data
        if ser is not None and getattr(self, "_series_stamp", None) != self._data_stamp(ts, ys):
            self.invalidate_device_series()         # the arrays were replaced or edited since the copy was made
            ser = None
        if ser is None and not force and getattr(self, "_n_eval", 0) <= 1:
            return None                             # still inside the model's first evaluation (however many internal calls it makes)
        if ser is None:
            from . import _backend
            t = ts.reshape(-1)
            self._series_stamp = self._data_stamp(ts, ys)
            if ts.dtype != np.float64 or t.size < 1 or not np.all(np.diff(t) >= 0):
                self._series = False
                return None
            try:
                self._series = _backend.Series(t, ys.reshape(-1))
            except RuntimeError:
                self._series = False
            ser = self._series
        return ser or None

    def invalidate_device_series(self):
        """The device copy of the series is rebuilt at the next evaluation and memoised likelihoods are dropped (done
        automatically when `data` is assigned or found edited)."""
        ser = getattr(self, "_series", None)
        if ser:
            ser.close()
        self._series = None
        self._series_stamp = None
        self._ll_memo = None

    def _device_forms(self):
        key = self._param_key()
        memo = getattr(self, "_forms_memo", None)
        if memo is not None and memo[2] is self.kernel and memo[0] == key:
            return memo[1]
        out = self._device_forms_uncached()
        self._forms_memo = (key, out, self.kernel)
        return out

    def _packed_fused(self, fused):
        """The fused model as the contiguous arrays the series calls take, memoised with the forms."""
        memo = getattr(self, "_packed_memo", None)
        if memo is not None and memo[0] is fused:
            return memo[1]
        from . import _backend
        sde, form = fused
        d = form[1].shape[0]
        pb = getattr(self, "_pack_buffer", None)
        if pb is None or pb.d != d:
            pb = self._pack_buffer = _backend.PackBuffer(d)
        packed = pb.pack(form, sde.P0, sde.H)       # (views into the model's own buffer: valid until the next setting)
        self._packed_memo = (fused, packed)
        return packed

    def _matern_forms(self):
        """(sde-like, (lam, N1, N2)) of a SINGLE Matern-1/2, -3/2 or -5/2 kernel without building its SDE: what a new
        hyper-parameter setting costs on the host in an optimiser loop (get_sde + stationarity check + nilpotent form:
        ~55 us for Matern-3/2, ~130 us for Matern-5/2 with its balancing sweep and Lyapunov solve) becomes a few scalar
        operations.  With lam = sqrt(2 nu) / l:
          Matern-1/2:  F = -lam, Pinf = s2, H = 1;
          Matern-3/2:  the companion form itself (matern32.py:10-28): N = [[lam, 1], [-lam^2, -lam]], N^2 = 0,
                       Pinf = diag(s2, lam^2 s2), H = (1, 0);
          Matern-5/2:  balanced (matern52.py): the balancing iteration commutes with the lengthscale's time scaling, so
                       F = lam F1, Pinf = s2 P1, H = H1 with (F1, P1, H1) the SDE at lam = 1, s2 = 1 -- built ONCE per model
                       by get_sde() and checked against get_sde() at the first setting it is used for (1e-10); a kernel
                       for which that check fails keeps the general path.
        None for anything else."""
        k = self.kernel
        name = type(k).__name__
        if name not in ("Matern12", "Matern32", "Matern52") or getattr(k, "kernels", None):
            return None
        if getattr(self, "_matern_fast", None) is False:
            return None
        s2, ell = float(k.variance), float(k.lengthscales)
        if not (s2 > 0.0 and ell > 0.0):
            return None
        from types import SimpleNamespace
        if name == "Matern12":
            lam = 1.0 / ell
            return (SimpleNamespace(P0=np.array([[s2]]), H=np.array([[1.0]])), (lam, np.zeros((1, 1)), np.zeros((1, 1))))
        if name == "Matern32":
            lam = np.sqrt(3.0) / ell
            return (SimpleNamespace(P0=np.array([[s2, 0.0], [0.0, lam * lam * s2]]), H=np.array([[1.0, 0.0]])),
                    (lam, np.array([[lam, 1.0], [-lam * lam, -lam]]), np.zeros((2, 2))))
        unit = getattr(self, "_matern52_unit", None)
        if unit is None:
            # (shared by every model of the process: a model built per evaluation -- the reference's speed protocol -- would
            # otherwise balance and solve the unit SDE once per call)
            ukey = (type(k), getattr(k, "_balancing_iter", None), config.NUMBER_OF_BALANCING_STEPS)
            unit = _MATERN52_UNITS.get(ukey)
        if unit is None:
            from . import _backend
            sde1 = type(k)(1.0, np.sqrt(5.0), **({"balancing_iter": k._balancing_iter} if hasattr(k, "_balancing_iter") else {})).get_sde()
            form1 = _backend.nilpotent_form(sde1.F)
            if form1 is None or abs(form1[0] - 1.0) > 1e-12:
                self._matern_fast = False
                return None
            unit = _MATERN52_UNITS[ukey] = (np.asarray(sde1.P0, np.float64), np.asarray(sde1.H, np.float64).reshape(1, -1),
                                            np.asarray(form1[1], np.float64), np.asarray(form1[2], np.float64))
        self._matern52_unit = unit
        lam = np.sqrt(5.0) / ell
        out = (SimpleNamespace(P0=s2 * unit[0], H=unit[1]), (lam, lam * unit[2], (lam * lam) * unit[3]))
        if getattr(self, "_matern_fast", None) is None:
            self._matern_fast = _MATERN52_CHECKED.get((type(k), getattr(k, "_balancing_iter", None), config.NUMBER_OF_BALANCING_STEPS))
        if getattr(self, "_matern_fast", None) is None:            # first use in the process: against the kernel's own get_sde()
            from . import _backend
            sde = k.get_sde()
            form = _backend.nilpotent_form(sde.F)
            close = lambda a, b: np.max(np.abs(np.asarray(a) - np.asarray(b))) <= 1e-10 * (1.0 + np.max(np.abs(np.asarray(b))))
            self._matern_fast = bool(form is not None and close(out[1][0], form[0]) and close(out[1][1], form[1])
                                     and close(out[1][2], form[2]) and close(out[0].P0, sde.P0)
                                     and close(out[0].H, np.asarray(sde.H).reshape(1, -1)))
            _MATERN52_CHECKED[(type(k), getattr(k, "_balancing_iter", None), config.NUMBER_OF_BALANCING_STEPS)] = self._matern_fast
            if not self._matern_fast:
                return None
        return out

    def _rbf_scaled_sde(self):
        """A single RBF kernel near a setting whose SDE is already built: the lengthscale is a scaling of time and the
        variance a scaling of Pinf, so the model at (s2, l) is the reference one at (s2_0, l_0) with F l_0 / l and
        Pinf s2 / s2_0 -- the same likelihood as get_sde()'s realisation gives, to rounding (they differ by a diagonal
        similarity where the balancing sweeps have not converged; tests/test_gpu_lti.py) -- without the 0.2 - 0.4 ms of
        get_sde() (polynomial roots, balancing, Lyapunov solve at d = 6) per step of an optimiser.  The reference setting
        is renewed whenever the lengthscale has drifted by more than 25 % from it.  None for other kernels."""
        from .kernels import RBF
        k = self.kernel
        if not isinstance(k, RBF) or getattr(k, "kernels", None):
            return None
        s2, ell = float(k.variance), float(k.lengthscales)
        if not (s2 > 0.0 and ell > 0.0):
            return None
        from . import _backend
        ref = getattr(self, "_rbf_ref", None)
        tag = (getattr(k, "_order", None), getattr(k, "_balancing_iter", None))
        if ref is not None and (ref[6] is not k or ref[5] != tag):
            ref = None                              # another kernel object (or order / balancing) than the reference's
        if ref is None or not (0.8 <= ell / ref[1] <= 1.25):
            sde = k.get_sde()
            F, P0 = np.asarray(sde.F, np.float64), np.asarray(sde.P0, np.float64)
            if not (_backend.LTI_DIM_MIN <= F.shape[0] <= _backend.LTI_DIM_MAX):
                return None
            L = np.asarray(sde.L, np.float64)
            LQL = L @ np.atleast_2d(np.asarray(sde.Q, np.float64)) @ L.T
            if np.max(np.abs(F @ P0 + P0 @ F.T + LQL)) > 1e-8 * max(1.0, float(np.max(np.abs(LQL)))):
                return None
            ref = self._rbf_ref = (s2, ell, F, P0, np.asarray(sde.H, np.float64), tag, k)
        from types import SimpleNamespace
        return SimpleNamespace(F=ref[2] * (ref[1] / ell), P0=ref[3] * (s2 / ref[0]), H=ref[4])

    def _device_forms_uncached(self):
        """(fused, lti) from ONE get_sde() (for composite kernels that call is the host cost of an evaluation):
        `fused` = (sde, (lam, N1, N2)) when the SDE has the closed-form discretisation of the fused HIP path
        (F = -lam I + N, N nilpotent, d <= 3: the Matern family), `lti` = the SDE when the general-LTI device path applies
        (state dimension 2..32); both need parallel=True and a stationary P0 -- the GPU discretisation forms
        Q = P0 - F_k P0 F_k^T.  The general-LTI path computes in fp64; a float32 model hands over its times and
        observations widened (N scalars each) and gets results rounded to float32 -- less traffic and better
        arithmetic than fp32 (N, d, d) arrays."""
        if not self.parallel:
            return None, None
        fast = self._matern_forms()
        if fast is not None:
            return fast, None
        from . import _backend
        scaled = self._rbf_scaled_sde()
        if scaled is not None:
            return None, scaled
        sde = self.kernel.get_sde()
        F, P0 = np.asarray(sde.F, np.float64), np.asarray(sde.P0, np.float64)
        L = np.asarray(sde.L, np.float64)
        LQL = L @ np.atleast_2d(np.asarray(sde.Q, np.float64)) @ L.T
        if np.max(np.abs(F @ P0 + P0 @ F.T + LQL)) > 1e-8 * max(1.0, float(np.max(np.abs(LQL)))):
            return None, None
        form = _backend.nilpotent_form(sde.F)
        if form is not None:
            return (sde, form), None
        if _backend.LTI_DIM_MIN <= F.shape[0] <= _backend.LTI_DIM_MAX:
            return None, sde
        return None, None

    def _fused_form(self):
        return self._device_forms()[0]

    def _lti_form(self):
        return self._device_forms()[1]

    def _make_model(self, ts):
        R = np.reshape(np.asarray(self.noise_variance, dtype=config.default_float()), (1, 1))
        return self.kernel.get_ssm(ts, R)

    @_public_evaluation
    def predict_f(self, Xnew, full_cov=False, full_output_cov=False):
        """Posterior mean (K, 1) and variance (K, 1) at `Xnew` (pssgp/model.py:92-111):
        merge train and query times, mark queries as missing, smooth, keep the query rows,
        project through H."""
        ts, ys = self.data
        dtype = config.default_float()
        Xnew = np.asarray(Xnew, dtype=dtype)
        squeezed_ts = ts.reshape(-1)
        squeezed_Xnew = Xnew.reshape(-1)
        fused, lti = self._device_forms()
        if fused is not None and squeezed_Xnew.size > 0 and dtype == np.float64 and np.all(np.diff(squeezed_Xnew) >= 0):
            ser = self._device_series()
            if ser is not None:
                # series and query grid resident on the device (merged once): the call sends the model, K means and
                # variances come back
                ser.set_queries(squeezed_Xnew)
                mean, var, ll = ser.gp_predict(self._packed_fused(fused), self.noise_variance)
                # (the filter pass of the prediction IS the training log-likelihood -- the query rows are missing
                # observations: an objective evaluated next at the same setting costs nothing)
                self._ll_memo = (self._param_key(), float(self.noise_variance), ser, np.float64(ll), self.kernel)
                return mean[:, None], var[:, None]
        if (fused is not None and squeezed_Xnew.size > 0 and np.all(np.diff(squeezed_ts) >= 0)
                and np.all(np.diff(squeezed_Xnew) >= 0)):
            # the whole of predict_f on the device: merge, missing-marking, filter + smoother, projection
            # through H at the query rows -- K means and variances come back, nothing else
            from . import _backend
            sde, form = fused
            mean, var, _ = _backend.gp_predict(form, sde.P0, sde.H, self.noise_variance, squeezed_ts, ys.reshape(-1),
                                               squeezed_Xnew)
            return mean[:, None], var[:, None]
        if (lti is not None and squeezed_Xnew.size > 0 and np.all(np.diff(squeezed_ts) >= 0)
                and np.all(np.diff(squeezed_Xnew) >= 0)):
            # kernels without the closed-form discretisation (RBF, Periodic, sums, products): merge, discretisation,
            # filter + smoother and projection on the device as well -- on the resident series where there is one
            from . import _backend
            ser = self._device_series() if squeezed_ts.dtype == np.float64 else None
            if ser is not None and ser.has_lti:
                ser.set_queries(squeezed_Xnew)
                mean, var, ll = ser.lti_predict(lti.F, lti.P0, lti.H, self.noise_variance)
                self._ll_memo = (self._param_key(), float(self.noise_variance), ser, config.default_float()(ll), self.kernel)
                return mean[:, None].astype(dtype), var[:, None].astype(dtype)
            mean, var, _ = _backend.lti_predict(lti.F, lti.P0, lti.H, self.noise_variance, squeezed_ts, ys.reshape(-1),
                                                squeezed_Xnew)
            return mean[:, None].astype(dtype), var[:, None].astype(dtype)
        nan_ys = np.full((squeezed_Xnew.shape[0], ys.shape[1]), np.nan, dtype=ys.dtype)
        all_ts, all_ys, all_flags = _merge_sorted(
            squeezed_ts, squeezed_Xnew, (ys, nan_ys),
            (np.zeros(squeezed_ts.shape, dtype=bool), np.ones(squeezed_Xnew.shape, dtype=bool)))
        if fused is not None:
            # times and observations straight into the scan kernels (Fs / Qs never materialised)
            from . import _backend
            sde, form = fused
            res = _backend.gp(form, sde.P0, sde.H, self.noise_variance, all_ts.astype(dtype), all_ys.reshape(-1),
                              want_smoothed=True)
            sms, sPs = res["sms"], res["sPs"]
            H = np.asarray(sde.H, dtype=dtype).reshape(1, -1)
        else:
            ssm = self._make_model(all_ts[:, None])
            sms, sPs = self._kfs(ssm, all_ys)
            H = np.asarray(ssm.H)
        sm, sP = sms[all_flags], sPs[all_flags]
        mean = sm @ H.T
        var = np.einsum("ai,nij,aj->na", H, sP, H)
        return mean, var

    @_public_evaluation
    def maximum_log_likelihood_objective(self):
        ts, Y = self.data
        fused, lti = self._device_forms()
        if fused is not None:
            from . import _backend
            ser = self._device_series() if ts.dtype == np.float64 else None
            if ser is not None:
                memo = getattr(self, "_ll_memo", None)
                if (memo is not None and memo[2] is ser and memo[4] is self.kernel and memo[1] == float(self.noise_variance)
                        and memo[0] == self._param_key()):
                    return memo[3]
                return np.float64(ser.gp_ll(self._packed_fused(fused), self.noise_variance))
            sde, form = fused
            return _backend.gp(form, sde.P0, sde.H, self.noise_variance, ts.reshape(-1), Y.reshape(-1))["ll"]
        if lti is not None:
            from . import _backend
            ser = self._device_series() if ts.dtype == np.float64 else None
            if ser is not None and ser.has_lti:
                memo = getattr(self, "_ll_memo", None)
                if (memo is not None and memo[2] is ser and memo[4] is self.kernel and memo[1] == float(self.noise_variance)
                        and memo[0] == self._param_key()):
                    return memo[3]
                return config.default_float()(ser.lti_ll(lti.F, lti.P0, lti.H, self.noise_variance))
            ll = _backend.lti_ll(lti.F, lti.P0, lti.H, self.noise_variance, ts.reshape(-1), Y.reshape(-1))
            return config.default_float()(ll)
        ssm = self._make_model(ts)
        _, _, ll = self._kf(ssm, Y)
        return ll

    # -- hyper-parameter gradients --------------------------------
    def trainable_parameters(self):
        """[(owner, attribute name)] in the order the gradient is returned: every leaf kernel's variance,
        lengthscale and period (the reference's gpflow Parameters: the Matern / RBF kernels' variance /
        lengthscales, Periodic's period and its base kernel's parameters, the leaves of sums and products in order),
        then the observation-noise variance (pssgp/model.py:68)."""
        self._param_key()                           # (validates the memoised structure of the kernel)
        return list(self._struct_memo[2]) + [(self, "noise_variance")]

    def _grad_blocks_matern(self):
        """_grad_blocks() of a single Matern-1/2, -3/2 or -5/2 kernel in closed form, from the memoised get_sde() of the
        evaluation -- no further SDE construction (four of them per parameter were what a small-N gradient call cost on
        the host).  With lam = sqrt(2 nu) / l and F = -lam I + N:
          variance s2:  Pinf is linear in it, nothing else moves  ->  dPinf = Pinf / s2;
          lengthscale:  dlam = -lam / l.  Matern-3/2 keeps the companion form (matern32.py:10-28: N = [[lam, 1],
            [-lam^2, -lam]], Pinf = diag(s2, lam^2 s2)); Matern-5/2 is balanced (matern52.py: balance_ss + Lyapunov), and
            Osborne's balancing iteration commutes with the time scaling x_i -> lam^i x_i that relates the companion
            forms at two lengthscales, so the balanced drift is lam * (a constant matrix) and Pinf, H do not move:
            dN = N / lam * dlam, dPinf = 0, dH = 0  (checked against the differences: tests/test_model_host.py);
          noise:        R alone.
        None for any other kernel."""
        k = self.kernel
        name = type(k).__name__
        if name not in ("Matern12", "Matern32", "Matern52") or getattr(k, "kernels", None) or not k.variance:
            return None
        fused = self._fused_form() if self.parallel else None
        if fused is None:
            return None
        sde, form = fused
        lam, N = np.float64(form[0]), np.asarray(form[1], np.float64)
        Pinf, H = np.asarray(sde.P0, np.float64), np.asarray(sde.H, np.float64).reshape(-1)
        zN, zP, zH, z = np.zeros_like(N), np.zeros_like(Pinf), np.zeros_like(H), np.float64(0.0)
        dlam = -lam / k.lengthscales
        if name == "Matern32":
            dN = dlam * np.array([[1.0, 0.0], [-2.0 * lam, -1.0]])
            dP = np.diag([0.0, 2.0 * lam * k.variance * dlam])
        else:
            dN, dP = N * (dlam / lam), zP
        by_name = {"variance": (z, zN, Pinf / k.variance, zH, z), "lengthscales": (dlam, dN, dP, zH, z),
                   "noise_variance": (z, zN, zP, zH, np.float64(1.0))}
        return [(lam, N, Pinf, H, np.float64(self.noise_variance))] + [by_name[n] for _, n in self.trainable_parameters()]

    def _grad_blocks(self, closed_form=True):
        """The fused model (lam, N, Pinf, H, R) and its partial derivatives with respect to each
        trainable parameter: closed form for a single Matern kernel; otherwise the SDE coefficients are low-degree
        rational functions of the parameters and a Richardson-extrapolated central difference of get_sde() is exact
        to ~1e-11."""
        closed = self._grad_blocks_matern() if closed_form else None
        if closed is not None:
            return closed

        def block():
            sde = self.kernel.get_sde()
            from . import _backend
            form = _backend.nilpotent_form(sde.F)
            if form is None:
                raise NotImplementedError("gradients need the closed-form (Matern-family) discretisation")
            return [np.float64(form[0]), np.asarray(form[1], np.float64), np.asarray(sde.P0, np.float64),
                    np.asarray(sde.H, np.float64).reshape(-1), np.float64(self.noise_variance)]

        base = block()
        blocks = [tuple(base)]
        zeros = [np.zeros_like(np.asarray(v, np.float64)) for v in base]
        for owner, name in self.trainable_parameters():
            x0 = getattr(owner, name)
            # two of the three derivatives are known in closed form -- and get_sde() (balancing sweep + Lyapunov solve)
            # is what a small-N gradient call costs on the host: the observation noise enters through R alone, and a
            # single Matern kernel's variance through Pinf alone, linearly (balance_ss leaves F, H independent of q)
            if owner is self and name == "noise_variance":
                blocks.append(tuple(zeros[:4]) + (np.float64(1.0),))
                continue
            if owner is self.kernel and name == "variance" and x0 != 0.0:
                blocks.append((zeros[0], zeros[1], base[2] / x0, zeros[3], np.float64(0.0)))
                continue

            def central(h):
                setattr(owner, name, x0 + h)
                up = block()
                setattr(owner, name, x0 - h)
                dn = block()
                return [(u - v) / (2.0 * h) for u, v in zip(up, dn)]

            try:
                h = 1e-3 * max(abs(x0), 1e-3)
                d1, d2 = central(h), central(0.5 * h)
                blocks.append(tuple((4.0 * b - a) / 3.0 for a, b in zip(d1, d2)))
            finally:
                setattr(owner, name, x0)
        return blocks

    def _grad_rows_composite(self):
        """(rows, block sizes) for pgps_gp_ll_grad_blocks_*: the block-nilpotent model (lam_b, N, Pinf, H, R) of a sum /
        product of Matern kernels and its partial derivatives with respect to each trainable parameter (Richardson
        central differences of get_sde(): its entries are low-degree rational functions of the parameters, exact to
        ~1e-11), or (None, None) when the kernel's drift does not have that form, the state dimension is not 2..6, or
        the partition into blocks changes under the perturbation."""
        from . import _backend

        def row():
            sde = self.kernel.get_sde()
            F = np.asarray(sde.F, np.float64)
            blocks = _backend.nilpotent_blocks(F)
            if blocks is None:
                return None, None
            Nm = F.copy()
            for lo, n, lam, _ in blocks:
                Nm[lo:lo + n, lo:lo + n] += lam * np.eye(n)
            return ([np.array([b[2] for b in blocks]), Nm, np.asarray(sde.P0, np.float64),
                     np.asarray(sde.H, np.float64).reshape(-1), np.float64(self.noise_variance)],
                    [b[1] for b in blocks])

        base, sizes = row()
        d = None if base is None else base[1].shape[0]
        if base is None or not (2 <= d <= _backend.GRAD_BLOCKS_DIM_MAX) or len(sizes) > _backend.GRAD_BLOCKS_MAX:
            self._no_composite = True          # (a property of the kernel's structure: not asked again)
            return None, None
        rows = [tuple(base)]
        for owner, name in self.trainable_parameters():
            x0 = getattr(owner, name)

            def central(h):
                setattr(owner, name, x0 + h)
                up, su = row()
                setattr(owner, name, x0 - h)
                dn, sd = row()
                if up is None or dn is None or su != sizes or sd != sizes:
                    # (a coupling entry near nilpotent_blocks' threshold: the partition found at x0 +- h differs)
                    raise _StructureChanged()
                return [(u - v) / (2.0 * h) for u, v in zip(up, dn)]

            try:
                h = 1e-3 * max(abs(x0), 1e-3)
                d1, d2 = central(h), central(0.5 * h)
                rows.append(tuple((4.0 * b - a) / 3.0 for a, b in zip(d1, d2)))
            except _StructureChanged:
                return None, None               # the caller differentiates the likelihood itself instead
            finally:
                setattr(owner, name, x0)
        return rows, sizes

    # Matern-5/2 (d = 3, three directions): above the one-launch length the dual-number pass is three launches of kernels
    # three times as heavy as the filter's; the adjoint pass on the general-LTI kernels costs 1.5 likelihoods.  Measured
    # (tools/grad_methods.py, one MI355X): N = 4096 199 -> 117 us, 32 768 186 -> 151 us, 131 072 320 -> 238 us; a tie at
    # 1000; Matern-3/2 (d = 2) is better off on duals at every length (79 vs 102 us at 4096).
    _MATERN52_ADJOINT_FROM = 2048

    def _matern_prepared(self, fused):
        """(F, Pinf, H, derivatives) of a single Matern kernel from its fused form, for the adjoint pass: the lengthscale
        scales time (dF = -F / l) and the variance Pinf, as for RBF below -- no get_sde() per optimiser step."""
        sde_like, (lam, N1, _) = fused
        k = self.kernel
        d = N1.shape[0]
        F = np.ascontiguousarray(np.asarray(N1, np.float64) - lam * np.eye(d))
        P0 = np.ascontiguousarray(sde_like.P0, np.float64)
        H = np.ascontiguousarray(np.asarray(sde_like.H, np.float64).reshape(-1))
        zF, zP, zH = np.zeros_like(F), np.zeros_like(P0), np.zeros((1, H.size))
        by_name = {"variance": (zF, P0 / float(k.variance), zH), "lengthscales": (-F / float(k.lengthscales), zP, zH)}
        return (F, P0, H, [by_name[a] for _, a in leaf_parameters(k)])

    def _fused_adjoint_pays(self, n):
        """Automatic choice between the dual-number pass and the adjoint pass of the fused path, from measurements on one
        MI355X (tools/grad_methods.py, microseconds per call, dual / adjoint): Matern-5/2 96 / 78 at N = 200, 198 / 102 at
        4096, 592 / 147 at 2^20: always the adjoint; Matern-3/2 51 / 54 at 1000 (one launch each), 77 / 64 at 4096, 186 / 92
        at 2^20: above the one-launch length; Matern-1/2 (two directions, d = 1) 46 / 53: duals."""
        name = type(self.kernel).__name__
        return name == "Matern52" or (name == "Matern32" and n > 2048)

    def _fused_adjoint_ll_and_grad(self, fused, wrt=None):
        """(ll, grad) of a single Matern kernel by the adjoint pass on the fused (closed-form discretisation) kernels
        (csrc/pgps_gpadj.hip.h): the device returns the model's adjoints [Abar | Ubar | Hbar | Rbar] from one filter pass
        and one reverse pass over the resident series; the lengthscale scales time (dF = -F / l) and the variance Pinf
        (dPinf = Pinf / s2), so  d ll / d l = -<Abar, F> / l,  d ll / d s2 = Ubar^T Pinf H^T / s2,  d ll / d R = Rbar.
        None when the series is not resident or the library has no such entry point."""
        ser = self._device_series()
        if ser is None or not getattr(ser, "has_gp_adj", False):
            return None
        packed = self._packed_fused(fused)
        lam, N1, _, Pinf, H, d = packed
        memo = getattr(self, "_fadj_memo", None)
        if memo is None or memo[0] is not packed:
            F = N1.reshape(-1).copy()
            F[::d + 1] -= lam                       # F = N1 - lam I
            self._param_key()
            memo = self._fadj_memo = (packed, F, Pinf @ H, [a for _, a in self._struct_memo[2]])
        _, F, PH, names = memo
        out = ser.gp_ll_grad_adj_raw(packed, self.noise_variance)
        dd = d * d
        k = self.kernel
        by_name = {"variance": float(out[1 + dd:1 + dd + d] @ PH) / float(k.variance),
                   "lengthscales": -float(out[1:1 + dd] @ F) / float(k.lengthscales)}
        g = np.array([by_name[a] for a in names] + [float(out[1 + dd + 2 * d])])
        if wrt is not None:
            keep = np.zeros(len(g), bool)
            keep[[int(i) for i in wrt]] = True
            g = np.where(keep, g, 0.0)
        return config.default_float()(out[0]), g

    def _adjoint_ll_and_grad(self, wrt=None, prepared=None):
        """(ll, grad) by the adjoint pass of the general-LTI device path (pgps_lti_ll_grad_f64): the device returns the
        adjoints of the model (F, Pinf, H, R) from one filter pass and one reverse pass, pssgp.kernels.sde_grads the
        model's derivatives in a frozen state basis; exact (no differences), ONE device call, any number of parameters.
        None when the kernel has no derivative rule, a derivative of the drift does not commute with it (the contraction
        the device makes would not apply), or the library lacks the entry point."""
        from . import _backend
        from .kernels.sde_grads import sde_with_grads
        key = self._param_key()
        memo = getattr(self, "_grads_memo", None)
        if prepared is not None:
            pass                                # the caller wrote the model and its derivatives down (Matern family)
        elif memo is not None and memo[2] is self.kernel and memo[0] == key:
            prepared = memo[1]
        else:
            prepared = None
            scaled = self._rbf_scaled_sde()
            if scaled is not None:
                # a single RBF kernel near a setting whose SDE is built (the steps of an optimiser or sampler): the
                # lengthscale is a scaling of time and the variance one of Pinf, so the model AND its derivatives are
                # written down from the reference realisation -- no polynomial roots, balancing or Lyapunov solve per step
                k = self.kernel
                F = np.ascontiguousarray(scaled.F, np.float64)
                P0 = np.ascontiguousarray(scaled.P0, np.float64)
                H = np.ascontiguousarray(np.asarray(scaled.H, np.float64).reshape(-1))
                zF, zP, zH = np.zeros_like(F), np.zeros_like(P0), np.zeros((1, H.size))
                by_name = {"variance": (zF, P0 / float(k.variance), zH), "lengthscales": (-F / float(k.lengthscales), zP, zH)}
                prepared = (F, P0, H, [by_name[a] for _, a in leaf_parameters(k)])
                sde = None
            else:
                try:
                    sde, grads = sde_with_grads(self.kernel)
                except (NotImplementedError, ZeroDivisionError, FloatingPointError):
                    sde = None
            if sde is not None:
                F, P0 = np.ascontiguousarray(sde.F, np.float64), np.ascontiguousarray(sde.P0, np.float64)
                H = np.ascontiguousarray(np.asarray(sde.H, np.float64).reshape(-1))
                L = np.asarray(sde.L, np.float64)
                LQL = L @ np.atleast_2d(np.asarray(sde.Q, np.float64)) @ L.T
                fmax = max(1.0, float(np.max(np.abs(F))))
                ok = (_backend.LTI_DIM_MIN <= F.shape[0] <= _backend.LTI_DIM_MAX and np.all(np.isfinite(F)) and np.all(np.isfinite(P0))
                      and np.max(np.abs(F @ P0 + P0 @ F.T + LQL)) <= 1e-8 * max(1.0, float(np.max(np.abs(LQL)))))
                for dF, dP, dH in grads if ok else ():
                    if not np.all(np.isfinite(dP)):
                        ok = False
                    if not dF.any():
                        continue                    # (most parameters move Pinf or H only: nothing to commute)
                    if not np.all(np.isfinite(dF)) or \
                            np.max(np.abs(F @ dF - dF @ F)) > 1e-9 * fmax * max(1.0, float(np.max(np.abs(dF)))):
                        ok = False
                if ok:
                    prepared = (F, P0, H, grads)
            self._grads_memo = (key, prepared, self.kernel)
        if prepared is None:
            return None
        F, P0, H, grads = prepared
        ts, Y = self.data
        ser = self._device_series() if ts.dtype == np.float64 else None
        try:
            if ser is not None and getattr(ser, "has_lti_grad", False):
                stats = ser.lti_ll_grad(F, P0, H, self.noise_variance)
            else:
                stats = _backend.lti_ll_grad(F, P0, H, self.noise_variance, ts.reshape(-1), Y.reshape(-1))
        except _backend.PgpsError as e:
            if getattr(e, "code", None) == _backend.E_UNSUPPORTED_DIM:
                return None
            raise
        except RuntimeError:
            return None
        # contraction <Abar, dF> + Ubar^T dPinf H^T + Hbar . dH per parameter, the identically-zero components skipped (a
        # parameter moves one of F, Pinf, H: one dot product each instead of three products on zeros -- `_backend.
        # contract_grad_stats` is the plain form the tests check this against)
        fast = getattr(self, "_contract_memo", None)
        if fast is None or fast[0] is not grads:
            h = H.reshape(-1)
            rows = [(dF.reshape(-1) if np.any(dF) else None, (np.asarray(dP) @ h) if np.any(dP) else None,
                     np.asarray(dH).reshape(-1) if np.any(dH) else None) for dF, dP, dH in grads]
            fast = self._contract_memo = (grads, rows)
        _, Abar, Ubar, Hbar, Rbar = stats
        Av = Abar.reshape(-1)
        g = np.array([(0.0 if a is None else float(Av @ a)) + (0.0 if u is None else float(Ubar @ u))
                      + (0.0 if hh is None else float(Hbar @ hh)) for a, u, hh in fast[1]] + [float(Rbar)])
        if wrt is not None:
            keep = np.zeros(len(g), bool)
            keep[[int(i) for i in wrt]] = True
            g = np.where(keep, g, 0.0)
        return config.default_float()(stats[0]), g

    @_public_evaluation
    def log_likelihood_and_grad(self, wrt=None, method=None):
        """(ll, grad): the marginal log-likelihood and its gradient with respect to
        `trainable_parameters()` -- what the reference obtains from tf.GradientTape over
        maximum_log_likelihood_objective (tests/test_gp_vs_kfs.py:53-78).  parallel=True.  Matern family (d <= 3):
        ONE pass of the parallel filter on dual numbers, exact (Matern-5/2 above 2048 points: the adjoint pass, which
        is cheaper there).  Every other kernel (RBF, Periodic, sums, products,
        2 <= d <= 32): the ADJOINT pass -- one filter pass and one reverse pass on the device, exact, whatever the number
        of parameters (`_adjoint_ll_and_grad`).  `method`: None = automatic, "adjoint", "dual" (sums / products of Matern
        kernels up to d = 6 on dual numbers) or "differences" (Richardson central differences of batched likelihoods:
        the cross-checks of the tests); `wrt`: only these parameter indices, the others get 0."""
        if not self.parallel:
            raise NotImplementedError("gradients run on the parallel (HIP) path: construct with parallel=True")
        if method not in (None, "adjoint", "dual", "differences"):
            raise ValueError(f"method = {method!r}: None (automatic), 'adjoint', 'dual' or 'differences'")
        from . import _backend
        ts, Y = self.data
        fused, lti = self._device_forms()
        if method == "differences":
            # asked for by name: Richardson differences of batched likelihoods whatever the kernel (one evaluation at a
            # time where no batched entry point covers the state dimension)
            return self._lti_ll_and_grad(batched=(fused is not None or lti is not None), wrt=wrt)
        if fused is None and lti is not None and method in (None, "adjoint"):
            # every kernel without the closed-form discretisation: the adjoint pass (two passes whatever the number of
            # parameters); None when the kernel has no derivative rule or the library no such entry point
            out = self._adjoint_ll_and_grad(wrt)
            if out is not None:
                return out
            if method == "adjoint":
                raise NotImplementedError("no adjoint gradient for this kernel / library")
        if fused is None:
            # sums / products of Matern kernels (block-nilpotent drift, d <= 6): exact, dual numbers through the scan
            rows, sizes = (None, None)
            if (method in (None, "dual") and lti is not None and lti.F.shape[0] <= _backend.GRAD_BLOCKS_DIM_MAX
                    and not getattr(self, "_no_composite", False)):
                rows, sizes = self._grad_rows_composite()
            if rows is not None:
                ll, g = _backend.gp_ll_grad_blocks(rows, sizes, ts.reshape(-1), Y.reshape(-1))
                if wrt is not None:             # (one pass gives every direction; the ones not asked for read 0)
                    keep = np.zeros(len(g), bool)
                    keep[[int(i) for i in wrt]] = True
                    g = np.where(keep, g, 0.0)
                return ll, g
            if method == "dual":
                raise NotImplementedError("method = 'dual': this kernel is no sum / product of Matern kernels of d <= "
                                          f"{_backend.GRAD_BLOCKS_DIM_MAX} (use 'adjoint' or 'differences')")
            if method == "adjoint":
                raise NotImplementedError("no adjoint gradient for this kernel / library")
            # no dual-number path: batched differences on the general-LTI kernels (d <= 16), one evaluation at a
            # time above that (e.g. the CO2 kernel at its reference order, d = 18)
            return self._lti_ll_and_grad(batched=lti is not None, wrt=wrt)
        if method in (None, "adjoint") and ts.dtype == np.float64 and (method == "adjoint" or self._fused_adjoint_pays(ts.shape[0])):
            # the adjoint pass on the fused path's own kernels (csrc/pgps_gpadj.hip.h)
            out = self._fused_adjoint_ll_and_grad(fused, wrt)
            if out is not None:
                return out
        if (method in (None, "adjoint") and type(self.kernel).__name__ == "Matern52" and ts.dtype == np.float64
                and ts.shape[0] > self._MATERN52_ADJOINT_FROM):
            out = self._adjoint_ll_and_grad(wrt, prepared=self._matern_prepared(fused))
            if out is not None:
                return out
        if method == "adjoint":
            # asked for by name and not available on the fused path (float32 data, a library without the entry point, a series
            # that cannot be made resident): the general adjoint pass on the same model, or an error -- never silently duals
            out = self._adjoint_ll_and_grad(wrt, prepared=self._matern_prepared(fused))
            if out is not None:
                return out
            raise NotImplementedError("method = 'adjoint' is not available for this model / library")
        ser = self._device_series() if ts.dtype == np.float64 else None
        if ser is not None:
            model, d, npar = _backend.pack_grad_model(self._grad_blocks())
            ll, g = ser.gp_ll_grad(model, d, npar)
        else:
            ll, g = _backend.gp_ll_grad(self._grad_blocks(), ts.reshape(-1), Y.reshape(-1))
        if wrt is not None:                 # (one pass gives every direction; the ones not asked for read 0)
            keep = np.zeros(len(g), bool)
            keep[[int(i) for i in wrt]] = True
            g = np.where(keep, g, 0.0)
        return ll, g

    def _lti_ll_and_grad(self, rel_step=1e-3, batched=True, wrt=None):
        """Kernels without the closed-form discretisation (RBF, Periodic, sums, products; d <= 16): the gradient by
        Richardson-extrapolated central differences -- 4 P + 1 likelihood evaluations, ALL IN ONE batched device call
        (pgps_lti_ll_batch_*), so its cost is that of one launch set, not of 4 P + 1.  Truncation error O(h^4):
        ~1e-8 relative for these smooth objectives (tests/test_gpu_lti.py checks it against the dense GP's
        gradient); the dual-number pass of the Matern family is exact."""
        params = self.trainable_parameters()
        x0 = np.array([getattr(o, n) for o, n in params], np.float64)
        rows = [x0]
        hs = rel_step * np.maximum(np.abs(x0), 1e-3)
        idx = list(range(len(params))) if wrt is None else [int(i) for i in wrt]   # `wrt`: only these parameters
        for i in idx:
            for mult in (1.0, -1.0, 0.5, -0.5):
                x = x0.copy()
                x[i] += mult * hs[i]
                rows.append(x)
        if batched:
            lls = self.log_likelihood_batch(np.stack(rows))
        else:
            lls = []
            try:
                for row in rows:
                    for (o, n), v in zip(params, row):
                        setattr(o, n, float(v))
                    lls.append(float(self.maximum_log_likelihood_objective()))
            finally:
                for (o, n), v in zip(params, x0):
                    setattr(o, n, float(v))
            lls = np.array(lls)
        grad = np.zeros(len(params))
        for j, i in enumerate(idx):
            up, dn, up2, dn2 = lls[1 + 4 * j:5 + 4 * j]
            d1 = (up - dn) / (2.0 * hs[i])
            d2 = (up2 - dn2) / hs[i]
            grad[i] = (4.0 * d2 - d1) / 3.0
        return float(lls[0]), grad

    @_public_evaluation
    def log_likelihood_batch(self, thetas):
        """Marginal log-likelihoods at B hyper-parameter settings in one call: `thetas` is (B, P) in the
        order of `trainable_parameters()`.  The B filters share the series and run side by side on
        the GPU (pgps_gp_ll_batch_*) -- the evaluation pattern of the reference's HMC / grid-search
        drivers (pssgp/experiments/*/mcmc.py), which loop over maximum_log_likelihood_objective."""
        if not self.parallel:
            raise NotImplementedError("batched evaluation runs on the parallel (HIP) path: construct with parallel=True")
        from . import _backend
        thetas = np.atleast_2d(np.asarray(thetas, np.float64))
        params = self.trainable_parameters()
        if thetas.shape[1] != len(params):
            raise ValueError(f"thetas has {thetas.shape[1]} columns, the model {len(params)} trainable parameters")
        saved = [getattr(o, n) for o, n in params]
        models, general = [], []
        ts, Y = self.data
        stream = None                   # state dimensions 17..32: asynchronous single evaluations, see below
        try:
            memo = {}                   # kernel parameters -> its SDE: settings that differ in the noise only share it
            from .kernels import Matern12, Matern32, Matern52, RBF
            leaf_variance = isinstance(self.kernel, (Matern12, Matern32, Matern52, RBF)) and params[0] == (self.kernel, "variance")
            for row in thetas:
                for (o, n), v in zip(params, row):
                    setattr(o, n, float(v))
                key = tuple(float(v) for v in row[:-1])
                if key not in memo and leaf_variance and key[0] != 0.0:
                    # a single Matern / RBF kernel: F, H do not depend on its variance and Pinf is linear in it
                    # (balance_ss normalises L and H, q carries the scale) -- one SDE per lengthscale
                    for other, (fo, Fo, Po, Ho) in list(memo.items()):
                        if other[1:] == key[1:] and other[0] != 0.0:
                            memo[key] = (fo, Fo, Po * (key[0] / other[0]), Ho)
                            break
                if key not in memo and leaf_variance and key[0] != 0.0 and len(key) == 2 and key[1] > 0.0:
                    # ... and its lengthscale is a scaling of time: k(tau / l).  The state-space model at lengthscale l is
                    # the one at l0 with F multiplied by l0 / l (same Pinf / variance, same H) -- the same model as
                    # get_sde() builds, in a realisation that differs from it by a diagonal similarity where the
                    # balancing sweeps have not converged (RBF), i.e. the same likelihood up to rounding.  Used for nearby
                    # lengthscales only (the perturbations of a difference quotient, the steps of a sampler): one SDE
                    # construction (0.2 - 0.4 ms for RBF order 6) instead of one per row.
                    for other, (fo, Fo, Po, Ho) in list(memo.items()):
                        if other[0] != 0.0 and other[1] > 0.0 and 0.8 <= key[1] / other[1] <= 1.25:
                            r = other[1] / key[1]
                            form = None if fo is None else (fo[0] * r, fo[1] * r, fo[2] * (r * r))
                            memo[key] = (form, Fo * r, Po * (key[0] / other[0]), Ho)
                            break
                if key not in memo:
                    sde = self.kernel.get_sde()
                    memo[key] = (_backend.nilpotent_form(sde.F), np.asarray(sde.F, np.float64), np.asarray(sde.P0, np.float64),
                                 np.asarray(sde.H, np.float64).reshape(-1))
                form, F, P0, H = memo[key]
                if form is not None:
                    models.append((form, P0, H, self.noise_variance))
                d = F.shape[0]
                if _backend.LTI_BATCH_DIM_MAX < d <= _backend.LTI_DIM_MAX:
                    # one device evaluation per setting on the wave-cooperative kernels, enqueued as soon as its model
                    # exists: the device runs setting i while the host builds the SDE of setting i + 1
                    if stream is None:
                        stream = _backend.LtiLlStream(ts.reshape(-1), Y.reshape(-1), thetas.shape[0])
                    stream.push(F, P0, H, self.noise_variance)
                    continue
                # kernels without the closed-form discretisation (RBF, Periodic, sums, products): general-LTI batch
                general.append((F, P0, H, self.noise_variance))
            if stream is not None:
                if stream.count != thetas.shape[0]:
                    raise ValueError("all settings of a batch must give the same state dimension")
                out, stream = stream.finish(), None
                return out
        finally:
            for (o, n), v in zip(params, saved):
                setattr(o, n, v)
            if stream is not None:
                stream.close()
        if len(models) == len(general):
            return _backend.gp_ll_batch(models, ts.reshape(-1), Y.reshape(-1))
        d = general[0][0].shape[0]
        if _backend.LTI_DIM_MIN <= d <= _backend.LTI_BATCH_DIM_MAX:
            ser = self._device_series() if ts.dtype == np.float64 else None
            if ser is not None and ser.has_lti:
                return ser.lti_ll_batch(general)
            return _backend.lti_ll_batch(general, ts.reshape(-1), Y.reshape(-1))
        raise NotImplementedError(f"batched evaluation covers the Matern family and state dimensions "
                                  f"{_backend.LTI_DIM_MIN}..{_backend.LTI_DIM_MAX}; this kernel has d = {d}")

    def log_posterior_density(self):
        return self.maximum_log_likelihood_objective()

    def training_loss(self):
        return -self.maximum_log_likelihood_objective()
