"""ctypes binding of libpgps.so (include/pgps.h) -- the only way pssgp reaches the GPU.

There is no CPU fallback: if the library is missing or no MI355X is visible the calls raise.
"""
import ctypes
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("PGPS_LIB", os.path.join(_HERE, "libpgps.so"))

PGPS_OK = 0
E_UNSUPPORTED_DIM = -2
PGPS_K_NAMES = None
COMM_ID_BYTES = 128             # PGPS_COMM_ID_BYTES

c_void_p, c_int, c_long, c_double, c_float = (ctypes.c_void_p, ctypes.c_int, ctypes.c_long,
                                              ctypes.c_double, ctypes.c_float)


class PgpsError(RuntimeError):
    def __init__(self, code, what, detail=""):
        self.code = code
        super().__init__(f"libpgps: {what} (code {code}){': ' + detail if detail else ''}")


_lib = None
_lib_lock = threading.Lock()


def _declare(lib):
    P = c_void_p
    lib.pgps_version.restype = c_int
    lib.pgps_strerror.restype = ctypes.c_char_p
    lib.pgps_strerror.argtypes = [c_int]
    lib.pgps_last_hip_error.restype = ctypes.c_char_p
    lib.pgps_last_hip_error.argtypes = [P]
    lib.pgps_kernel_name.restype = ctypes.c_char_p
    lib.pgps_kernel_name.argtypes = [c_int]
    lib.pgps_device_count.argtypes = [ctypes.POINTER(c_int)]
    lib.pgps_create.argtypes = [c_int, ctypes.POINTER(P)]
    lib.pgps_destroy.argtypes = [P]
    lib.pgps_set_stream.argtypes = [P, P]
    lib.pgps_use_own_stream.argtypes = [P]
    lib.pgps_synchronize.argtypes = [P]
    lib.pgps_status.argtypes = [P, ctypes.POINTER(c_int)]
    lib.pgps_set_chunk.argtypes = [P, c_int]
    lib.pgps_set_stage.argtypes = [P, c_int]
    lib.pgps_set_family.argtypes = [P, c_int]
    lib.pgps_set_block.argtypes = [P, c_int]
    if hasattr(lib, "pgps_set_dma"):
        lib.pgps_set_dma.argtypes = [P, c_int]
    if hasattr(lib, "pgps_set_resident"):       # (absent from libraries built before round 5: A/B runs load those)
        lib.pgps_set_resident.argtypes = [P, c_int]
        lib.pgps_resident_stamps.argtypes = [P, P, c_int, ctypes.POINTER(c_int)]
    if hasattr(lib, "pgps_set_shortcut"):
        lib.pgps_set_shortcut.argtypes = [P, c_int]
    if hasattr(lib, "pgps_set_rc_scan"):
        lib.pgps_set_rc_scan.argtypes = [P, c_int]
    if hasattr(lib, "pgps_set_one_launch"):
        lib.pgps_set_one_launch.argtypes = [P, c_int]
    lib.pgps_get_geometry.argtypes = [P, c_long, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int)]
    lib.pgps_set_single_pass.argtypes = [P, c_int, c_int]
    lib.pgps_get_chunk.argtypes = [P, c_long, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]
    lib.pgps_malloc.argtypes = [P, ctypes.c_size_t, ctypes.POINTER(P)]
    lib.pgps_free.argtypes = [P, P]
    lib.pgps_memcpy_h2d.argtypes = [P, P, P, ctypes.c_size_t]
    lib.pgps_memcpy_d2h.argtypes = [P, P, P, ctypes.c_size_t]
    lib.pgps_profile_enable.argtypes = [P, c_int]
    lib.pgps_profile_sample.argtypes = [P, c_int]
    lib.pgps_profile_calibrate.argtypes = [P, ctypes.POINTER(c_double)]
    lib.pgps_profile_read.argtypes = [P, ctypes.POINTER(c_double), ctypes.POINTER(c_long), c_int]
    lib.pgps_seg_record_len.argtypes = [c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]
    lib.pgps_comm_get_unique_id.argtypes = [P]
    lib.pgps_comm_init.argtypes = [P, P, c_int, c_int]
    lib.pgps_comm_destroy.argtypes = [P]
    lib.pgps_comm_info.argtypes = [P, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]
    if hasattr(lib, "pgps_comm_count"):         # (absent from libraries built before round 3: A/B runs load those)
        lib.pgps_comm_count.argtypes = [P, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]
    if hasattr(lib, "pgps_set_grad_pack"):
        lib.pgps_set_grad_pack.argtypes = [P, c_long]
    if hasattr(lib, "pgps_comm_library"):
        lib.pgps_comm_library.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
    lib.pgps_comm_allgather_dev.argtypes = [P, P, P, ctypes.c_size_t]
    for suf, real in (("f64", c_double), ("f32", c_float)):
        for dev in ("", "_dev"):
            getattr(lib, f"pgps_discretise{dev}_{suf}").argtypes = [P, c_long, c_int, P, P, P, real, P, P]
            getattr(lib, f"pgps_pkf{dev}_{suf}").argtypes = [P, c_long, c_int, P, P, P, P, real, P, P, P, P]
            getattr(lib, f"pgps_pks{dev}_{suf}").argtypes = [P, c_long, c_int, P, P, P, P, P, P]
            getattr(lib, f"pgps_pkfs{dev}_{suf}").argtypes = [P, c_long, c_int, P, P, P, P, real, P, P, P, P, P, P]
        for dev in ("", "_dev"):
            getattr(lib, f"pgps_gp{dev}_{suf}").argtypes = [P, c_long, c_int, c_double, P, P, P, P, c_double, P, c_double,
                                                            P, P, P, P, P, P]
        getattr(lib, f"pgps_seg_filter_reduce_dev_{suf}").argtypes = [P, c_long, c_int, c_int, c_int, P, P, P, P, real,
                                                                      P, P]
        getattr(lib, f"pgps_seg_filter_apply_dev_{suf}").argtypes = [P, c_long, c_int, c_int, c_int, P, P, P, P, real,
                                                                     P, P, P, P, P]
        getattr(lib, f"pgps_seg_smoother_apply_dev_{suf}").argtypes = [P, c_long, c_int, c_int, c_int, P, P, P, P, P,
                                                                       P, P, P]
        getattr(lib, f"pgps_pkfs_seg_dev_{suf}").argtypes = [P, c_long, c_int, P, P, P, P, real, P, P, P, P, P, P]
    for dev in ("", "_dev"):
        getattr(lib, f"pgps_gp_ll_grad_blocks{dev}_f64").argtypes = [P, c_long, c_int, c_int, P, c_int, P, P, c_double, P, P]
    return lib


def load_library():
    """Load libpgps.so (once).  Raises if it has not been built (`python __graft_entry__.py`)."""
    global _lib
    with _lib_lock:
        if _lib is None:
            if not os.path.exists(_LIB_PATH):
                raise PgpsError(-100, "library not built", f"{_LIB_PATH} missing; run "
                                "`make -C parallel-gps_amd/csrc` (or __graft_entry__.build())")
            _lib = _declare(ctypes.CDLL(_LIB_PATH))
        return _lib


def check(ctx, code, what):
    if code != PGPS_OK:
        lib = load_library()
        msg = lib.pgps_strerror(code).decode()
        detail = lib.pgps_last_hip_error(ctx.handle).decode() if ctx is not None and code in (-3, -7) else ""
        raise PgpsError(code, f"{what}: {msg}", detail)


class Context:
    """One libpgps context = one GPU + one stream + scratch (+ the RCCL communicator of a sharded series).

    libpgps allows one call in flight per context (include/pgps.h); `lock` (re-entrant) serialises the calls made
    through this object, so Python threads that share a context -- e.g. parallel MCMC chains on the default
    context of `get_context` -- take turns instead of racing on its scratch.  Multi-call sequences that keep state
    in the context (LtiLlStream, the three-phase segment protocol) hold the lock for their whole duration."""

    def __init__(self, device=0):
        self.lock = threading.RLock()
        self.lib = load_library()
        n = c_int(0)
        self.lib.pgps_device_count(ctypes.byref(n))
        if n.value <= 0:
            raise PgpsError(-6, "no HIP device visible: the parallel path needs an MI355X (there is no CPU fallback)")
        self.handle = c_void_p()
        code = self.lib.pgps_create(int(device), ctypes.byref(self.handle))
        if code != PGPS_OK:
            raise PgpsError(code, "pgps_create: " + self.lib.pgps_strerror(code).decode())
        self.device = int(device)

    def close(self):
        if getattr(self, "handle", None) is not None and self.handle.value:
            with self.lock:
                self.lib.pgps_destroy(self.handle)
                self.handle = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- knobs ---------------------------------------------------------------------------
    def set_chunk(self, steps_per_lane):
        check(self, self.lib.pgps_set_chunk(self.handle, int(steps_per_lane)), "pgps_set_chunk")

    def set_stage(self, steps_per_subtile):
        """-1 auto, 0 direct global accesses, 2 / 4 steps per LDS-staged sub-tile."""
        check(self, self.lib.pgps_set_stage(self.handle, int(steps_per_subtile)), "pgps_set_stage")

    def set_single_pass(self, mode=-1, window=0):
        """Single-pass filter kernel: -1 auto, 0 off, 1 on; window = look-back window in tiles (0 = keep)."""
        check(self, self.lib.pgps_set_single_pass(self.handle, int(mode), int(window)), "pgps_set_single_pass")

    def set_family(self, family):
        """0 auto, 1 lane-chunk kernels (d <= 6), 2 wave-cooperative (any d <= 32), 3 row-cooperative (2 <= d <= 16), 4 quad-cooperative
        level-1 kernels (fp32, 5 <= d <= 8)."""
        check(self, self.lib.pgps_set_family(self.handle, int(family)), "pgps_set_family")

    def get_family(self, n_steps, d, f32=False, what=2):
        """PGPS_FAMILY_* code of the kernels a pkf (what = 0) / pks (1) / pkfs (2) / segment-phase (3) call will run on
        (1 lane-chunk 256 lanes, 11 lane-chunk 128 lanes, 2 wave-cooperative, 3 row-cooperative, 4 quad-cooperative, 5 two-rows);
        None with a library that cannot say."""
        if not hasattr(self.lib, "pgps_get_family"):
            return None
        fam = c_int(0)
        check(self, self.lib.pgps_get_family(self.handle, c_long(int(n_steps)), c_int(int(d)), c_int(1 if f32 else 0),
                                             c_int(int(what)), ctypes.byref(fam)), "pgps_get_family")
        return fam.value

    def set_f32_policy(self, policy):
        """float32 series through a smoother (pkfs / pks): 0 = fp64 arithmetic on the float32 arrays where the grid is too
        dense for float32 arithmetic (probed per call on the device; always above d = 16), 1 = float32 arithmetic whatever
        the grid, 2 = always fp64 arithmetic.  `status()` & 4 tells whether a call was promoted."""
        if not hasattr(self.lib, "pgps_set_f32_policy"):
            raise RuntimeError("this libpgps has no pgps_set_f32_policy")
        check(self, self.lib.pgps_set_f32_policy(self.handle, int(policy)), "pgps_set_f32_policy")

    def set_block(self, lanes):
        """Lanes per workgroup of the lane-chunk kernels: 0 = automatic, 128, 256 (pgps_set_block)."""
        check(self, self.lib.pgps_set_block(self.handle, int(lanes)), "pgps_set_block")

    def set_dma(self, mode):
        """LDS-DMA ring in the Kalman pass (d = 2 fp64, 128-lane build): -1 automatic, 0 off, 1 on (pgps_set_dma)."""
        if hasattr(self.lib, "pgps_set_dma"):
            check(self, self.lib.pgps_set_dma(self.handle, int(mode)), "pgps_set_dma")

    def set_resident(self, mode):
        """Filter + smoother in ONE resident launch (fp64, d = 2, up to 4096 steps per CU): -1 automatic (from 2^17 steps),
        0 never, 1 wherever the series fits, 2 = 1 + in-kernel phase stamps (pgps_set_resident)."""
        if hasattr(self.lib, "pgps_set_resident"):
            check(self, self.lib.pgps_set_resident(self.handle, int(mode)), "pgps_set_resident")

    def set_shortcut(self, on):
        """Forgetting shortcut for the carry across workgroups (lane-chunk and resident kernels): 1 where it applies (default), 0 never."""
        if hasattr(self.lib, "pgps_set_shortcut"):
            check(self, self.lib.pgps_set_shortcut(self.handle, int(on)), "pgps_set_shortcut")

    def resident_stamps(self):
        """(workgroups, 16) cycle stamps of the last resident launch made under set_resident(2) (diagnostics)."""
        n = c_int(0)
        check(self, self.lib.pgps_resident_stamps(self.handle, None, 0, ctypes.byref(n)), "pgps_resident_stamps")
        out = np.zeros((max(n.value, 1), 16), np.int64)
        check(self, self.lib.pgps_resident_stamps(self.handle, _ptr(out), n.value, ctypes.byref(n)), "pgps_resident_stamps")
        return out[:n.value]

    def set_one_launch(self, max_steps):
        """Fused calls of short series in ONE launch up to max_steps steps: -1 automatic (2048 steps: kOneLaunchAuto), 0 never."""
        if hasattr(self.lib, "pgps_set_one_launch"):
            check(self, self.lib.pgps_set_one_launch(self.handle, int(max_steps)), "pgps_set_one_launch")

    def set_grad_pack(self, max_steps):
        """Gradient calls at d <= 2: one derivative direction per model up to this many steps (-1 automatic, 0 never)."""
        if hasattr(self.lib, "pgps_set_grad_pack"):
            check(self, self.lib.pgps_set_grad_pack(self.handle, int(max_steps)), "pgps_set_grad_pack")

    def set_rc_scan(self, mode):
        """Scans of the chain totals (row- / quad-cooperative families): -1 automatic, 0 one launch per level, 1 blocked."""
        if hasattr(self.lib, "pgps_set_rc_scan"):
            check(self, self.lib.pgps_set_rc_scan(self.handle, int(mode)), "pgps_set_rc_scan")

    def get_geometry(self, n, d):
        """(lanes per workgroup, steps per lane, workgroups) of a lane-chunk call of n steps at state dimension d <= 6."""
        lanes, lc, nb = c_int(0), c_int(0), c_int(0)
        check(self, self.lib.pgps_get_geometry(self.handle, c_long(int(n)), c_int(int(d)), ctypes.byref(lanes),
                                               ctypes.byref(lc), ctypes.byref(nb)), "pgps_get_geometry")
        return lanes.value, lc.value, nb.value

    def get_chunk(self, n_steps):
        lc, nb = c_int(0), c_int(0)
        check(self, self.lib.pgps_get_chunk(self.handle, int(n_steps), ctypes.byref(lc), ctypes.byref(nb)),
              "pgps_get_chunk")
        return lc.value, nb.value

    def set_stream(self, hip_stream_handle):
        """Launch on an external hipStream_t (an integer handle; 0 / None = the HIP null stream)."""
        check(self, self.lib.pgps_set_stream(self.handle, c_void_p(hip_stream_handle or 0)), "pgps_set_stream")

    def use_own_stream(self):
        check(self, self.lib.pgps_use_own_stream(self.handle), "pgps_use_own_stream")

    def status(self):
        """Device-side diagnostic flags since the last call (0 = none); synchronises."""
        f = c_int(0)
        check(self, self.lib.pgps_status(self.handle, ctypes.byref(f)), "pgps_status")
        return f.value

    def synchronize(self):
        check(self, self.lib.pgps_synchronize(self.handle), "pgps_synchronize")

    def profile_enable(self, mask=0x7f):
        """mask: bit i = time every launch of kernel slot i (PGPS_K_*); 0 = off."""
        check(self, self.lib.pgps_profile_enable(self.handle, int(mask)), "pgps_profile_enable")

    def profile_sample(self, every_n):
        check(self, self.lib.pgps_profile_sample(self.handle, int(every_n)), "pgps_profile_sample")

    def profile_calibrate(self):
        """Mean elapsed ms of an empty hipEvent pair (the pair's own contribution to a timing)."""
        v = c_double(0.0)
        check(self, self.lib.pgps_profile_calibrate(self.handle, ctypes.byref(v)), "pgps_profile_calibrate")
        return v.value

    def profile_read(self, reset=True):
        k = 0                                   # slots of THIS library (PGPS_K_COUNT: 6 until round 4, 7 with the resident launch)
        while k < 16 and self.lib.pgps_kernel_name(k):
            k += 1
        ms = (c_double * 16)()
        cnt = (c_long * 16)()
        check(self, self.lib.pgps_profile_read(self.handle, ms, cnt, int(bool(reset))), "pgps_profile_read")
        return {self.lib.pgps_kernel_name(i).decode(): (ms[i], cnt[i]) for i in range(k)}

    # -- raw device memory -----------------------------------------------------------------
    def malloc(self, nbytes):
        p = c_void_p()
        with self.lock:
            check(self, self.lib.pgps_malloc(self.handle, int(nbytes), ctypes.byref(p)), "pgps_malloc")
        return p.value

    def free(self, ptr):
        with self.lock:
            check(self, self.lib.pgps_free(self.handle, c_void_p(ptr)), "pgps_free")

    def h2d(self, dptr, arr):
        arr = np.ascontiguousarray(arr)
        with self.lock:
            check(self, self.lib.pgps_memcpy_h2d(self.handle, c_void_p(dptr), arr.ctypes.data_as(c_void_p), arr.nbytes),
                  "pgps_memcpy_h2d")

    def d2h(self, arr, dptr):
        assert arr.flags["C_CONTIGUOUS"]
        with self.lock:
            check(self, self.lib.pgps_memcpy_d2h(self.handle, arr.ctypes.data_as(c_void_p), c_void_p(dptr), arr.nbytes),
                  "pgps_memcpy_d2h")

    # -- the communicator of a series sharded over GPUs (RCCL, owned by the context) -------------
    @staticmethod
    def comm_library():
        """Which RCCL libpgps uses -- loaded on first use: the copy already in the process (torch's, when
        torch.distributed's nccl backend is there) or the loader's librccl.so.1 -- or, prefixed with "unavailable: ", why
        there is none."""
        lib = load_library()
        if not hasattr(lib, "pgps_comm_library"):
            return "linked at build time"
        buf = ctypes.create_string_buffer(512)
        code = lib.pgps_comm_library(buf, 512)
        text = buf.value.decode(errors="replace")
        return text if code == 0 else "unavailable: " + text

    @staticmethod
    def comm_unique_id():
        """128 opaque bytes from one rank, to be handed to every rank's comm_init (any transport: a file, MPI, a
        torch.distributed / TCP store ...)."""
        buf = ctypes.create_string_buffer(COMM_ID_BYTES)
        code = load_library().pgps_comm_get_unique_id(buf)
        if code != PGPS_OK:
            raise PgpsError(code, "pgps_comm_get_unique_id")
        return buf.raw

    def comm_init(self, unique_id, rank, nranks):
        if len(unique_id) != COMM_ID_BYTES:
            raise ValueError(f"unique id must be {COMM_ID_BYTES} bytes")
        with self.lock:
            check(self, self.lib.pgps_comm_init(self.handle, ctypes.c_char_p(bytes(unique_id)), int(rank), int(nranks)),
                  "pgps_comm_init")

    def comm_destroy(self):
        with self.lock:
            check(self, self.lib.pgps_comm_destroy(self.handle), "pgps_comm_destroy")

    def comm_info(self):
        r, n = c_int(0), c_int(0)
        check(self, self.lib.pgps_comm_info(self.handle, ctypes.byref(r), ctypes.byref(n)), "pgps_comm_info")
        return r.value, n.value

    def comm_count(self):
        """(ranks, this rank) as RCCL itself reports them for the context's communicator (ncclCommCount /
        ncclCommUserRank); (0, 0) without a communicator."""
        n, r = c_int(0), c_int(0)
        if not hasattr(self.lib, "pgps_comm_count"):        # (a library built before round 3: A/B runs load those)
            return 0, 0
        check(self, self.lib.pgps_comm_count(self.handle, ctypes.byref(n), ctypes.byref(r)), "pgps_comm_count")
        return n.value, r.value

    # -- generic call by name ----------------------------------------------------------------
    def call(self, name, *args):
        with self.lock:
            check(self, getattr(self.lib, name)(self.handle, *args), name)


_contexts = {}
_ctx_lock = threading.Lock()


def get_context(device=0):
    with _ctx_lock:
        ctx = _contexts.get(device)
        if ctx is None:
            ctx = _contexts[device] = Context(device)
        return ctx


def _suffix(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float64:
        return "f64", c_double
    if dtype == np.float32:
        return "f32", c_float
    raise TypeError(f"unsupported dtype {dtype}; use float32 or float64")


def _ptr(a):
    return None if a is None else a.ctypes.data_as(c_void_p)


def _prep(a, dtype, shape=None):
    a = np.ascontiguousarray(a, dtype=dtype)
    if shape is not None:
        a = a.reshape(shape)
    return a


def discretise(F, Pinf, ts, t0=0.0, device=0):
    """Fs, Qs = expm(dt F), Pinf - Fs Pinf Fs^T on the GPU (host arrays in and out)."""
    dtype = np.asarray(ts).dtype if np.asarray(ts).dtype in (np.float32, np.float64) else np.float64
    suf, real = _suffix(dtype)
    F = _prep(F, dtype)
    d = F.shape[0]
    Pinf = _prep(Pinf, dtype, (d, d))
    ts = _prep(ts, dtype, (-1,))
    N = ts.shape[0]
    Fs = np.empty((N, d, d), dtype)
    Qs = np.empty((N, d, d), dtype)
    get_context(device).call(f"pgps_discretise_{suf}", c_long(N), c_int(d), _ptr(F), _ptr(Pinf), _ptr(ts),
                             real(float(t0)), _ptr(Fs), _ptr(Qs))
    return Fs, Qs


def _unpack_lgssm(lgssm, dtype):
    P0, Fs, Qs, H, R = lgssm
    Fs = _prep(Fs, dtype)
    N, d = Fs.shape[0], Fs.shape[1]
    return (_prep(P0, dtype, (d, d)), Fs, _prep(Qs, dtype, (N, d, d)), _prep(H, dtype, (d,)),
            float(np.asarray(R).reshape(())), N, d)


def _dtype_of(lgssm):
    dt = np.asarray(lgssm[1]).dtype
    return dt if dt in (np.float32, np.float64) else np.dtype(np.float64)


def pkf(lgssm, observations, return_loglikelihood=False, device=0):
    dtype = _dtype_of(lgssm)
    suf, real = _suffix(dtype)
    P0, Fs, Qs, H, R, N, d = _unpack_lgssm(lgssm, dtype)
    ys = _prep(observations, dtype, (-1,))
    if ys.shape[0] != N:
        raise ValueError(f"observations has {ys.shape[0]} rows, the model {N} steps")
    fms = np.empty((N, d), dtype)
    fPs = np.empty((N, d, d), dtype)
    ll = c_double(0.0)
    get_context(device).call(f"pgps_pkf_{suf}", c_long(N), c_int(d), _ptr(P0), _ptr(Fs), _ptr(Qs), _ptr(H), real(R),
                             _ptr(ys), _ptr(fms), _ptr(fPs),
                             ctypes.cast(ctypes.byref(ll), c_void_p) if return_loglikelihood else None)
    if return_loglikelihood:
        return fms, fPs, np.asarray(ll.value, dtype=dtype)
    return fms, fPs


def pks(lgssm, ms, Ps, device=0):
    dtype = _dtype_of(lgssm)
    suf, _ = _suffix(dtype)
    _, Fs, Qs, *_ = lgssm
    Fs = _prep(Fs, dtype)
    N, d = Fs.shape[0], Fs.shape[1]
    Qs = _prep(Qs, dtype, (N, d, d))
    ms = _prep(ms, dtype, (N, d))
    Ps = _prep(Ps, dtype, (N, d, d))
    sms = np.empty((N, d), dtype)
    sPs = np.empty((N, d, d), dtype)
    get_context(device).call(f"pgps_pks_{suf}", c_long(N), c_int(d), _ptr(Fs), _ptr(Qs), _ptr(ms), _ptr(Ps),
                             _ptr(sms), _ptr(sPs))
    return sms, sPs


def pkfs(lgssm, observations, return_filtered=False, return_loglikelihood=False, device=0):
    dtype = _dtype_of(lgssm)
    suf, real = _suffix(dtype)
    P0, Fs, Qs, H, R, N, d = _unpack_lgssm(lgssm, dtype)
    ys = _prep(observations, dtype, (-1,))
    if ys.shape[0] != N:
        raise ValueError(f"observations has {ys.shape[0]} rows, the model {N} steps")
    fms = np.empty((N, d), dtype) if return_filtered else None
    fPs = np.empty((N, d, d), dtype) if return_filtered else None
    sms = np.empty((N, d), dtype)
    sPs = np.empty((N, d, d), dtype)
    ll = c_double(0.0)
    get_context(device).call(f"pgps_pkfs_{suf}", c_long(N), c_int(d), _ptr(P0), _ptr(Fs), _ptr(Qs), _ptr(H), real(R),
                             _ptr(ys), _ptr(fms), _ptr(fPs), _ptr(sms), _ptr(sPs),
                             ctypes.cast(ctypes.byref(ll), c_void_p))
    out = (sms, sPs)
    if return_filtered:
        out += (fms, fPs)
    if return_loglikelihood:
        out += (np.asarray(ll.value, dtype=dtype),)
    return out


# ----------------------------------------------------------------------------------------------
# fused path: discretisation inside the scan kernels (SDEs with F = -lam I + N, N nilpotent, d <= 3)
# ----------------------------------------------------------------------------------------------
def nilpotent_form(F, tol=1e-9):
    """(lam, N1, N2) with F = -lam I + N, N^d = 0, N1 = N, N2 = N^2 / 2 -- or None if F is not of
    that form (it is for every Matern kernel, balanced or not) or d > 3."""
    F = np.asarray(F, dtype=np.float64)
    d = F.shape[0]
    if d > 3:
        return None
    lam = -float(np.trace(F)) / d
    N = F + lam * np.eye(d)
    Np = np.linalg.matrix_power(N, d)
    scale = max(1.0, float(np.max(np.abs(F)))) ** d
    if not np.all(np.abs(Np) <= tol * scale) or lam <= 0:
        return None
    return lam, np.ascontiguousarray(N), np.ascontiguousarray(0.5 * (N @ N))


def gp(form, Pinf, H, R, ts, ys, t0=0.0, want_filtered=False, want_smoothed=False, device=0):
    """Fused filter (+ smoother) + log-likelihood straight from times and observations.

    Returns a dict with "ll" and, on request, "fms", "fPs", "sms", "sPs" (host numpy arrays)."""
    lam, N1, N2 = form
    ts_a = np.asarray(ts)
    dtype = ts_a.dtype if ts_a.dtype in (np.float32, np.float64) else np.dtype(np.float64)
    suf, _ = _suffix(dtype)
    ts_a = _prep(ts_a, dtype, (-1,))
    ys_a = _prep(ys, dtype, (-1,))
    N = ts_a.shape[0]
    if ys_a.shape[0] != N:
        raise ValueError(f"observations has {ys_a.shape[0]} rows, the series {N} steps")
    d = N1.shape[0]
    Pinf = _prep(Pinf, np.float64, (d, d))
    H = _prep(H, np.float64, (d,))
    want_filtered = want_filtered or want_smoothed
    out = {}
    fms = np.empty((N, d), dtype) if want_filtered else None
    fPs = np.empty((N, d, d), dtype) if want_filtered else None
    sms = np.empty((N, d), dtype) if want_smoothed else None
    sPs = np.empty((N, d, d), dtype) if want_smoothed else None
    ll = c_double(0.0)
    get_context(device).call(f"pgps_gp_{suf}", c_long(N), c_int(d), c_double(lam), _ptr(_prep(N1, np.float64)),
                             _ptr(_prep(N2, np.float64)), _ptr(Pinf), _ptr(H), c_double(float(R)), _ptr(ts_a),
                             c_double(float(t0)), _ptr(ys_a), _ptr(fms), _ptr(fPs), _ptr(sms), _ptr(sPs),
                             ctypes.cast(ctypes.byref(ll), c_void_p))
    out["ll"] = np.asarray(ll.value, dtype=dtype)
    if want_filtered:
        out["fms"], out["fPs"] = fms, fPs
    if want_smoothed:
        out["sms"], out["sPs"] = sms, sPs
    return out


def gp_predict(form, Pinf, H, R, ts, ys, tq, t0=0.0, device=0):
    """predict_f on the device (pgps_gp_predict_*): merge of the sorted `ts` (N) and `tq` (K), fused
    filter + smoother over the N + K steps, posterior mean / variance of f = H x at the K query
    times.  Returns (mean (K,), var (K,), ll of the training series)."""
    lam, N1, N2 = form
    ts_a = np.asarray(ts)
    dtype = ts_a.dtype if ts_a.dtype in (np.float32, np.float64) else np.dtype(np.float64)
    suf, _ = _suffix(dtype)
    ts_a = _prep(ts_a, dtype, (-1,))
    ys_a = _prep(ys, dtype, (-1,))
    tq_a = _prep(tq, dtype, (-1,))
    N, K = ts_a.shape[0], tq_a.shape[0]
    if ys_a.shape[0] != N:
        raise ValueError(f"observations has {ys_a.shape[0]} rows, the series {N} steps")
    d = N1.shape[0]
    mean, var = np.empty(K, dtype), np.empty(K, dtype)
    ll = c_double(0.0)
    get_context(device).call(f"pgps_gp_predict_{suf}", c_long(N), c_long(K), c_int(d), c_double(lam),
                             _ptr(_prep(N1, np.float64)), _ptr(_prep(N2, np.float64)), _ptr(_prep(Pinf, np.float64, (d, d))),
                             _ptr(_prep(H, np.float64, (d,))), c_double(float(R)), _ptr(ts_a), _ptr(ys_a),
                             c_double(float(t0)), _ptr(tq_a), _ptr(mean), _ptr(var),
                             ctypes.cast(ctypes.byref(ll), c_void_p))
    return mean, var, ll.value


# state dimensions of the general-LTI device path: row-cooperative kernels up to 16 (batched evaluation only there),
# wave-cooperative kernels from 17 to 32
LTI_DIM_MIN, LTI_DIM_MAX, LTI_BATCH_DIM_MAX = 2, 32, 16


def _lti_model(F, Pinf, H):
    F = _prep(F, np.float64)
    d = F.shape[0]
    if not (LTI_DIM_MIN <= d <= LTI_DIM_MAX):
        raise ValueError(f"the general-LTI device path covers state dimensions {LTI_DIM_MIN}..{LTI_DIM_MAX}, got {d}")
    return F, _prep(Pinf, np.float64, (d, d)), _prep(H, np.float64, (d,)), d


def lti_ll(F, Pinf, H, R, ts, ys, t0=0.0, device=0):
    """Log-likelihood of any LTI state-space GP on the device (pgps_lti_ll_f64): discretisation, parallel filter
    and the likelihood terms with nothing written per step.  `Pinf` must be the stationary covariance of the SDE
    (it is every kernel's P0)."""
    F, Pinf, H, d = _lti_model(F, Pinf, H)
    ts_a = _prep(ts, np.float64, (-1,))
    ys_a = _prep(ys, np.float64, (-1,))
    if ys_a.shape[0] != ts_a.shape[0]:
        raise ValueError(f"observations has {ys_a.shape[0]} rows, the series {ts_a.shape[0]} steps")
    ll = c_double(0.0)
    get_context(device).call("pgps_lti_ll_f64", c_long(ts_a.shape[0]), c_int(d), _ptr(F), _ptr(Pinf), _ptr(H),
                             c_double(float(R)), _ptr(ts_a), _ptr(ys_a), c_double(float(t0)),
                             ctypes.cast(ctypes.byref(ll), c_void_p))
    return ll.value


def split_grad_stats(out, d):
    """[ll | Abar | Ubar | Hbar | Rbar] of pgps_lti_ll_grad_* -> (ll, Abar (d, d), Ubar (d,), Hbar (d,), Rbar)."""
    dd = d * d
    return (float(out[0]), out[1:1 + dd].reshape(d, d).copy(), out[1 + dd:1 + dd + d].copy(),
            out[1 + dd + d:1 + dd + 2 * d].copy(), float(out[1 + dd + 2 * d]))


def contract_grad_stats(stats, H, grads):
    """d ll / d theta for the kernel's parameters (grads: [(dF, dPinf, dH)] of pssgp.kernels.sde_grads, every dF
    commuting with F) followed by d ll / d R -- the host half of the adjoint gradient (include/pgps.h,
    pgps_lti_ll_grad_f64)."""
    _, Abar, Ubar, Hbar, Rbar = stats
    h = np.asarray(H, np.float64).reshape(-1)
    g = [float(np.sum(Abar * dF) + Ubar @ (np.asarray(dP, np.float64) @ h) + Hbar @ np.asarray(dH, np.float64).reshape(-1))
         for dF, dP, dH in grads]
    return np.array(g + [Rbar], np.float64)


def lti_ll_grad(F, Pinf, H, R, ts, ys, t0=0.0, device=0):
    """Log-likelihood and the model's adjoints of any LTI state-space GP (pgps_lti_ll_grad_f64): one filter pass and one
    reverse pass on the device, whatever the number of hyper-parameters.  Returns (ll, Abar, Ubar, Hbar, Rbar)."""
    F, Pinf, H, d = _lti_model(F, Pinf, H)
    ts_a = _prep(ts, np.float64, (-1,))
    ys_a = _prep(ys, np.float64, (-1,))
    if ys_a.shape[0] != ts_a.shape[0]:
        raise ValueError(f"observations has {ys_a.shape[0]} rows, the series {ts_a.shape[0]} steps")
    ctx = get_context(device)
    if not hasattr(ctx.lib, "pgps_lti_ll_grad_f64"):
        raise RuntimeError("this libpgps has no pgps_lti_ll_grad_* entry points")
    out = np.zeros(2 + d * d + 2 * d, np.float64)
    ctx.call("pgps_lti_ll_grad_f64", c_long(ts_a.shape[0]), c_int(d), _ptr(F), _ptr(Pinf), _ptr(H), c_double(float(R)),
             _ptr(ts_a), _ptr(ys_a), c_double(float(t0)), _ptr(out))
    return split_grad_stats(out, d)


def lti_predict(F, Pinf, H, R, ts, ys, tq, t0=0.0, device=0):
    """predict_f of any LTI state-space GP on the device (pgps_lti_predict_f64): merge of the sorted `ts` (N) and
    `tq` (K), discretisation, filter + smoother over the N + K steps, posterior mean / variance of f = H x at the K
    query times.  Returns (mean (K,), var (K,), ll of the training series)."""
    F, Pinf, H, d = _lti_model(F, Pinf, H)
    ts_a = _prep(ts, np.float64, (-1,))
    ys_a = _prep(ys, np.float64, (-1,))
    tq_a = _prep(tq, np.float64, (-1,))
    N, K = ts_a.shape[0], tq_a.shape[0]
    if ys_a.shape[0] != N:
        raise ValueError(f"observations has {ys_a.shape[0]} rows, the series {N} steps")
    mean, var = np.empty(K, np.float64), np.empty(K, np.float64)
    ll = c_double(0.0)
    get_context(device).call("pgps_lti_predict_f64", c_long(N), c_long(K), c_int(d), _ptr(F), _ptr(Pinf), _ptr(H),
                             c_double(float(R)), _ptr(ts_a), _ptr(ys_a), c_double(float(t0)), _ptr(tq_a), _ptr(mean),
                             _ptr(var), ctypes.cast(ctypes.byref(ll), c_void_p))
    return mean, var, ll.value


class LtiLlStream:
    """Log-likelihoods of several general LTI models over one series, one ASYNCHRONOUS device evaluation each
    (pgps_lti_ll_dev_f64): the series is uploaded once, the results are read once at the end, and the host prepares
    model i + 1 (its get_sde()) while the device runs model i.  For the state dimensions above the batch kernels'
    (17..32, e.g. the reference's CO2 kernel, d = 18)."""

    def __init__(self, ts, ys, capacity, t0=0.0, device=0):
        ts_a = _prep(ts, np.float64, (-1,))
        ys_a = _prep(ys, np.float64, (-1,))
        if ys_a.shape[0] != ts_a.shape[0]:
            raise ValueError(f"observations has {ys_a.shape[0]} rows, the series {ts_a.shape[0]} steps")
        self.n, self.t0, self.capacity, self.count = ts_a.shape[0], float(t0), int(capacity), 0
        self.d_ts = self.d_ys = self.d_ll = None
        # A context of its own: the asynchronous evaluations keep their scratch in the context between push() and
        # finish(), so the stream must own one -- and with a private context it never holds the shared default
        # context's lock (a stream dropped without finish() / close() used to leave that lock held for ever).
        self.ctx = Context(device)
        try:
            self.d_ts = self.ctx.malloc(ts_a.nbytes)
            self.d_ys = self.ctx.malloc(ys_a.nbytes)
            self.d_ll = self.ctx.malloc(8 * max(self.capacity, 1))
            self.ctx.h2d(self.d_ts, ts_a)
            self.ctx.h2d(self.d_ys, ys_a)
        except Exception:
            self.close()
            raise

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:       # noqa: BLE001 -- interpreter shutdown
            pass

    def push(self, F, Pinf, H, R):
        if self.count >= self.capacity:
            raise ValueError("LtiLlStream is full")
        F, Pinf, H, d = _lti_model(F, Pinf, H)
        self.ctx.call("pgps_lti_ll_dev_f64", c_long(self.n), c_int(d), _ptr(F), _ptr(Pinf), _ptr(H), c_double(float(R)),
                      c_void_p(self.d_ts), c_void_p(self.d_ys), c_double(self.t0), c_void_p(self.d_ll + 8 * self.count))
        self.count += 1

    def finish(self):
        """The log-likelihoods pushed so far (waits for the device); releases the device buffers."""
        out = np.empty(self.count, np.float64)
        try:
            if self.count:
                self.ctx.d2h(out, self.d_ll)
        finally:
            self.close()
        return out

    def close(self):
        ctx = getattr(self, "ctx", None)
        if ctx is None:
            return
        self.ctx = None
        try:
            for name in ("d_ts", "d_ys", "d_ll"):
                p = getattr(self, name, None)
                setattr(self, name, None)
                if p:
                    ctx.free(p)
        finally:
            ctx.close()


class _Packed(tuple):
    """(lam, N1, N2, Pinf, H, d) of Series.pack with the arrays' ctypes pointers made once (`.ptrs`: four data_as() calls are
    3 us of a 36 us evaluation)."""

    def __new__(cls, items, ptrs=None):
        self = super().__new__(cls, items)
        self.ptrs = tuple(_ptr(a) for a in items[1:5]) if ptrs is None else ptrs
        return self


class PackBuffer:
    """One model's (N1, N2, Pinf, H) in a buffer that lives as long as the model: a new hyper-parameter setting copies 30
    doubles into it and reuses the four ctypes pointers (making them is ~10 us of a 40 us evaluation at a new setting).
    The tuple `pack()` returns is valid until the next `pack()`: the series calls are synchronous, nothing keeps an older one."""

    def __init__(self, d):
        self.d = d
        self.buf = np.zeros(3 * d * d + d, np.float64)
        dd = d * d
        self.views = (self.buf[0:dd].reshape(d, d), self.buf[dd:2 * dd].reshape(d, d), self.buf[2 * dd:3 * dd].reshape(d, d),
                      self.buf[3 * dd:3 * dd + d])
        base = self.buf.ctypes.data
        self.ptrs = (c_void_p(base), c_void_p(base + 8 * dd), c_void_p(base + 16 * dd), c_void_p(base + 24 * dd))

    def pack(self, form, Pinf, H):
        lam, N1, N2 = form
        v = self.views
        v[0][...] = N1
        v[1][...] = N2
        v[2][...] = Pinf
        v[3][...] = np.asarray(H).reshape(-1)
        return _Packed((float(lam), v[0], v[1], v[2], v[3], self.d), self.ptrs)


class Series:
    """A series kept on the device across calls (pgps_series_*, include/pgps.h): what an optimiser or sampler loop
    evaluates thousands of times is ONE (ts, ys) at changing hyper-parameters, and predict_f on a fixed grid.  The
    handle holds ts, ys and -- once set -- the query grid merged with them; a call sends the fused model's scalars and
    returns the log-likelihood (+ gradient), or the K posterior means and variances.  fp64, Matern-family models."""

    def __init__(self, ts, ys, t0=0.0, device=0):
        self.ctx = get_context(device)
        ts_a = _prep(ts, np.float64, (-1,))
        ys_a = _prep(ys, np.float64, (-1,))
        if ys_a.shape[0] != ts_a.shape[0]:
            raise ValueError(f"observations has {ys_a.shape[0]} rows, the series {ts_a.shape[0]} steps")
        lib = self.ctx.lib
        if not hasattr(lib, "pgps_series_create_f64"):
            raise RuntimeError("this libpgps has no pgps_series_* entry points")
        P = c_void_p
        lib.pgps_series_create_f64.argtypes = [P, c_long, P, P, c_double, ctypes.POINTER(P)]
        lib.pgps_series_set_queries_f64.argtypes = [P, c_long, P]
        lib.pgps_series_destroy.argtypes = [P]
        lib.pgps_series_gp_ll_f64.argtypes = [P, c_int, c_double, P, P, P, P, c_double, P]
        lib.pgps_series_gp_ll_grad_f64.argtypes = [P, c_int, c_int, P, P]
        lib.pgps_series_gp_predict_f64.argtypes = [P, c_int, c_double, P, P, P, P, c_double, P, P, P]
        self.has_lti = hasattr(lib, "pgps_series_lti_ll_f64")
        if self.has_lti:
            lib.pgps_series_lti_ll_f64.argtypes = [P, c_int, P, P, P, c_double, P]
            lib.pgps_series_lti_predict_f64.argtypes = [P, c_int, P, P, P, c_double, P, P, P]
            lib.pgps_series_lti_ll_batch_f64.argtypes = [P, c_int, c_int, P, P]
        self.has_lti_grad = hasattr(lib, "pgps_series_lti_ll_grad_f64")
        if self.has_lti_grad:
            lib.pgps_series_lti_ll_grad_f64.argtypes = [P, c_int, P, P, P, c_double, P]
        self.has_gp_adj = hasattr(lib, "pgps_series_gp_ll_grad_adj_f64")
        if self.has_gp_adj:
            lib.pgps_series_gp_ll_grad_adj_f64.argtypes = [P, c_int, c_double, P, P, P, P, c_double, P]
        self.N, self.K = ts_a.shape[0], 0
        self._tq = None
        h = P()
        with self.ctx.lock:
            check(self.ctx, lib.pgps_series_create_f64(self.ctx.handle, c_long(self.N), _ptr(ts_a), _ptr(ys_a),
                                                       c_double(float(t0)), ctypes.byref(h)), "pgps_series_create_f64")
        self.handle = h
        self._ll = c_double(0.0)
        self._llp = ctypes.cast(ctypes.byref(self._ll), P)
        self._gout = np.zeros(32, np.float64)
        self._goutp = _ptr(self._gout)
        self._lti_slots = {}

    def set_queries(self, tq):
        """The (sorted) query times of predict(); merged with the series on the device once."""
        tq_a = _prep(tq, np.float64, (-1,))
        if self._tq is not None and self._tq.shape == tq_a.shape and np.array_equal(self._tq, tq_a):
            return
        with self.ctx.lock:
            check(self.ctx, self.ctx.lib.pgps_series_set_queries_f64(self.handle, c_long(tq_a.shape[0]), _ptr(tq_a)),
                  "pgps_series_set_queries_f64")
        self._tq, self.K = tq_a.copy(), tq_a.shape[0]
        self._mean, self._var = np.empty(self.K, np.float64), np.empty(self.K, np.float64)
        self._meanp, self._varp = _ptr(self._mean), _ptr(self._var)

    @staticmethod
    def pack(form, Pinf, H):
        """The fused model as the contiguous float64 arrays the calls take: (lam, N1, N2, Pinf, H, d)."""
        lam, N1, N2 = form
        d = N1.shape[0]
        c = lambda a, shape: a if (type(a) is np.ndarray and a.dtype == np.float64 and a.flags.c_contiguous and a.shape == shape) \
            else _prep(a, np.float64, shape)
        Hv = np.asarray(H, np.float64).reshape(-1)
        return _Packed((float(lam), c(N1, (d, d)), c(N2, (d, d)), c(Pinf, (d, d)), c(Hv, (d,)), d))

    def gp_ll(self, packed, R):
        lam, N1, N2, Pinf, H, d = packed
        p1, p2, p3, p4 = packed.ptrs if hasattr(packed, "ptrs") else (_ptr(N1), _ptr(N2), _ptr(Pinf), _ptr(H))
        with self.ctx.lock:
            check(self.ctx, self.ctx.lib.pgps_series_gp_ll_f64(self.handle, d, lam, p1, p2, p3, p4,
                                                               float(R), self._llp), "pgps_series_gp_ll_f64")
        return self._ll.value

    # -- any kernel's LTI model (F, Pinf, H), fp64, 2 <= d <= 32: pgps_series_lti_* ------------------------------
    def _lti_slot(self, F, Pinf, H):
        """(d, pF, pPinf, pH, out, pout): the model copied into this handle's buffer for its state dimension -- the ctypes
        pointers of a dimension are made once (three data_as() calls and an output array per evaluation were ~10 us)."""
        F = np.asarray(F)
        d = F.shape[0]
        slot = self._lti_slots.get(d)
        if slot is None:
            if F.ndim != 2 or F.shape[1] != d or not (LTI_DIM_MIN <= d <= LTI_DIM_MAX):
                raise ValueError(f"the general-LTI device path covers state dimensions {LTI_DIM_MIN}..{LTI_DIM_MAX}, got {F.shape}")
            dd = d * d
            buf = np.zeros(2 * dd + d, np.float64)
            out = np.zeros(2 + dd + 2 * d, np.float64)
            base = buf.ctypes.data
            slot = self._lti_slots[d] = (buf, buf[0:dd].reshape(d, d), buf[dd:2 * dd].reshape(d, d), buf[2 * dd:2 * dd + d],
                                         c_void_p(base), c_void_p(base + 8 * dd), c_void_p(base + 16 * dd), out, _ptr(out))
        slot[1][...] = F
        slot[2][...] = Pinf
        slot[3][...] = np.asarray(H).reshape(-1)
        return d, slot[4], slot[5], slot[6], slot[7], slot[8]

    def lti_ll(self, F, Pinf, H, R):
        d, pF, pP, pH, _, _ = self._lti_slot(F, Pinf, H)
        with self.ctx.lock:
            check(self.ctx, self.ctx.lib.pgps_series_lti_ll_f64(self.handle, d, pF, pP, pH, float(R), self._llp),
                  "pgps_series_lti_ll_f64")
        return self._ll.value

    def lti_ll_grad(self, F, Pinf, H, R):
        """(ll, Abar, Ubar, Hbar, Rbar): the log-likelihood and the model's adjoints (pgps_series_lti_ll_grad_f64)."""
        d, pF, pP, pH, out, pout = self._lti_slot(F, Pinf, H)
        with self.ctx.lock:
            check(self.ctx, self.ctx.lib.pgps_series_lti_ll_grad_f64(self.handle, d, pF, pP, pH, float(R), pout),
                  "pgps_series_lti_ll_grad_f64")
        return split_grad_stats(out, d)

    def lti_predict(self, F, Pinf, H, R):
        """(mean (K,), var (K,), ll) at the query grid of set_queries()."""
        d, pF, pP, pH, _, _ = self._lti_slot(F, Pinf, H)
        with self.ctx.lock:
            check(self.ctx, self.ctx.lib.pgps_series_lti_predict_f64(self.handle, d, pF, pP, pH, float(R),
                                                                    self._meanp, self._varp, self._llp),
                  "pgps_series_lti_predict_f64")
        return self._mean.copy(), self._var.copy(), self._ll.value

    def lti_ll_batch(self, models):
        """B log-likelihoods: models = [(F, Pinf, H, R)] of one state dimension (2 <= d <= 16)."""
        table, d = _lti_table(models)
        out = np.empty(table.shape[0], np.float64)
        with self.ctx.lock:
            check(self.ctx, self.ctx.lib.pgps_series_lti_ll_batch_f64(self.handle, table.shape[0], d, _ptr(table), _ptr(out)),
                  "pgps_series_lti_ll_batch_f64")
        return out

    def gp_ll_grad(self, model, d, npar):
        with self.ctx.lock:
            check(self.ctx, self.ctx.lib.pgps_series_gp_ll_grad_f64(self.handle, d, npar, _ptr(model), _ptr(self._gout)),
                  "pgps_series_gp_ll_grad_f64")
        return float(self._gout[0]), self._gout[1:1 + npar].copy()

    def gp_ll_grad_adj(self, packed, R):
        """(ll, Abar, Ubar, Hbar, Rbar) of the fused (Matern-family) model by the adjoint pass on the lane-chunk kernels
        (pgps_series_gp_ll_grad_adj_f64): the same statistics as lti_ll_grad(), contracted by contract_grad_stats()."""
        lam, N1, N2, Pinf, H, d = packed
        with self.ctx.lock:
            check(self.ctx, self.ctx.lib.pgps_series_gp_ll_grad_adj_f64(self.handle, d, lam, _ptr(N1), _ptr(N2), _ptr(Pinf), _ptr(H),
                                                                        float(R), _ptr(self._gout)), "pgps_series_gp_ll_grad_adj_f64")
        return split_grad_stats(self._gout[:2 + d * d + 2 * d].copy(), d)

    def gp_ll_grad_adj_raw(self, packed, R):
        """The same call, results left in the handle's output buffer [ll | Abar | Ubar | Hbar | Rbar] (a view: valid until
        the next call) -- for callers that contract them in place."""
        lam, N1, N2, Pinf, H, d = packed
        p1, p2, p3, p4 = packed.ptrs if hasattr(packed, "ptrs") else (_ptr(N1), _ptr(N2), _ptr(Pinf), _ptr(H))
        with self.ctx.lock:
            check(self.ctx, self.ctx.lib.pgps_series_gp_ll_grad_adj_f64(self.handle, d, lam, p1, p2, p3, p4,
                                                                        float(R), self._goutp), "pgps_series_gp_ll_grad_adj_f64")
        return self._gout

    def gp_predict(self, packed, R):
        """(mean (K,), var (K,), ll) at the query grid of set_queries()."""
        lam, N1, N2, Pinf, H, d = packed
        p1, p2, p3, p4 = packed.ptrs if hasattr(packed, "ptrs") else (_ptr(N1), _ptr(N2), _ptr(Pinf), _ptr(H))
        mean, var = np.empty(self.K, np.float64), np.empty(self.K, np.float64)
        with self.ctx.lock:
            check(self.ctx, self.ctx.lib.pgps_series_gp_predict_f64(self.handle, d, lam, p1, p2, p3,
                                                                    p4, float(R), _ptr(mean), _ptr(var), self._llp),
                  "pgps_series_gp_predict_f64")
        return mean, var, self._ll.value

    def close(self):
        h, self.handle = getattr(self, "handle", None), None
        if h and self.ctx is not None and getattr(self.ctx.handle, "value", None):
            with self.ctx.lock:
                self.ctx.lib.pgps_series_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:       # noqa: BLE001 -- interpreter shutdown
            pass


def _lti_table(models):
    """(B x [F | Pinf | H | R] float64 table, d) of a list of (F, Pinf, H, R), all of one state dimension."""
    d, table = None, None
    for b, (F, Pinf, H, R) in enumerate(models):
        F, Pinf, H, dm = _lti_model(F, Pinf, H)
        if dm > LTI_BATCH_DIM_MAX:
            raise ValueError(f"batched general-LTI evaluation covers state dimensions up to {LTI_BATCH_DIM_MAX}, got {dm}")
        if d is None:
            d = dm
            table = np.empty((len(models), 2 * d * d + d + 1), np.float64)
        if dm != d:
            raise ValueError("all models of a batch must have the same state dimension")
        row = table[b]
        row[:d * d] = F.reshape(-1)
        row[d * d:2 * d * d] = Pinf.reshape(-1)
        row[2 * d * d:2 * d * d + d] = H.reshape(-1)
        row[-1] = R
    return table, d


def lti_ll_batch(models, ts, ys, t0=0.0, device=0):
    """Log-likelihoods of B general LTI models over one series in one set of launches (pgps_lti_ll_batch_f64).

    `models`: list of (F, Pinf, H, R) -- what lti_ll() takes, once per model; all of one state dimension."""
    packed, d = _lti_table(models)
    ts_a = _prep(ts, np.float64, (-1,))
    ys_a = _prep(ys, np.float64, (-1,))
    if ys_a.shape[0] != ts_a.shape[0]:
        raise ValueError(f"observations has {ys_a.shape[0]} rows, the series {ts_a.shape[0]} steps")
    out = np.zeros(len(models), np.float64)
    get_context(device).call("pgps_lti_ll_batch_f64", c_int(len(models)), c_long(ts_a.shape[0]), c_int(d), _ptr(packed),
                             _ptr(ts_a), _ptr(ys_a), c_double(float(t0)), _ptr(out))
    return out


def gp_ll_batch(models, ts, ys, t0=0.0, device=0):
    """Log-likelihoods of B models over one series (pgps_gp_ll_batch_*).

    `models`: list of (form=(lam, N1, N2), Pinf, H, R) -- what gp() takes, once per model."""
    ts_a = np.asarray(ts)
    dtype = ts_a.dtype if ts_a.dtype in (np.float32, np.float64) else np.dtype(np.float64)
    suf, _ = _suffix(dtype)
    ts_a = _prep(ts_a, dtype, (-1,))
    ys_a = _prep(ys, dtype, (-1,))
    if ys_a.shape[0] != ts_a.shape[0]:
        raise ValueError(f"observations has {ys_a.shape[0]} rows, the series {ts_a.shape[0]} steps")
    d = np.asarray(models[0][0][1]).shape[0]
    rows = []
    for (lam, N1, N2), Pinf, H, R in models:
        if np.asarray(N1).shape[0] != d:
            raise ValueError("all models of a batch must have the same state dimension")
        rows.append(np.concatenate([[float(lam)], np.asarray(N1, np.float64).reshape(-1), np.asarray(N2, np.float64).reshape(-1),
                                    np.asarray(Pinf, np.float64).reshape(-1), np.asarray(H, np.float64).reshape(-1),
                                    [float(R)]]))
    packed = np.ascontiguousarray(np.stack(rows), dtype=np.float64)
    out = np.zeros(len(models), np.float64)
    get_context(device).call(f"pgps_gp_ll_batch_{suf}", c_int(len(models)), c_long(ts_a.shape[0]), c_int(d), _ptr(packed),
                             _ptr(ts_a), c_double(float(t0)), _ptr(ys_a), _ptr(out))
    return out


def pack_grad_model(blocks):
    """(1 + np, 1 + 2 d^2 + d + 1) array [lam | N1 | Pinf | H | R] per block, as pgps_gp_ll_grad_* reads it."""
    d = np.asarray(blocks[0][1]).shape[0]
    dd = d * d
    model = np.empty((len(blocks), 2 + 2 * dd + d), np.float64)
    for r, (lam, N1, Pinf, H, R) in enumerate(blocks):
        row = model[r]
        row[0] = lam
        row[1:1 + dd] = np.asarray(N1, np.float64).reshape(-1)          # (a shape mismatch raises here)
        row[1 + dd:1 + 2 * dd] = np.asarray(Pinf, np.float64).reshape(-1)
        row[1 + 2 * dd:1 + 2 * dd + d] = np.asarray(H, np.float64).reshape(-1)
        row[-1] = R
    return model, d, len(blocks) - 1


GRAD_BLOCKS_DIM_MAX = 6        # pgps_gp_ll_grad_blocks_*: state dimension, blocks
GRAD_BLOCKS_MAX = 4


def nilpotent_blocks(F, tol=1e-9):
    """[(start, size, lam, N)] when F is block diagonal (contiguous blocks) with every block -lam I + N, N nilpotent --
    sums and products of Matern kernels, balanced or not -- else None.  Blocks = connected components of F's pattern."""
    F = np.asarray(F, np.float64)
    d = F.shape[0]
    scale = max(1.0, float(np.max(np.abs(F))))
    adj = (np.abs(F) > tol * scale) | (np.abs(F.T) > tol * scale) | np.eye(d, dtype=bool)
    comp = -np.ones(d, int)
    ncomp = 0
    for i in range(d):
        if comp[i] >= 0:
            continue
        stack, comp[i] = [i], ncomp
        while stack:
            u = stack.pop()
            for v in np.nonzero(adj[u])[0]:
                if comp[v] < 0:
                    comp[v] = ncomp
                    stack.append(v)
        ncomp += 1
    if np.any(np.diff(comp) < 0) or np.any(np.diff(comp) > 1):
        return None                                 # components must be contiguous index ranges, in order
    blocks = []
    for b in range(ncomp):
        idx = np.nonzero(comp == b)[0]
        lo, n = int(idx[0]), int(idx.size)
        Fb = F[lo:lo + n, lo:lo + n]
        lam = -float(np.trace(Fb)) / n              # all eigenvalues of a Matern block equal -lam (companion form)
        Nb = Fb + lam * np.eye(n)
        if not np.all(np.isfinite(Nb)) or np.max(np.abs(np.linalg.matrix_power(Nb, 4))) > tol * scale ** 4:
            return None                             # not nilpotent, or beyond the device series (it stops at N^3 / 6)
        blocks.append((lo, n, lam, Nb))
    return blocks


def gp_ll_grad_blocks(rows, bsizes, ts, ys, t0=0.0, device=0):
    """Log-likelihood and its exact gradient for a block-nilpotent composite model (pgps_gp_ll_grad_blocks_f64, fp64,
    2 <= d <= 6).  `rows`: 1 + np tuples (lams [nblk], N (d, d), Pinf (d, d), H (d,), R) -- the model, then its partial
    derivative with respect to each hyper-parameter; `bsizes`: the block sizes.  Returns (ll, grad[np])."""
    d = int(np.asarray(rows[0][1]).shape[0])
    nblk = len(bsizes)
    packed = []
    for lams, Nm, Pinf, H, R in rows:
        lam4 = np.zeros(GRAD_BLOCKS_MAX)
        lam4[:nblk] = np.asarray(lams, np.float64).reshape(-1)
        packed.append(np.concatenate([lam4, np.asarray(Nm, np.float64).reshape(-1), np.asarray(Pinf, np.float64).reshape(-1),
                                      np.asarray(H, np.float64).reshape(-1), [float(R)]]))
    model = np.ascontiguousarray(np.stack(packed), dtype=np.float64)
    npar = len(rows) - 1
    ts_a = _prep(ts, np.float64, (-1,))
    ys_a = _prep(ys, np.float64, (-1,))
    if ys_a.shape[0] != ts_a.shape[0]:
        raise ValueError(f"observations has {ys_a.shape[0]} rows, the series {ts_a.shape[0]} steps")
    bs = np.ascontiguousarray(bsizes, dtype=np.int32)
    out = np.zeros(1 + npar, np.float64)
    get_context(device).call("pgps_gp_ll_grad_blocks_f64", c_long(ts_a.shape[0]), c_int(d), c_int(nblk), _ptr(bs), c_int(npar),
                             _ptr(model), _ptr(ts_a), c_double(float(t0)), _ptr(ys_a), _ptr(out))
    return out[0], out[1:]


def gp_ll_grad(blocks, ts, ys, t0=0.0, device=0):
    """Log-likelihood and its gradient on the fused path (pgps_gp_ll_grad_f64, fp64, d <= 3).

    `blocks`: list of 1 + np tuples (lam, N1, Pinf, H, R) -- the model, then its partial derivative
    with respect to each hyper-parameter.  Returns (ll, grad[np])."""
    model, d, npar = pack_grad_model(blocks)
    ts_a = _prep(ts, np.float64, (-1,))
    ys_a = _prep(ys, np.float64, (-1,))
    if ys_a.shape[0] != ts_a.shape[0]:
        raise ValueError(f"observations has {ys_a.shape[0]} rows, the series {ts_a.shape[0]} steps")
    out = np.zeros(1 + npar, np.float64)
    get_context(device).call("pgps_gp_ll_grad_f64", c_long(ts_a.shape[0]), c_int(d), c_int(npar), _ptr(model),
                             _ptr(ts_a), c_double(float(t0)), _ptr(ys_a), _ptr(out))
    return out[0], out[1:]
