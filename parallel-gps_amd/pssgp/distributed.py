"""One long series sharded over the GPUs of a node: contiguous time segments, one process per
GPU, stitched by two tiny all-gathers of segment totals (RCCL over xGMI).

The product path is `ShardedScan`: the libpgps context owns the RCCL communicator (`pgps_comm_init`) and one call
per pass (`pgps_pkfs_seg_dev_*`) enqueues reduce -> ncclAllGather -> filter -> ncclAllGather -> smoother on the
context's stream -- no framework, no host round trip.  Only the 128-byte communicator id has to reach every rank
once (`share_unique_id`: a file, or any broadcast the launcher offers).  `SegmentScan` is the older framework-hosted
variant (the same three library phases with torch.distributed's collectives in between), kept for hosts that
already live inside torch.distributed and for the CPU / gloo tests of the protocol.

The reference has no multi-device path; this module is the MI355X
extension of `pkfs`.  A prefix scan over an associative operator splits into segments whose
only coupling is one total per segment:

    rank r:  k_filter_reduce  -> rec_f[r] = [ (A,b,C,J,eta) of the segment | F_0, Q_0 ]
             all_gather(rec_f)                                 # nranks x (3d^2+3d+d(d+1)) scalars
             fold totals of ranks < r -> carry-in (m, P);  F_0, Q_0 of rank r+1 -> halo
             k_filter_apply   -> fms, fPs, rec_s[r] = [ (E,g,L) of the segment | ll partial ]
             all_gather(rec_s)
             fold totals of ranks > r -> smoothed state at the first step of rank r+1
             k_smoother_apply -> sms, sPs;  ll = sum of the partials

Payloads are a few hundred bytes: the exchange is latency-bound, not xGMI-bandwidth-bound,
which is why it is an all-gather (one hop on the fully connected mesh) and not a ring reduce.

`run_protocol` is the same protocol with the three phases and the gather injected, so it can be driven by
several contexts on one GPU or by CPU stand-ins under gloo in the tests.
"""
import ctypes
import os
import time

import numpy as np

# NOTE: import this module (or torch) BEFORE the first libpgps context is created: PyTorch-ROCm ships
# its own HIP runtime and both libraries must share one copy of it in the process.
try:
    import torch as _torch  # noqa: F401
except Exception:  # torch is only needed by SegmentScan
    _torch = None


def record_lengths(d):
    """(rec_filter, rec_smoother, smoother pad) in scalars -- must match pgps_seg_record_len."""
    sym = d * (d + 1) // 2
    nfilt = d * d + d + 2 * sym + d
    nsmth = d * d + d + sym
    pad = nsmth + (nsmth & 1)
    return nfilt + 2 * d * d, pad + 2, pad


def run_protocol(rank, nranks, phase_reduce, phase_filter, phase_smoother, all_gather):
    """The three-phase segment protocol for one rank.

    phase_reduce()            -> rec_f                      (this rank's filter record)
    phase_filter(gathered_f)  -> rec_s                      (writes fms, fPs as a side effect)
    phase_smoother(gathered_s)-> result                     (writes sms, sPs, ll)
    all_gather(rec)           -> (nranks, len(rec)) array/tensor of every rank's record
    """
    del rank, nranks
    gathered_f = all_gather(phase_reduce())
    gathered_s = all_gather(phase_filter(gathered_f))
    return phase_smoother(gathered_s)


def share_unique_id(rank, path=None, broadcast=None, timeout=120.0):
    """The communicator id of rank 0 on every rank.  Either `broadcast(bytes_or_None) -> bytes` (whatever the launcher
    offers: MPI bcast, a torch.distributed / TCP store, ...) or a file all ranks can see (`path`, written atomically
    by rank 0, polled by the others).  No GPU work."""
    from . import _backend
    uid = _backend.Context.comm_unique_id() if rank == 0 else None
    if broadcast is not None:
        uid = broadcast(uid)
    elif path is not None:
        if rank == 0:
            tmp = f"{path}.tmp.{os.getpid()}"
            with open(tmp, "wb") as fh:
                fh.write(uid)
            os.replace(tmp, path)
        else:
            deadline = time.monotonic() + timeout
            while True:
                try:
                    with open(path, "rb") as fh:
                        uid = fh.read()
                    if len(uid) == _backend.COMM_ID_BYTES:
                        break
                except FileNotFoundError:
                    pass
                if time.monotonic() > deadline:
                    raise TimeoutError(f"no communicator id at {path} after {timeout} s")
                time.sleep(0.01)
    elif rank != 0:
        raise ValueError("share_unique_id needs `path` or `broadcast` when there is more than one rank")
    return bytes(uid)


class ShardedScan:
    """pkfs + log-likelihood for this rank's contiguous segment of a series sharded over `world` GPUs, with the
    exchange inside libpgps (RCCL communicator owned by the context).  Device buffers are plain device pointers
    (integers): from Context.malloc, or the data_ptr() of any device-array library."""

    def __init__(self, ctx, unique_id, rank, world, d, dtype):
        self.ctx, self.rank, self.world, self.d = ctx, int(rank), int(world), int(d)
        self.suf = "f64" if np.dtype(dtype) == np.float64 else "f32"
        self.real = ctypes.c_double if self.suf == "f64" else ctypes.c_float
        ctx.comm_init(unique_id, rank, world)

    def pkfs(self, n_local, P0, Fs, Qs, H, R, ys, fms, fPs, sms, sPs, ll):
        """Asynchronous on the context's stream; every argument but n_local and R is a device pointer."""
        P = ctypes.c_void_p
        self.ctx.call(f"pgps_pkfs_seg_dev_{self.suf}", ctypes.c_long(n_local), ctypes.c_int(self.d), P(P0), P(Fs), P(Qs),
                      P(H), self.real(R), P(ys), P(fms), P(fPs), P(sms), P(sPs), P(ll))

    def close(self):
        self.ctx.comm_destroy()


class SegmentScan:
    """pkfs for the segment of `rank`; device tensors are torch tensors on `torch_device`.  The collectives are
    torch.distributed's, so libpgps must launch on the stream they are enqueued on: the constructor binds the
    context to torch's current stream of `torch_device` (a context's own stream is a private non-blocking one, and
    nothing would order the kernels that write the records against the all-gathers otherwise)."""

    def __init__(self, ctx, rank, world, d, dtype, torch_device=None, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.ctx, self.rank, self.world, self.d = ctx, int(rank), int(world), int(d)
        self.suf = "f64" if np.dtype(dtype) == np.float64 else "f32"
        self.real = ctypes.c_double if self.suf == "f64" else ctypes.c_float
        tdtype = torch.float64 if self.suf == "f64" else torch.float32
        rf, rs, _ = record_lengths(d)
        lf, ls = ctypes.c_int(0), ctypes.c_int(0)
        code = ctx.lib.pgps_seg_record_len(ctypes.c_int(d), ctypes.byref(lf), ctypes.byref(ls))
        assert code == 0 and (lf.value, ls.value) == (rf, rs), "record layout mismatch with libpgps"
        self.group = group
        if torch_device is not None and torch.device(torch_device).type == "cuda":
            ctx.set_stream(torch.cuda.current_stream(torch_device).cuda_stream)
        kw = dict(dtype=tdtype, device=torch_device)
        self.rec_f = torch.zeros(rf, **kw)
        self.rec_s = torch.zeros(rs, **kw)
        self.gathered_f = torch.zeros((world, rf), **kw)
        self.gathered_s = torch.zeros((world, rs), **kw)

    def _gather(self, out, rec):
        if self.world == 1:
            out.copy_(rec.view(1, -1))
        elif self.dist.get_backend(self.group) == "nccl":
            self.dist.all_gather_into_tensor(out, rec, group=self.group)       # RCCL over xGMI
        else:
            # e.g. gloo (tests: several ranks sharing one GPU): list form, staged through the host
            parts = [self.torch.empty_like(rec) for _ in range(self.world)]
            self.dist.all_gather(parts, rec, group=self.group)
            out.copy_(self.torch.stack(parts))
        return out

    def pkfs(self, n_local, P0, Fs, Qs, H, R, ys, fms, fPs, sms, sPs, ll):
        """All arguments are device tensors of this rank's segment (ll: float64[>=1])."""
        P = lambda t: ctypes.c_void_p(t.data_ptr())
        c, suf = self.ctx, self.suf
        N, d, r, w = ctypes.c_long(n_local), ctypes.c_int(self.d), ctypes.c_int(self.rank), ctypes.c_int(self.world)

        def phase_reduce():
            c.call(f"pgps_seg_filter_reduce_dev_{suf}", N, d, r, w, P(P0), P(Fs), P(Qs), P(H), self.real(R), P(ys),
                   P(self.rec_f))
            return self.rec_f

        def phase_filter(gathered_f):
            c.call(f"pgps_seg_filter_apply_dev_{suf}", N, d, r, w, P(P0), P(Fs), P(Qs), P(H), self.real(R), P(ys),
                   P(gathered_f), P(fms), P(fPs), P(self.rec_s))
            return self.rec_s

        def phase_smoother(gathered_s):
            c.call(f"pgps_seg_smoother_apply_dev_{suf}", N, d, r, w, P(Fs), P(Qs), P(fms), P(fPs), P(gathered_s),
                   P(sms), P(sPs), P(ll))
            return ll

        gathers = iter((self.gathered_f, self.gathered_s))
        with c.lock:        # the three phases share the context's scratch: nothing else may use it in between
            return run_protocol(self.rank, self.world, phase_reduce, phase_filter, phase_smoother,
                                lambda rec: self._gather(next(gathers), rec))


def split_segments(n_total, world):
    """Contiguous [lo, hi) bounds per rank; the first n_total % world ranks get one extra step."""
    base, extra = divmod(int(n_total), int(world))
    bounds, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < extra else 0)
        bounds.append((lo, hi))
        lo = hi
    return bounds
