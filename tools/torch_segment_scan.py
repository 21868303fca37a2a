"""SegmentScan: the three-phase segment protocol of pssgp.distributed with torch.distributed's collectives in between.

TOOLING, not product: the product's multi-GPU path is pssgp.distributed.ShardedScan (the libpgps context owns the RCCL
communicator, one library call per pass, no framework).  This variant exists for dry runs where RCCL cannot be used --
several ranks sharing one GPU under gloo (tests/test_segments.py, `bench.py --all-on-gpu0`) -- and as bench.py's
explicitly flagged fallback (`"exchange_fallback": true` in its JSON line) when the in-library communicator cannot be built.
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "parallel-gps_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

from pssgp.distributed import record_lengths, run_protocol  # noqa: E402


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
