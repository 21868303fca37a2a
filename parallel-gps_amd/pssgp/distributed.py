"""One long series sharded over the GPUs of a node: contiguous time segments, one process per
GPU, stitched by two tiny all-gathers of segment totals (RCCL over xGMI).

The product path is `ShardedScan`: the libpgps context owns the RCCL communicator (`pgps_comm_init`) and one call
per pass (`pgps_pkfs_seg_dev_*`) enqueues reduce -> ncclAllGather -> filter -> ncclAllGather -> smoother on the
context's stream -- no framework, no host round trip.  Only the 128-byte communicator id has to reach every rank
once (`share_unique_id`: a file, or any broadcast the launcher offers).  (The older framework-hosted variant --
the same three library phases with torch.distributed's collectives in between -- is test / benchmark tooling and
lives outside the product, in tools/torch_segment_scan.py.)

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
import threading
import time

import numpy as np

# NOTE for hosts that also use PyTorch-ROCm: import torch BEFORE the first libpgps context is created (it ships its
# own HIP runtime and both libraries must share one copy of it in the process).  This module itself needs no framework.


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


_UID_MAGIC = b"PGPSUID2"


def _run_tag(run_id):
    """16 bytes that identify one launch: a file left behind by an earlier run at the same path is not this run's."""
    import hashlib
    if run_id is None:
        run_id = "|".join(os.environ.get(k, "") for k in ("TORCHELASTIC_RUN_ID", "MASTER_ADDR", "MASTER_PORT", "SLURM_JOB_ID"))
    return hashlib.sha256(str(run_id).encode()).digest()[:16]


def _unique_launch(run_id):
    """True when `run_id` (or the launcher's variables) actually distinguishes this launch from the one before it at the
    same path: plain `torchrun` gives TORCHELASTIC_RUN_ID = 'none' and the same address / port every time."""
    if run_id is not None:
        return True
    rid = os.environ.get("TORCHELASTIC_RUN_ID", "")
    return bool(os.environ.get("SLURM_JOB_ID")) or (rid not in ("", "none"))


def _read_small(path, want=None):
    try:
        with open(path, "rb") as fh:
            blob = fh.read()
        return blob if (want is None or len(blob) == want) else None
    except (FileNotFoundError, PermissionError):
        return None


def _write_atomic(path, blob):
    tmp = f"{path}.tmp.{os.getpid()}.{threading.get_ident()}"
    with open(tmp, "wb") as fh:
        fh.write(blob)
    os.replace(tmp, path)


def _handshake_files(path, world):
    return [f"{path}.hello.{r}" for r in range(1, world)] + [f"{path}.ack.{r}" for r in range(1, world)]


def share_unique_id(rank, path=None, broadcast=None, timeout=120.0, run_id=None, world=None, make_id=None):
    """The communicator id of rank 0 on every rank.  Either `broadcast(bytes_or_None) -> bytes` (whatever the launcher
    offers: MPI bcast, a torch.distributed / TCP store, ...) or a file all ranks can see (`path`).  Joining an id that is
    not this run's makes ncclCommInitRank hang, so a file is accepted only when it PROVABLY belongs to this launch -- never
    because it looks recent (file ages compare a shared file system's clock with the local one, and a relaunch right after a
    crash finds the dead run's file seconds old):

    * `run_id` given, or the launcher provides one that differs from run to run (SLURM_JOB_ID, a real TORCHELASTIC_RUN_ID):
      rank 0 writes the file atomically with a tag of that id, the others accept only a file carrying THIS run's tag;
    * otherwise (plain `torchrun`: TORCHELASTIC_RUN_ID = 'none', the same address and port every time) `world` must be
      given and the ranks shake hands through files next to `path`: every other rank draws a random token and leaves it in
      `<path>.hello.<rank>`; rank 0 -- which first removes whatever an earlier run left there -- publishes the id together
      with the tokens it has seen, a rank accepts the file only if it carries ITS token and acknowledges with
      `<path>.ack.<rank>`; rank 0 returns once every acknowledgement matches (re-publishing if a rank re-drew its token).
      A file of a dead run cannot carry a token drawn after it died.
    * neither: ValueError -- fail rather than join an id that cannot be verified.

    Rank 0 removes the files with `remove_unique_id_file(path)` once every rank has called pgps_comm_init.  `make_id`
    (tests) replaces pgps_comm_get_unique_id.  No GPU work."""
    from . import _backend
    uid = None
    if rank == 0:
        uid = make_id() if make_id is not None else _backend.Context.comm_unique_id()
    nid = _backend.COMM_ID_BYTES
    if broadcast is not None:
        uid = broadcast(uid)
    elif path is not None:
        deadline = time.monotonic() + timeout
        if _unique_launch(run_id):
            tag = _run_tag(run_id)
            want = len(_UID_MAGIC) + len(tag) + nid
            if rank == 0:
                _write_atomic(path, _UID_MAGIC + tag + uid)
            else:
                while True:
                    blob = _read_small(path, want)
                    if blob is not None and blob.startswith(_UID_MAGIC + tag):
                        uid = blob[len(_UID_MAGIC) + len(tag):]
                        break
                    if time.monotonic() > deadline:
                        raise TimeoutError(f"no communicator id of this run at {path} after {timeout} s")
                    time.sleep(0.01)
        elif world is not None and int(world) >= 1:
            world = int(world)
            magic = _UID_MAGIC + b"HANDSHAK"
            want = len(magic) + nid + 16 * (world - 1)
            if rank == 0:
                for f in [path] + _handshake_files(path, world):
                    try:
                        os.unlink(f)
                    except FileNotFoundError:
                        pass
                published = None
                while True:
                    toks = [_read_small(f"{path}.hello.{r}", 16) for r in range(1, world)]
                    if all(t is not None for t in toks) and toks != published:
                        _write_atomic(path, magic + uid + b"".join(toks))
                        published = toks
                    if published is not None and all(_read_small(f"{path}.ack.{r}", 16) == published[r - 1] for r in range(1, world)):
                        break
                    if time.monotonic() > deadline:
                        raise TimeoutError(f"ranks missing from the id handshake at {path} after {timeout} s")
                    time.sleep(0.01)
            else:
                token = os.urandom(16)
                hello, ack = f"{path}.hello.{rank}", f"{path}.ack.{rank}"
                while True:
                    if _read_small(hello, 16) != token:         # (rank 0 clears what it finds when it starts: say it again)
                        _write_atomic(hello, token)
                    blob = _read_small(path, want)
                    if blob is not None and blob.startswith(magic):
                        off = len(magic) + nid + 16 * (rank - 1)
                        if blob[off:off + 16] == token:
                            uid = blob[len(magic):len(magic) + nid]
                            _write_atomic(ack, token)
                            break
                    if time.monotonic() > deadline:
                        raise TimeoutError(f"no communicator id carrying this rank's token at {path} after {timeout} s")
                    time.sleep(0.01)
        else:
            raise ValueError("share_unique_id(path=...): this launch has no id that differs from run to run (plain torchrun) -- "
                             "pass run_id=<unique per launch>, or world=<ranks> for the token handshake, or broadcast=")
    elif rank != 0:
        raise ValueError("share_unique_id needs `path` or `broadcast` when there is more than one rank")
    return bytes(uid)


def remove_unique_id_file(path, world=None):
    """Rank 0, after every rank has joined the communicator: the id file (and the handshake's files) have served."""
    import glob as _glob
    files = [path] + (_handshake_files(path, int(world)) if world else _glob.glob(path + ".hello.*") + _glob.glob(path + ".ack.*"))
    for f in files:
        try:
            os.unlink(f)
        except FileNotFoundError:
            pass


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


def split_segments(n_total, world):
    """Contiguous [lo, hi) bounds per rank; the first n_total % world ranks get one extra step."""
    base, extra = divmod(int(n_total), int(world))
    bounds, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < extra else 0)
        bounds.append((lo, hi))
        lo = hi
    return bounds
