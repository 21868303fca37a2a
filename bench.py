#!/usr/bin/env python3
"""bench.py -- timesteps/s of filter + smoother + log-likelihood (BASELINE.json's metric).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one series already resident in HBM: `pgps_pkfs_dev_f64`, producing the
filtered and smoothed moments of every time step and the log-likelihood -- at d = 2 fp64 up to 2^20 steps ONE resident
launch (k_pkfs_resident: Fs, Qs, ys read once, every output written once), otherwise k_filter_reduce + k_filter_apply +
k_smoother_apply.

Workload at N = 1: BASELINE.json configs[1] (c2) -- Matern-3/2 (state dim 2), 2^20 steps, fp64,
irregular times, observations drawn from the model's own prior.
At N > 1: BASELINE.json configs[3] (c4) -- the same model over ONE series of 2^24 steps split into N
contiguous segments, rank r owning [r * 2^24 / N, (r+1) * 2^24 / N) ("strong" scaling; `--scaling weak`
keeps 2^20 steps per GPU instead).  The segments are stitched by two all-gathers of segment totals
over RCCL, issued by libpgps itself on the context's stream (pgps_pkfs_seg_dev_f64; `--exchange torch`
runs the older torch.distributed-hosted variant).

`python bench.py --gpus N` works unaided: without a launcher's WORLD_SIZE in the environment the parent
-- before it imports torch or touches the GPU -- starts `python -m torch.distributed.run` with N fresh
child processes and relays rank 0's JSON line.

PyTorch is plumbing here: device buffers, the stream handed to libpgps, and (gloo, CPU) the rendezvous,
the barrier and the max-over-ranks of the timing.
"""
import argparse
import ctypes
import glob
import json
import os
import socket
import subprocess
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # RCCL across processes: dmabuf IPC only on this driver

# one host thread is all this benchmark needs: multi-threaded BLAS in the (untimed) data generation
# burns the container's CPU quota and the throttling then lands inside the timed region
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
    os.environ.setdefault(_v, "1")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "parallel-gps_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
# vector (non-MFMA) FMA peaks, SURVEY.md 8(d): the path has no dense contraction, the matrix cores never apply
VECTOR_PEAK_TFLOPS = {"f64": 78.6, "f32": 157.3}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--log2n", type=int, default=0,
                    help="log2 of the series length: of the WHOLE series with --scaling strong (default 24 = config c4), "
                         "of the steps per GPU with --scaling weak and at one GPU (default 20 = config c2)")
    ap.add_argument("--scaling", default="auto", choices=["auto", "strong", "weak"],
                    help="auto = strong (one 2^24-step series split over the GPUs) when --gpus > 1")
    ap.add_argument("--exchange", default="auto", choices=["auto", "lib", "torch"],
                    help="lib: RCCL communicator owned by the libpgps context, one call per pass (default); torch: the "
                         "three library phases with torch.distributed collectives in between (--dist-backend)")
    ap.add_argument("--kernel", default="matern32",
                    choices=["matern32", "matern52", "matern12", "c5", "periodic10", "co2"] + [f"rbf{n}" for n in range(2, 33)],
                    help="rbfN = RBF of order N (state dimension N)")
    ap.add_argument("--family", type=int, default=0,
                    help="0 auto, 1 lane-chunk, 2 wave-cooperative, 3 row-cooperative, 4 quad-cooperative (fp32, d = 5..8) kernels")
    ap.add_argument("--rc-scan", type=int, default=-1, choices=[-1, 0, 1],
                    help="scans of the chain totals (families 3, 4): -1 auto, 0 one launch per Kogge-Stone level, 1 blocked")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--block", type=int, default=0, choices=[0, 128, 256],
                    help="lanes per workgroup of the lane-chunk kernels: 0 = library default (pgps_set_block)")
    ap.add_argument("--dma", type=int, default=-1, choices=[-1, 0, 1],
                    help="LDS-DMA ring in the Kalman pass (d = 2 fp64): -1 = library default, 0 off, 1 on (pgps_set_dma)")
    ap.add_argument("--chunk", type=int, default=0, help="steps per lane (0 = library default)")
    ap.add_argument("--stage", type=int, default=-1, help="LDS staging: -1 auto, 0 off, 2 / 4 steps per sub-tile")
    ap.add_argument("--path", default="lgssm", choices=["lgssm", "fused", "fused-ll"],
                    help="lgssm: pkfs on resident Fs/Qs/ys (the reference's pkf/pks contract); fused: pgps_gp on "
                         "resident ts/ys (discretisation inside the scan); fused-ll: log-likelihood only")
    ap.add_argument("--resident", type=int, default=-1, choices=[-1, 0, 1],
                    help="filter + smoother in one resident launch (d = 2 fp64, up to 4096 steps per CU): -1 library default "
                         "(on from 2^18 steps), 0 never (three launches), 1 wherever it fits (pgps_set_resident)")
    ap.add_argument("--rewarm-ms", type=float, default=60.0,
                    help="back-to-back untimed passes for at least this long right before the timed region (clocks at load)")
    ap.add_argument("--single-pass", type=int, default=-1,
                    help="single-pass (look-back) filter kernel: -1 auto, 0 off (three launches), 1 on")
    ap.add_argument("--dist-backend", default="nccl",
                    help="--exchange torch only: torch.distributed backend of the collectives (nccl = RCCL; gloo for dry runs)")
    ap.add_argument("--all-on-gpu0", action="store_true",
                    help="dry run of the multi-rank path on a 1-GPU box: every rank uses GPU 0 (RCCL refuses two ranks on "
                         "one device, so this implies --exchange torch --dist-backend gloo)")
    ap.add_argument("--force-segments", action="store_true",
                    help="run the multi-GPU segment protocol even at one GPU (measures its overhead)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--main-only", action="store_true",
                    help="skip the fused / LTI side legs (profiling runs: only the measured path's kernels are launched)")
    ap.add_argument("--event-every", type=int, default=8,
                    help="hipEvent-time every n-th launch of the dominant kernel inside the timed region")
    ap.add_argument("--nan-frac", type=float, default=0.0)
    ap.add_argument("--grid", default="baseline", choices=["baseline", "reference"],
                    help="time stamps: BASELINE's irregular steps of ~0.05 (default) or the reference's own benchmark grid "
                         "np.linspace(0, 4, N) (toy_models/common.py:31-32) -- dense: float32 smoother calls are promoted there")
    ap.add_argument("--f32-policy", type=int, default=0, choices=[0, 1, 2],
                    help="float32 series: 0 automatic promotion to fp64 arithmetic on dense grids, 1 never, 2 always "
                         "(pgps_set_f32_policy)")
    args = ap.parse_args()
    if args.scaling == "auto":
        args.scaling = "strong" if args.gpus > 1 else "weak"
    if args.log2n <= 0:
        args.log2n = 24 if args.scaling == "strong" and args.gpus > 1 else 20
    if args.all_on_gpu0:
        args.exchange, args.dist_backend = "torch", "gloo"
    if args.exchange == "auto":
        args.exchange = "lib"
    return args


def self_launch(args):
    """`python bench.py --gpus N` with no launcher around it: start N fresh ranks (nothing in THIS process has touched
    torch or the GPU yet), relay rank 0's JSON line and the launcher's exit code."""
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith('{"metric"')]
    for ln in proc.stdout.splitlines():
        if not ln.startswith('{"metric"'):
            print(ln, file=sys.stderr)
    if lines:
        print(lines[-1])
    sys.stdout.flush()
    return proc.returncode if proc.returncode != 0 or lines else 1


def make_kernel(name):
    from pssgp.kernels import Matern12, Matern32, Matern52, RBF, Periodic, SquaredExponential
    if name.startswith("rbf"):
        return RBF(1.0, 1.0, order=int(name[3:]), balancing_iter=10)
    return {"matern12": lambda: Matern12(1.0, 1.0), "matern32": lambda: Matern32(1.0, 1.0),
            "matern52": lambda: Matern52(1.0, 1.0),
            # BASELINE config c5: quasi-periodic (Periodic * Matern32) + Matern52, d = 11
            "c5": lambda: Periodic(SquaredExponential(1.0, 1.0), period=1.0, order=1) * Matern32(1.0, 1.0)
            + Matern52(1.0, 1.0),
            "periodic10": lambda: Periodic(SquaredExponential(1.0, 1.0), period=1.0, order=10),
            # the reference's CO2 kernel at its default order 3 (pssgp/experiments/co2/mcmc.py:42-65): d = 18, the
            # wave-cooperative family
            "co2": lambda: Periodic(SquaredExponential(5.0, 1.0), period=1.0, order=3) * Matern32(0.1, 50.0)
            + Matern32(1.0, 100.0)}[name]()


def sample_prior_observations(P0, Fs, Qs, H, R, rng):
    """y_k = H x_k + N(0, R) with x from the SSM prior, by a log-depth doubling scan over the
    affine maps x -> F_k x + w_k in numpy (host-side data generation, not the product)."""
    n, d = Fs.shape[0], Fs.shape[1]

    def psd_sqrt(A):
        w, V = np.linalg.eigh(0.5 * (A + np.swapaxes(A, -1, -2)))
        return V * np.sqrt(np.clip(w, 0.0, None))[..., None, :]

    A = np.array(Fs, dtype=np.float64)
    b = np.einsum("nij,nj->ni", psd_sqrt(np.asarray(Qs, np.float64)), rng.standard_normal((n, d)))
    x0 = psd_sqrt(np.asarray(P0, np.float64)) @ rng.standard_normal(d)
    b[0] += A[0] @ x0
    s = 1
    while s < n:
        A2 = A.copy()
        b2 = b.copy()
        b2[s:] = np.einsum("nij,nj->ni", A[s:], b[:-s]) + b[s:]
        A2[s:] = A[s:] @ A[:-s]
        A, b = A2, b2
        s *= 2
    h = np.asarray(H, np.float64).reshape(d)
    return b @ h + np.sqrt(R) * rng.standard_normal(n)


def dominant_symbol(slot, d, suf, fam):
    """The device function(s) behind a timing slot (slots are named after the lane-chunk kernels), from the family the
    LIBRARY says the call runs on (pgps_get_family: PGPS_FAMILY_* of include/pgps.h) -- no copy of its dispatch rule here."""
    t = "double" if suf == "f64" else "float"
    if fam == 4:
        return {"k_filter_reduce": f"pgps::qc::q_reduce1<{d}> + the scan of the chain totals (rc_scan_blk_f / rc_ks_filter<float, {d}>)",
                "k_filter_apply": f"pgps::qc::q_apply1<{d}, ...>",
                "k_smoother_reduce": f"the scan of the smoothing totals (rc_scan_blk_s / rc_ks_smoother<float, {d}>)",
                "k_smoother_apply": f"pgps::qc::q_smooth1<{d}>"}[slot]
    if fam == 3:
        return {"k_filter_reduce": "pgps::rc::rc_reduce1<{t}, {d}> + the scan of the chain totals (rc_scan_blk_f / rc_ks_filter<{t}, {d}>)",
                "k_filter_apply": "pgps::rc::rc_apply1<{t}, {d}, ...>",
                "k_smoother_reduce": "the scan of the smoothing totals (rc_scan_blk_s / rc_ks_smoother<{t}, {d}>)",
                "k_smoother_apply": "pgps::rc::rc_smooth1<{t}, {d}, false>"}[slot].format(t=t, d=d)
    if fam == 5:
        # the two-rows level-1 kernels (csrc/pgps_rc2.hip.h), one instantiation per d >= 18; levels 2 and 3 stay wc_*
        dp = max(d, 18)
        return {"k_filter_reduce": f"pgps::rc2::rc2_reduce1<{t}, {dp}> + pgps::wc::wc_reduce2 + rc2_ks_filter levels + wc_enter1",
                "k_filter_apply": f"pgps::rc2::rc2_apply1<{t}, {dp}, ...>",
                "k_smoother_reduce": "pgps::wc::wc_sreduce2 + rc2_ks_smoother levels + wc_senter1",
                "k_smoother_apply": f"pgps::rc2::rc2_smooth1<{t}, {dp}, ...>"}[slot]
    if fam == 2:
        return {"k_filter_reduce": "pgps::wc::wc_reduce1/2 + wc_ks_filter levels", "k_filter_apply": "pgps::wc::wc_apply1",
                "k_smoother_reduce": "pgps::wc::wc_sreduce2 + wc_ks_smoother levels", "k_smoother_apply": "pgps::wc::wc_smooth1"}[slot]
    if slot == "k_pkfs_resident":
        return f"pgps::k_pkfs_resident<{t}, {d}, 16, false>"
    # lane-chunk kernels: the 128-lane build carries the suffix _n
    return f"pgps::{slot}{'_n' if fam == 11 else ''}<{t}, {d}, ...>"


def workload_name(args, d, suf, n_total, n_local, world):
    tag = {("matern32", "f64", 20, 1): "c2: ", ("rbf6", "f32", 20, 1): "c3: ", ("c5", "f64", 20, 1): "c5: "}.get(
        (args.kernel, suf, args.log2n, world), "")
    if world > 1 and args.scaling == "strong" and (args.kernel, suf, args.log2n) == ("matern32", "f64", 24):
        tag = "c4: "
    if world > 1 and args.scaling == "strong":
        return (f"{tag}{args.kernel} state-dim {d}, ONE series of N=2^{args.log2n} steps split into {world} contiguous "
                f"segments of {n_local} (strong scaling), {suf}, irregular times, prior-sampled observations")
    return (f"{tag}{args.kernel} state-dim {d}, N=2^{args.log2n} steps per GPU ({n_total} total), {suf}, irregular "
            f"times, prior-sampled observations")


def host_core_share():
    """Threads for the all-cores CPU baseline: the scheduler affinity capped by the cgroup CPU quota (a GPU box shows
    every core of the host but grants a share of them; more threads than the share only get throttled) and by 32."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:                      # cgroup v2: "<quota> <period>" or "max <period>"
            q, p = fh.read().split()[:2]
            if q != "max":
                n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
                q, p = int(fq.read()), int(fp.read())
                if q > 0:
                    n = min(n, max(1, q // p))
        except Exception:
            pass
    return max(1, min(n, 32))


# compulsory bytes per time step of each launch slot at the reference's pkf / pks contract (w =
# sizeof scalar): the filter's (3d^2+d+1)w belong to the Kalman pass (reads Fs, Qs, ys, writes fms, fPs), the smoother's
# (4d^2+2d)w to the backward pass; the reduce pass re-reads Fs, Qs, ys -- (2d^2+1)w it has to touch, none of them part
# of B_alg = (7d^2+3d+1)w (which is what `whole_path_frac` is measured against)
SLOT_BYTES = {"k_filter_reduce": lambda d, w: (2 * d * d + 1) * w,
              "k_filter_apply": lambda d, w: (3 * d * d + d + 1) * w,
              "k_smoother_reduce": lambda d, w: (3 * d * d + d) * w,
              "k_smoother_apply": lambda d, w: (4 * d * d + 2 * d) * w,
              # the resident launch is the whole pass: every byte of B_alg (it MOVES (5d^2+2d+1)w: the smoother's inputs never
              # leave the chip)
              "k_pkfs_resident": lambda d, w: (7 * d * d + 3 * d + 1) * w}
SLOT_INDEX = {"k_filter_reduce": 0, "k_filter_apply": 1, "k_smoother_reduce": 2, "k_smoother_apply": 3, "k_pkfs_resident": 6}
SLOT_MASK = sum(1 << i for i in SLOT_INDEX.values())


def slot_bytes(slot, d, w, fam):
    """Bytes of the contract per time step that the kernels of a launch slot touch themselves, for the family the call
    runs on -- or None where a slot holds no per-step traffic at all.  The cooperative families (row / quad: 3, 4; wave /
    two-rows: 2, 5) keep only the scans of their chain totals in the smoother's reduce slot: a few records per chain, not
    bytes per step -- priced with the lane-chunk model that slot read frac 6.27 in round 4."""
    if fam in (2, 3, 4, 5) and slot == "k_smoother_reduce":
        return None
    return SLOT_BYTES[slot](d, w)


def slot_moved_bytes(slot, d, w, fam):
    """What the slot's kernels really move per step where that differs from the contract: the cooperative families hand the
    smoothing elements (E, g, packed L) from the Kalman pass to the backward pass instead of re-reading Fs, Qs, fms, fPs."""
    if fam in (3, 4):
        elem = d * d + d + d * (d + 1) // 2
        if slot == "k_filter_apply":
            return (3 * d * d + d + 1 + elem) * w
        if slot == "k_smoother_apply":
            return (elem + d * d + d) * w
    if slot == "k_pkfs_resident":
        return (5 * d * d + 2 * d + 1) * w
    return slot_bytes(slot, d, w, fam)


def vector_fp(d, suf, n_local, ms_per_pass, alg_bytes_step):
    flops_step = 64 * d ** 3 + 40 * d ** 2
    peak = VECTOR_PEAK_TFLOPS[suf]
    achieved = flops_step * n_local / (ms_per_pass * 1e-3) / 1e12
    lb_hbm = alg_bytes_step * n_local / (HBM_PEAK_GBPS * 1e9) * 1e3
    lb_fp = flops_step * n_local / (peak * 1e12) * 1e3
    return {"flops_per_step": flops_step, "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
            "lower_bound_ms": max(lb_hbm, lb_fp), "lower_bound_by": "hbm" if lb_hbm >= lb_fp else "vector_fp",
            "frac_of_lower_bound": max(lb_hbm, lb_fp) / ms_per_pass}


def timed_rounds(fn, stream, sync, reps=20, rounds=5, warm=5):
    """GPU-event time per call of `fn`: `rounds` rounds of `reps` back-to-back calls after `warm` untimed ones; median,
    min and max over the rounds (one mean of 20 cannot tell a slow box from a noisy one)."""
    import torch
    for _ in range(warm):
        fn()
    sync()
    per = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            fn()
        e1.record(stream)
        sync()
        per.append(e0.elapsed_time(e1) / reps)
    per.sort()
    return {"ms_per_step": per[len(per) // 2], "ms_min": per[0], "ms_max": per[-1], "rounds": rounds, "reps_per_round": reps}


def one_gpu_reference(ctx, sde, d, suf, dtype_np, dtype_t, dev, stream, ts_all, n_total, noise):
    """ms per pass of the whole n_total-step series on this one GPU (the N = 1 point of the strong-scaling curve, same
    workload as the N > 1 line).  The observations are plain normal draws: the kernels' time does not depend on them."""
    import torch
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    real = ctypes.c_double if suf == "f64" else ctypes.c_float
    dev_from = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    F_d, P0_d = dev_from(np.asarray(sde.F, dtype_np)), dev_from(np.asarray(sde.P0, dtype_np))
    H_d = dev_from(np.asarray(sde.H, dtype_np).reshape(-1))
    ts_d = dev_from(ts_all.astype(dtype_np))
    Fs = torch.empty((n_total, d, d), dtype=dtype_t, device=dev)
    Qs = torch.empty((n_total, d, d), dtype=dtype_t, device=dev)
    ctx.call(f"pgps_discretise_dev_{suf}", ctypes.c_long(n_total), ctypes.c_int(d), P(F_d), P(P0_d), P(ts_d), real(0.0),
             P(Fs), P(Qs))
    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    ys = torch.randn(n_total, dtype=dtype_t, device=dev, generator=g) * float(np.sqrt(sde.P0[0, 0] + noise))
    fms, sms = (torch.empty((n_total, d), dtype=dtype_t, device=dev) for _ in range(2))
    fPs, sPs = (torch.empty((n_total, d, d), dtype=dtype_t, device=dev) for _ in range(2))
    ll = torch.zeros((2,), dtype=torch.float64, device=dev)

    def step():
        ctx.call(f"pgps_pkfs_dev_{suf}", ctypes.c_long(n_total), ctypes.c_int(d), P(P0_d), P(Fs), P(Qs), P(H_d), real(noise),
                 P(ys), P(fms), P(fPs), P(sms), P(sPs), P(ll))

    r = timed_rounds(step, stream, lambda: torch.cuda.synchronize(dev), reps=5, rounds=3, warm=3)
    del Fs, Qs, ys, fms, sms, fPs, sPs
    torch.cuda.empty_cache()
    return r["ms_per_step"]


def allgather_latency(ctx, d, dtype_t, dev, stream, world, n=100):
    """Median GPU-event time (microseconds) of an isolated ncclAllGather of each of the two segment records over the
    context's communicator: [filter record, smoother record]."""
    import torch
    from pssgp.distributed import record_lengths
    rf, rs, _ = record_lengths(d)
    out = []
    for reclen in (rf, rs):
        send = torch.zeros(reclen, dtype=dtype_t, device=dev)
        recv = torch.zeros((world, reclen), dtype=dtype_t, device=dev)
        nbytes = send.numel() * send.element_size()
        call = lambda: ctx.call("pgps_comm_allgather_dev", ctypes.c_void_p(send.data_ptr()), ctypes.c_void_p(recv.data_ptr()),
                                ctypes.c_size_t(nbytes))
        for _ in range(10):
            call()
        torch.cuda.synchronize(dev)
        ts_ = []
        for _ in range(n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            call()
            e1.record(stream)
            torch.cuda.synchronize(dev)
            ts_.append(e0.elapsed_time(e1) * 1e3)
        ts_.sort()
        out.append(ts_[len(ts_) // 2])
    return out


def kernel_source_sha():
    """sha256 over the kernel sources (csrc/*.hip, *.h): what a committed PMC figure was measured on (tools/pmc_traffic.py
    stamps its entries with it; a figure whose stamp differs from the tree's is reported with traffic_stale = true)."""
    import hashlib
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(ROOT, "parallel-gps_amd", "csrc", "*.h*"))):
        with open(path, "rb") as fh:
            h.update(os.path.basename(path).encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def committed_traffic(key, slot):
    """(bytes, source, stale): HBM-side bytes per launch of `slot` for workload `key` from the newest committed PMC
    summary that has it (profiles/r*_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes,
    fetch_factor * FETCH + WRITE); stale = the kernels have changed since it was collected (or it carries no stamp)."""
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
        try:
            with open(path) as fh:
                tj = json.load(fh)
            if key in tj and slot in tj[key]:
                stale = tj[key].get("_kernel_source_sha") != kernel_source_sha()
                return (tj[key][slot]["traffic_bytes"],
                        f"profiles/{os.path.basename(path)} (rocprofv3 --pmc, fetch_factor*FETCH_SIZE + WRITE_SIZE)", stale)
        except Exception:
            continue
    return None, None, None


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    # stdout carries the ONE JSON line and nothing else: whatever a library writes to file descriptor 1 on the way (RCCL
    # prints a version banner there when a communicator is created) goes to stderr
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    if os.environ.get("PGPS_BENCH_WATCHDOG"):       # debugging aid: dump every thread's stack after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["PGPS_BENCH_WATCHDOG"]), exit=True)
    import torch
    import torch.distributed as dist
    from pssgp import _backend

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    if args.all_on_gpu0:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_lib_exchange = args.exchange == "lib"
    if world > 1:
        if not use_lib_exchange and args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            # control plane only (rendezvous, barrier, max of the timings): CPU, never in the data path
            dist.init_process_group("gloo", rank=rank, world_size=world)

    dtype_np = np.float64 if args.dtype == "f64" else np.float32
    dtype_t = torch.float64 if args.dtype == "f64" else torch.float32
    suf = args.dtype
    w = 8 if suf == "f64" else 4
    if args.scaling == "strong":
        n_total = 1 << args.log2n
        if n_total % world:
            raise SystemExit(f"2^{args.log2n} steps do not split evenly over {world} GPUs")
        n_local = n_total // world
    else:
        n_local = 1 << args.log2n
        n_total = n_local * world

    # ---- model + synthetic series (seeded; identical on every rank, each keeps its segment) ----
    kern = make_kernel(args.kernel)
    sde = kern.get_sde()
    d = sde.F.shape[0]
    noise = 0.1
    rng = np.random.default_rng(0)
    ts_all = np.cumsum(0.05 * rng.uniform(0.5, 1.5, size=n_total))
    if args.grid == "reference":
        ts_all = np.linspace(0.0, 4.0, n_total)
    lo, hi = rank * n_local, (rank + 1) * n_local
    ts = ts_all[lo:hi]
    t_prev = 0.0 if rank == 0 else float(ts_all[lo - 1])

    ctx = _backend.Context(local_rank)
    stream = torch.cuda.current_stream(dev)
    ctx.set_stream(stream.cuda_stream)
    if args.chunk:
        ctx.set_chunk(args.chunk)
    ctx.set_stage(args.stage)
    ctx.set_family(args.family)
    ctx.set_block(args.block)
    ctx.set_dma(args.dma)
    ctx.set_rc_scan(args.rc_scan)
    ctx.set_single_pass(args.single_pass, 0)
    ctx.set_resident(args.resident)
    if args.f32_policy:
        ctx.set_f32_policy(args.f32_policy)

    def dev_from(a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    F_d = dev_from(np.asarray(sde.F, dtype_np))
    P0_d = dev_from(np.asarray(sde.P0, dtype_np))
    H_d = dev_from(np.asarray(sde.H, dtype_np).reshape(-1))
    ts_d = dev_from(ts.astype(dtype_np))
    Fs_d = torch.empty((n_local, d, d), dtype=dtype_t, device=dev)
    Qs_d = torch.empty((n_local, d, d), dtype=dtype_t, device=dev)
    real = ctypes.c_double if suf == "f64" else ctypes.c_float
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    # LTI discretisation on the GPU (the product path); Fs / Qs stay resident
    ctx.call(f"pgps_discretise_dev_{suf}", ctypes.c_long(n_local), ctypes.c_int(d), P(F_d), P(P0_d), P(ts_d),
             real(t_prev), P(Fs_d), P(Qs_d))
    torch.cuda.synchronize(dev)
    # observations from the model's prior (host), generated from the GPU-discretised model
    Fs_h, Qs_h = Fs_d.cpu().numpy(), Qs_d.cpu().numpy()
    ys_h = sample_prior_observations(sde.P0, Fs_h, Qs_h, sde.H, noise, np.random.default_rng(1000 + rank))
    if args.nan_frac > 0:
        ys_h[np.random.default_rng(7 + rank).random(n_local) < args.nan_frac] = np.nan
    ys_d = dev_from(ys_h.astype(dtype_np))

    fms = torch.empty((n_local, d), dtype=dtype_t, device=dev)
    fPs = torch.empty((n_local, d, d), dtype=dtype_t, device=dev)
    sms = torch.empty((n_local, d), dtype=dtype_t, device=dev)
    sPs = torch.empty((n_local, d, d), dtype=dtype_t, device=dev)
    ll_d = torch.zeros((2,), dtype=torch.float64, device=dev)

    rccl_info, exchange_fallback, one_gpu_ms = None, False, None
    if world > 1 and args.scaling == "strong" and args.path == "lgssm":
        # the reference of the strong-scaling factor: the SAME series length on ONE GPU (rank 0), outside the timed
        # region, before any communicator exists; the other ranks wait at the barrier
        if rank == 0:
            one_gpu_ms = one_gpu_reference(ctx, sde, d, suf, dtype_np, dtype_t, dev, stream, ts_all, n_total, noise)
        dist.barrier()

    if args.path != "lgssm":
        form = _backend.nilpotent_form(sde.F)
        if form is None or world != 1:
            raise SystemExit("--path fused needs a Matern kernel (d <= 3) and one GPU")
        lam, N1, N2 = form
        Pinf_h = np.ascontiguousarray(sde.P0, np.float64)
        H_h = np.ascontiguousarray(np.asarray(sde.H, np.float64).reshape(-1))
        HP = lambda a_: a_.ctypes.data_as(ctypes.c_void_p)
        null = ctypes.c_void_p(0)
        full = args.path == "fused"

        def step():
            ctx.call(f"pgps_gp_dev_{suf}", ctypes.c_long(n_local), ctypes.c_int(d), ctypes.c_double(lam), HP(N1), HP(N2),
                     HP(Pinf_h), HP(H_h), ctypes.c_double(noise), P(ts_d), ctypes.c_double(t_prev), P(ys_d),
                     P(fms) if full else null, P(fPs) if full else null, P(sms) if full else null,
                     P(sPs) if full else null, P(ll_d))
    elif world == 1 and not args.force_segments:
        def step():
            ctx.call(f"pgps_pkfs_dev_{suf}", ctypes.c_long(n_local), ctypes.c_int(d), P(P0_d), P(Fs_d), P(Qs_d),
                     P(H_d), real(noise), P(ys_d), P(fms), P(fPs), P(sms), P(sPs), P(ll_d))
    elif use_lib_exchange:
        # the product's multi-GPU path: the context owns the RCCL communicator, one library call per pass
        from pssgp import distributed as pdist
        import faulthandler

        # ncclCommInitRank is collective: a rank that cannot take part must say so BEFORE anyone enters it, or the others
        # wait for it for ever.  So: (1) rank 0 makes the id, or fails, and broadcasts the outcome over gloo; (2) every
        # rank reports over gloo whether it is ready; (3) only then the collective init, under a watchdog that ends
        # the process (non-zero: the launcher then ends the job) if a peer never joins.
        uid = None
        if rank == 0:
            try:
                uid = _backend.Context.comm_unique_id()
            except Exception as e:                  # noqa: BLE001
                print(f"[bench rank 0] pgps_comm_get_unique_id failed ({e!r})", file=sys.stderr)
        if world > 1:
            box = [uid]
            dist.broadcast_object_list(box, src=0)
            uid = box[0]
        ready = int(uid is not None and len(uid) == _backend.COMM_ID_BYTES)
        if world > 1:
            flag = torch.tensor([ready], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ready = int(flag.item())
        lib_ok = 0
        if ready:
            faulthandler.dump_traceback_later(int(os.environ.get("PGPS_COMM_INIT_TIMEOUT", "180")), exit=True)
            try:
                seg = pdist.ShardedScan(ctx, uid, rank, world, d, dtype_np)
                lib_ok = 1
            except Exception as e:                  # noqa: BLE001 -- reported, and decided together below
                print(f"[bench rank {rank}] RCCL communicator inside libpgps failed ({e!r})", file=sys.stderr)
            finally:
                faulthandler.cancel_dump_traceback_later()
        if world > 1:                               # every rank takes the same path
            flag = torch.tensor([lib_ok], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            lib_ok = int(flag.item())
        if lib_ok:
            ptrs = [t.data_ptr() for t in (P0_d, Fs_d, Qs_d, H_d)] + [noise] + [t.data_ptr() for t in (ys_d, fms, fPs, sms, sPs, ll_d)]
            n_rccl, r_rccl = ctx.comm_count()       # what RCCL itself says (ncclCommCount / ncclCommUserRank)
            if (n_rccl, r_rccl) != (world, rank):
                raise SystemExit(f"RCCL communicator reports {n_rccl} ranks / rank {r_rccl}, expected {world} / {rank}")
            rccl_info = {"in_library": True, "ranks": n_rccl, "allgather_us": allgather_latency(ctx, d, dtype_t, dev, stream, world),
                         "library": _backend.Context.comm_library()}

            def step():
                seg.pkfs(n_local, *ptrs)
        else:
            # the framework-hosted variant (tools/torch_segment_scan.py): the three library phases with torch.distributed's
            # RCCL collectives in between -- a FALLBACK, flagged as such in the JSON line
            print(f"[bench rank {rank}] falling back to --exchange torch", file=sys.stderr)
            from tools.torch_segment_scan import SegmentScan
            use_lib_exchange = False
            exchange_fallback = True
            try:
                ctx.comm_destroy()
            except Exception:                       # noqa: BLE001
                pass
            group = dist.new_group(backend="nccl") if world > 1 else None
            seg = SegmentScan(ctx, rank, world, d, dtype_np, torch_device=dev, group=group)
            rccl_info = {"in_library": False, "ranks": world if world > 1 else 1, "allgather_us": None,
                         "via": "torch.distributed nccl (fallback)"}

            def step():
                seg.pkfs(n_local, P0_d, Fs_d, Qs_d, H_d, noise, ys_d, fms, fPs, sms, sPs, ll_d)
    else:
        from tools.torch_segment_scan import SegmentScan
        seg = SegmentScan(ctx, rank, world, d, dtype_np, torch_device=dev)
        rccl_info = {"in_library": False, "ranks": world if args.dist_backend == "nccl" else 0, "allgather_us": None,
                     "via": f"torch.distributed {args.dist_backend} (--exchange torch)"}

        def step():
            seg.pkfs(n_local, P0_d, Fs_d, Qs_d, H_d, noise, ys_d, fms, fPs, sms, sPs, ll_d)

    def barrier():
        if world > 1:
            dist.barrier()

    # ---- warm-up, then the timed region ---------------------------------------------------
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    # which launch slot dominates a pass: measured (events around every launch of a short untimed run), not assumed
    n_probe = max(3, min(args.warmup, 10))
    ctx.profile_read(reset=True)
    ctx.profile_sample(1)
    ctx.profile_enable(SLOT_MASK)
    for _ in range(n_probe):
        step()
    probe = ctx.profile_read(reset=True)
    ctx.profile_enable(0)
    # (only slots whose kernels touch bytes of the contract can be priced against it: the cooperative families' scan slots cannot)
    fam_probe = ctx.get_family(n_local, d, f32=(suf == "f32"), what=3 if (world > 1 or args.force_segments) else 2)
    per_pass = {k: v[0] / n_probe for k, v in probe.items() if v[1] and k in SLOT_BYTES and slot_bytes(k, d, w, fam_probe) is not None}
    dominant = max(per_pass, key=per_pass.get) if per_pass else "k_smoother_apply"
    empty_pair_ms = ctx.profile_calibrate()
    ctx.profile_read(reset=True)
    # A timed launch goes out through hipExtLaunchKernelGGL with a start / stop event pair, which costs the stream ~5 us
    # (measured in round 5: a 20-pass region with every launch stamped 61.7 us per pass, GPU-event time, against 56.8 us in a
    # 200-pass region stamping every 8th -- the whole "driver protocol is 8 % slower" gap of rounds 2-4; a 55 ms re-warm
    # changed nothing).  A short region therefore stamps every 5th launch (4 of the driver's 20), a long one every
    # --event-every-th: <= 2 % of the region either way.
    sample_every = max(1, args.event_every) if args.steps > 32 else max(1, min(5, args.steps // 4 if args.steps >= 8 else 1))
    ctx.profile_sample(sample_every)
    dom_slot = SLOT_INDEX[dominant]
    # let the host's CPU-quota window refill after the data generation (a throttled host thread shows up as tens of
    # milliseconds of wall time with an idle GPU) -- and then bring the GPU back to its clocks under load: after half a
    # second of idling a handful of passes (round 4: min(warmup, 10) = 0.45 ms of work) left the 20-step timed region of the
    # driver's protocol 8 % slower than a 200-step one (88.8 against 81.8 us per pass, GPU-event time, same kernels).  The
    # re-warm is TIME based: untimed back-to-back passes for >= --rewarm-ms right before the region (`rewarm_ms` in the line).
    time.sleep(0.5)
    rw0 = time.perf_counter()

    def rewarm_batch():
        for _ in range(16):
            step()
        torch.cuda.synchronize(dev)

    rewarm_batch()
    batch_ms = (time.perf_counter() - rw0) * 1e3
    if world > 1:           # every rank makes the SAME number of passes (each carries the segment exchange's collectives)
        bm = torch.tensor([batch_ms], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(bm, op=dist.ReduceOp.MAX)
        batch_ms = float(bm.item())
    n_more = int(min(4096, max(0, np.ceil((args.rewarm_ms - batch_ms) / max(batch_ms, 1e-3)))))
    for _ in range(n_more):
        rewarm_batch()
    rewarm_passes = 16 * (1 + n_more)
    rewarm_ms = (time.perf_counter() - rw0) * 1e3
    ctx.profile_read(reset=True)
    ctx.profile_enable((1 << dom_slot) if args.event_every > 0 else 0)   # time the dominant slot's launches
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        step()
    ev1.record(stream)
    t_enq = time.perf_counter()
    while not ev1.query():              # spin: a blocking wait can oversleep by tens of ms on a busy host
        pass
    torch.cuda.synchronize(dev)
    barrier()
    t1 = time.perf_counter()
    gpu_ms = ev0.elapsed_time(ev1)
    elapsed = t1 - t0
    prof = ctx.profile_read(reset=True)
    ctx.profile_enable(0)
    if world > 1:
        tmax = torch.tensor([elapsed, gpu_ms], dtype=torch.float64,
                            device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed, gpu_ms = float(tmax[0].item()), float(tmax[1].item())

    # per-kernel breakdown (separate, untimed loop with events around every launch)
    ctx.profile_sample(1)
    ctx.profile_enable(0x7f)
    n_break = min(args.steps, 50)
    for _ in range(n_break):
        step()
    breakdown = ctx.profile_read(reset=True)
    ctx.profile_enable(0)

    ll_val = float(ll_d[0].item())
    status_flags = ctx.status()         # bit 2: float32 calls of this run ran in fp64 arithmetic (dense grid / policy)
    promoted = suf == "f32" and bool(status_flags & 4)
    # the kernel family the timed calls ran on, as the library itself reports it (a promoted float32 call: the fp64 family)
    fam = ctx.get_family(n_local, d, f32=(suf == "f32" and not promoted), what=3 if (world > 1 or args.force_segments) else 2)

    # ---- roofline of the dominant kernel -----------------------------------------------------
    dom_ms, dom_n = prof[dominant]
    if not dom_n:                                          # --event-every 0: fall back to the breakdown loop's timing
        dom_ms, dom_n = breakdown[dominant]
    # the events are stamped by hipExtLaunchKernelGGL with the dispatch's own start / end timestamps
    # (the clock rocprofv3's kernel trace reads), so no event-packet overhead is included.  A slot of the row- /
    # wave-cooperative families can hold several launches per pass (the Kogge-Stone levels sit in the reduce slots);
    # the two slots that can dominate there, k_filter_apply and k_smoother_apply, are one launch per pass
    dom_raw_ms = dom_ms / max(dom_n, 1)
    dom_avg_s = max(dom_raw_ms, 1e-9) * 1e-3
    dom_bytes = (slot_bytes(dominant, d, w, fam) or 0) * n_local
    achieved = dom_bytes / dom_avg_s / 1e9 if dom_avg_s > 0 else 0.0
    alg_bytes_step = (7 * d * d + 3 * d + 1) * w
    # HBM-side bytes per launch of the dominant kernel from the committed PMC profile of this workload
    traffic, traffic_src, traffic_stale = None, None, None
    default_cfg = (world == 1 and not args.chunk and args.stage < 0 and args.family == 0 and args.path == "lgssm"
                   and args.grid == "baseline" and not args.f32_policy)
    tkey = f"{args.kernel}_{suf}_log2n{args.log2n}"
    if default_cfg:
        traffic, traffic_src, traffic_stale = committed_traffic(tkey, dominant)
    # every launch slot of the pass against the bytes of the contract it touches itself (two slots within a few per cent of
    # each other swap the `dominant` role from box to box: read them side by side)
    slots = {}
    for sname in SLOT_BYTES:
        ms_tot, n_l = breakdown.get(sname, (0.0, 0))
        if not n_l:
            continue
        ms_pass = ms_tot / n_break
        bs = slot_bytes(sname, d, w, fam)
        ab = bs * n_local if bs is not None else None      # None: the slot's kernels touch no per-step bytes (scans of chain totals)
        mb = slot_moved_bytes(sname, d, w, fam)
        tb = committed_traffic(tkey, sname) if default_cfg else (None, None, None)
        slots[sname] = {"ms_per_pass": ms_pass, "launches_per_pass": n_l / n_break, "alg_bytes": ab,
                        "moved_bytes_model": mb * n_local if mb is not None else None,
                        "frac": (ab / (ms_pass * 1e-3) / 1e9 / HBM_PEAK_GBPS if ms_pass > 0 else 0.0) if ab is not None else None,
                        "traffic": tb[0], "traffic_stale": tb[2]}
    whole_gbps = alg_bytes_step * n_total * args.steps / (gpu_ms * 1e-3) / 1e9 / world     # per GPU

    out = {
        "metric": "timesteps/sec (filter+smooth+log-lik)",
        "value": n_total * args.steps / elapsed,
        "unit": "timesteps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "rewarm_ms": rewarm_ms,             # untimed back-to-back passes right before the timed region (clocks under load)
        "rewarm_passes": rewarm_passes,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": args.scaling if world > 1 else None,     # one GPU: neither weak nor strong
        "vs_baseline": None,
        "dtype": suf,
        "data": "synthetic",
        "path": args.path,
        "config": {"workload": workload_name(args, d, suf, n_total, n_local, world),
                   "steps_per_gpu": n_local, "steps_total": n_total, "state_dim": d,
                   "parallelism": "1 GPU" if world == 1 else
                   f"{world} contiguous time segments, 2 all-gathers of segment totals per pass ("
                   + ("RCCL ncclAllGather issued by libpgps on the context's stream" if use_lib_exchange
                      else f"torch.distributed {args.dist_backend}") + ")"},
        "roofline": {"bound": "hbm", "kernel": dominant, "kernel_symbol": dominant_symbol(dominant, d, "f64" if promoted else suf, fam),
                     "kernel_family": fam,
                     "achieved": achieved, "peak": HBM_PEAK_GBPS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                     "traffic_source": traffic_src, "traffic_stale": traffic_stale, "slots": slots,
                     "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_ms": dom_avg_s * 1e3,
                     "launches_timed": dom_n, "launch_sampled_every": sample_every, "empty_event_pair_ms": empty_pair_ms,
                     "dominant_by": {k: round(v, 6) for k, v in per_pass.items()},
                     # the whole pass against its algorithmic bytes B_alg = (7d^2+3d+1)w per step, per GPU,
                     # from the GPU-event time of the timed region
                     "whole_path_GBps": whole_gbps, "whole_path_frac": whole_gbps / HBM_PEAK_GBPS,
                     "algorithmic_bytes_per_step": alg_bytes_step,
                     # from d = 6 the path sits near the vector-FMA ridge, so the same pass is also
                     # priced against the vector peak with the work-efficient-scan estimate 64 d^3 + 40 d^2 flop per
                     # step; lower_bound_ms = max(bytes / HBM peak, flops / vector peak) for this rank's steps
                     "vector_fp": vector_fp(d, suf, n_local, gpu_ms / args.steps, alg_bytes_step)},
        "whole_path_effective_GBps": alg_bytes_step * n_total * args.steps / elapsed / 1e9,
        "kernel_ms": {k: (v[0] / v[1] if v[1] else 0.0) for k, v in breakdown.items() if v[1]},
        "kernel_ms_per_pass": {k: v[0] / n_break for k, v in breakdown.items() if v[1]},
        "host_enqueue_ms_per_step": (t_enq - t0) / args.steps * 1e3,
        "gpu_event_ms_per_step": gpu_ms / args.steps,
        "log_likelihood": ll_val,
        "grid": args.grid,
        "f32_promoted": bool(status_flags & 4) if suf == "f32" else None,
        "chunk": ctx.get_chunk(n_local),
        # lane-chunk kernels (d <= 6): lanes per workgroup, steps per lane, workgroups of the pass that was timed
        "lane_geometry": list(ctx.get_geometry(n_local, d)) if fam in (1, 11) else None,
    }

    if world > 1 or args.force_segments:
        out["rccl"] = rccl_info
        out["exchange_fallback"] = bool(exchange_fallback)
    if one_gpu_ms is not None:
        n_ms = elapsed / args.steps * 1e3
        out["strong_scaling"] = {"one_gpu_ms": one_gpu_ms, "n_gpu_ms": n_ms, "speedup": one_gpu_ms / n_ms,
                                 "efficiency": one_gpu_ms / n_ms / world,
                                 "series": f"the same 2^{args.log2n} steps in one pgps_pkfs_dev_{suf} call on rank 0's GPU, "
                                           "median of 3 rounds of 5 passes, before the communicator is built"}

    # ---- the same workload through the fused entry point (ts, ys resident; Fs / Qs never read) ------------
    if rank == 0 and world == 1 and args.path == "lgssm" and not args.main_only and _backend.nilpotent_form(sde.F) is not None:
        lam, N1, N2 = _backend.nilpotent_form(sde.F)
        Pinf_h = np.ascontiguousarray(sde.P0, np.float64)
        H_h = np.ascontiguousarray(np.asarray(sde.H, np.float64).reshape(-1))
        HP = lambda a_: a_.ctypes.data_as(ctypes.c_void_p)
        null = ctypes.c_void_p(0)

        def gp_step(full):
            ctx.call(f"pgps_gp_dev_{suf}", ctypes.c_long(n_local), ctypes.c_int(d), ctypes.c_double(lam), HP(N1), HP(N2),
                     HP(Pinf_h), HP(H_h), ctypes.c_double(noise), P(ts_d), ctypes.c_double(t_prev), P(ys_d),
                     P(fms) if full else null, P(fPs) if full else null, P(sms) if full else null,
                     P(sPs) if full else null, P(ll_d))

        fused = {}
        sync = lambda: torch.cuda.synchronize(dev)
        rounds, reps = (5, 20) if args.steps >= 20 else (2, max(1, args.steps))
        for name, full in (("filter+smooth+log-lik", True), ("log-lik only", False)):
            r_ = timed_rounds(lambda: gp_step(full), stream, sync, reps=reps, rounds=rounds)
            r_.update({"timesteps_per_s": n_local / r_["ms_per_step"] * 1e3, "log_likelihood": float(ll_d[0].item())})
            if full:
                # the same contract priced the same way (B_alg per step against 8 TB/s), next to the bytes this road really has
                # to move: (t, y) in, four moment arrays out -- Fs / Qs never exist
                moved = (2 + 2 * (d * d + d)) * w
                gbps = alg_bytes_step * n_local / (r_["ms_per_step"] * 1e-3) / 1e9
                tr = committed_traffic(tkey, "fused_path") if default_cfg else (None, None, None)
                r_["roofline"] = {"bound": "hbm", "algorithmic_bytes_per_step": alg_bytes_step, "achieved": gbps, "peak": HBM_PEAK_GBPS,
                                  "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS, "moved_bytes_per_step": moved,
                                  "moved_GBps": moved * n_local / (r_["ms_per_step"] * 1e-3) / 1e9,
                                  "traffic": tr[0], "traffic_source": tr[1], "traffic_stale": tr[2]}
            fused[name] = r_
        # geometry of the fused kernels (always the 256-lane build: csrc/pgps_inst.hip launch_gp)
        g_lc, g_nb = ctx.get_chunk(n_local)
        fused["geometry"] = {"lanes_per_workgroup": 256, "steps_per_lane": g_lc, "workgroups": g_nb}
        if d <= 2 and dtype_np == np.float64:
            from pssgp.model import StateSpaceGP
            gm = StateSpaceGP((np.zeros((1, 1)), np.zeros((1, 1))), kern, noise_variance=noise, parallel=True)
            gmodel, _, npar = _backend.pack_grad_model(gm._grad_blocks())
            g_d = torch.zeros((8,), dtype=torch.float64, device=dev)

            def grad_step():
                ctx.call("pgps_gp_ll_grad_dev_f64", ctypes.c_long(n_local), ctypes.c_int(d), ctypes.c_int(npar),
                         HP(gmodel), P(ts_d), ctypes.c_double(t_prev), P(ys_d), P(g_d))

            r_ = timed_rounds(grad_step, stream, sync, reps=reps, rounds=rounds)
            r_.update({"timesteps_per_s": n_local / r_["ms_per_step"] * 1e3, "log_likelihood": float(g_d[0].item()),
                       "gradient": [float(v) for v in g_d[1:1 + npar].tolist()]})
            fused["log-lik + gradient (3 hyper-parameters, forward-mode duals)"] = r_
            if hasattr(ctx.lib, "pgps_gp_ll_grad_adj_dev_f64"):
                # the adjoint pass of the fused path (csrc/pgps_gpadj.hip.h): [ll | Abar | Ubar | Hbar | Rbar], contracted here
                a_d = torch.zeros((2 + d * d + 2 * d,), dtype=torch.float64, device=dev)

                def adj_step():
                    ctx.call("pgps_gp_ll_grad_adj_dev_f64", ctypes.c_long(n_local), ctypes.c_int(d), ctypes.c_double(lam), HP(N1),
                             HP(N2), HP(Pinf_h), HP(H_h), ctypes.c_double(noise), P(ts_d), ctypes.c_double(t_prev), P(ys_d), P(a_d))

                r2 = timed_rounds(adj_step, stream, sync, reps=reps, rounds=rounds)
                st = a_d.cpu().numpy()
                Fm = np.asarray(N1, np.float64).reshape(d, d) - lam * np.eye(d)
                g_adj = [float(st[1 + d * d:1 + d * d + d] @ (Pinf_h @ H_h)) / float(kern.variance),
                         -float(st[1:1 + d * d] @ Fm.reshape(-1)) / float(kern.lengthscales), float(st[1 + d * d + 2 * d])]
                r2.update({"timesteps_per_s": n_local / r2["ms_per_step"] * 1e3, "log_likelihood": float(st[0]), "gradient": g_adj,
                           "max_rel_diff_to_duals": float(np.max(np.abs(np.array(g_adj) - np.array(r_["gradient"])))
                                                          / max(1e-300, float(np.max(np.abs(r_["gradient"])))))})
                fused["log-lik + gradient (adjoint pass: one filter pass, one reverse pass)"] = r2
        # predict_f on the device: N training steps + N/4 query times (merge + filter + smoother + projection)
        kq = max(1, n_local // 4)
        tq_d = (torch.rand(kq, dtype=torch.float64, device=dev) * float(ts_d[-1].item())).sort().values.to(dtype_t)
        pm_d = torch.empty(kq, dtype=dtype_t, device=dev)
        pv_d = torch.empty(kq, dtype=dtype_t, device=dev)

        def predict_step():
            ctx.call(f"pgps_gp_predict_dev_{suf}", ctypes.c_long(n_local), ctypes.c_long(kq), ctypes.c_int(d),
                     ctypes.c_double(lam), HP(N1), HP(N2), HP(Pinf_h), HP(H_h), ctypes.c_double(noise), P(ts_d), P(ys_d),
                     ctypes.c_double(t_prev), P(tq_d), P(pm_d), P(pv_d), P(ll_d))

        r_ = timed_rounds(predict_step, stream, sync, reps=reps, rounds=rounds)
        r_["merged_timesteps_per_s"] = (n_local + kq) / r_["ms_per_step"] * 1e3
        fused[f"predict_f on device (N train + N/4 = {kq} queries)"] = r_
        fused["note"] = ("pgps_gp_dev: discretisation fused into the scan kernels, inputs are (ts, ys) only; "
                         "GPU-event time, not part of `value`")
        out["fused_path"] = fused

    # ---- general-LTI device path (any kernel, fp64, 2 <= d <= 16): ts, ys in, results out ---------------------
    if (rank == 0 and world == 1 and args.path == "lgssm" and not args.main_only and dtype_np == np.float64 and 2 <= d <= 16
            and _backend.nilpotent_form(sde.F) is None):
        F_h = np.ascontiguousarray(sde.F, np.float64)
        Pinf_h = np.ascontiguousarray(sde.P0, np.float64)
        H_h = np.ascontiguousarray(np.asarray(sde.H, np.float64).reshape(-1))
        HP = lambda a_: a_.ctypes.data_as(ctypes.c_void_p)
        kq = max(1, n_local // 4)
        tq_d = (torch.rand(kq, dtype=torch.float64, device=dev) * float(ts_d[-1].item())).sort().values
        pm_d = torch.empty(kq, dtype=torch.float64, device=dev)
        pv_d = torch.empty(kq, dtype=torch.float64, device=dev)

        def lti_ll_step():
            ctx.call("pgps_lti_ll_dev_f64", ctypes.c_long(n_local), ctypes.c_int(d), HP(F_h), HP(Pinf_h), HP(H_h),
                     ctypes.c_double(noise), P(ts_d), P(ys_d), ctypes.c_double(t_prev), P(ll_d))

        def lti_predict_step():
            ctx.call("pgps_lti_predict_dev_f64", ctypes.c_long(n_local), ctypes.c_long(kq), ctypes.c_int(d), HP(F_h),
                     HP(Pinf_h), HP(H_h), ctypes.c_double(noise), P(ts_d), P(ys_d), ctypes.c_double(t_prev), P(tq_d),
                     P(pm_d), P(pv_d), P(ll_d))

        lti = {}
        sync = lambda: torch.cuda.synchronize(dev)
        for name, fn, steps in (("log-lik only (discretise + filter)", lti_ll_step, n_local),
                                (f"predict_f on device (N train + N/4 = {kq} queries)", lti_predict_step, n_local + kq)):
            r_ = timed_rounds(fn, stream, sync, reps=min(max(1, args.steps), 10), rounds=5 if args.steps >= 10 else 2, warm=2)
            r_.update({"timesteps_per_s": steps / r_["ms_per_step"] * 1e3, "log_likelihood": float(ll_d[0].item())})
            lti[name] = r_
        lti["note"] = ("pgps_lti_*_dev_f64: discretisation, scan and projection on the device, inputs are (ts, ys[, tq]) "
                       "only; GPU-event time, not part of `value`")
        out["lti_path"] = lti

    # ---- CPU baseline: the oracle's C restatement of the reference's sequential path ----------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import c_oracle
        ssm = (np.asarray(sde.P0), Fs_h.astype(np.float64), Qs_h.astype(np.float64),
               np.asarray(sde.H).reshape(1, -1), np.array([[noise]]))
        times = []
        reps = 5
        for _ in range(reps):
            c0 = time.perf_counter()
            res = c_oracle.kfs(ssm, ys_h, dtype_np)
            times.append(time.perf_counter() - c0)
        cpu_t = float(np.median(times))
        out["cpu_baseline"] = {"value": n_local / cpu_t, "unit": "timesteps/s", "cores": 1, "kind": "port",
                               "sample": f"the full workload ({n_local} steps), sequential kf+ks "
                                         f"(oracle/kalman_seq.c), median of {reps} runs"}
        # the same pass on all host cores (chunked scan, oracle/kalman_par.c, OpenMP) -- for context
        if dtype_np == np.float64 and d <= 16:
            os.environ.pop("OMP_NUM_THREADS", None)             # pinned to 1 above for numpy's sake only
            cores = host_core_share()
            ptimes = []
            for _ in range(reps):
                c0 = time.perf_counter()
                pres = c_oracle.par_kfs(ssm, ys_h, cores)
                ptimes.append(time.perf_counter() - c0)
            par_t = float(np.median(ptimes))
            out["cpu_baseline_all_cores"] = {
                "value": n_local / par_t, "unit": "timesteps/s", "cores": cores, "kind": "port",
                "sample": f"the full workload ({n_local} steps), chunked scan with OpenMP over {cores} threads "
                          f"(oracle/kalman_par.c), median of {reps} runs",
                "ll_rel_vs_sequential": abs(pres[4] - res[4]) / abs(res[4])}
        # the checker also checks: GPU result vs the sequential oracle on the benchmarked arrays
        cf, cP, cs, csP, cll = res
        out["parity_vs_cpu_oracle"] = {"ll_rel": abs(ll_val - cll) / abs(cll)}
        if args.path != "fused-ll":
            out["parity_vs_cpu_oracle"].update({
                "smoothed_mean_rel": float(np.max(np.abs(sms.cpu().numpy() - cs)) / np.max(np.abs(cs))),
                "smoothed_cov_rel": float(np.max(np.abs(sPs.cpu().numpy() - csP)) / np.max(np.abs(csP)))})
    if rank == 0:
        print(json.dumps(out), file=json_out, flush=True)
    if use_lib_exchange and (world > 1 or args.force_segments) and args.path == "lgssm":
        ctx.synchronize()
        seg.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
