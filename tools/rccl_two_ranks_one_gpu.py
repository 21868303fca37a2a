"""Can two ranks of the in-library RCCL exchange share ONE GPU?  (The boxes of this pool have one GPU each, so the product's
multi-rank path -- pgps_comm_init + pgps_pkfs_seg_dev_* -- has only ever run with a communicator of size one.)  Two fresh
processes, both on GPU 0, the id handed over through a file, a watchdog on the collective init; if RCCL accepts the
communicator, a series split in two goes through the sharded pass and is compared with the unsharded one.
  python tools/rccl_two_ranks_one_gpu.py           (exits 0 and says which of the two happened)"""
import multiprocessing as mp
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rank_main(rank, world, path, n, q):
    import faulthandler
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "parallel-gps_amd"))
    import ctypes
    import numpy as np
    from pssgp import _backend as B, distributed as pdist
    from pssgp.kernels import Matern32
    faulthandler.dump_traceback_later(60, exit=True)           # a rendezvous that never completes ends the rank
    try:
        ctx = B.Context(0)
        uid = pdist.share_unique_id(rank, path=path, run_id="two-on-one")
        d = 2
        try:
            seg = pdist.ShardedScan(ctx, uid, rank, world, d, np.float64)
        except Exception as e:                      # noqa: BLE001
            q.put((rank, "init-failed", repr(e)))
            return
        faulthandler.cancel_dump_traceback_later()
        faulthandler.dump_traceback_later(60, exit=True)
        sde = Matern32(1.0, 1.0).get_sde()
        t = np.cumsum(0.05 * np.random.default_rng(0).uniform(0.5, 1.5, n))
        Fs, Qs = B.discretise(sde.F, sde.P0, t, 0.0)
        y = np.sin(t) + 0.3 * np.random.default_rng(1).standard_normal(n)
        lo, hi = rank * n // world, (rank + 1) * n // world
        host = dict(P0=np.asarray(sde.P0, np.float64), Fs=Fs[lo:hi], Qs=Qs[lo:hi], H=np.asarray(sde.H, np.float64).reshape(-1), ys=y[lo:hi])
        ptr = {}
        for k, a in host.items():
            a = np.ascontiguousarray(a, np.float64)
            ptr[k] = ctx.malloc(a.nbytes)
            ctx.h2d(ptr[k], a)
        nl = hi - lo
        out = {k: np.empty(s) for k, s in (("fms", (nl, d)), ("fPs", (nl, d, d)), ("sms", (nl, d)), ("sPs", (nl, d, d)))}
        for k, a in out.items():
            ptr[k] = ctx.malloc(a.nbytes)
        ptr["ll"] = ctx.malloc(16)
        seg.pkfs(nl, ptr["P0"], ptr["Fs"], ptr["Qs"], ptr["H"], 0.1, ptr["ys"], ptr["fms"], ptr["fPs"], ptr["sms"], ptr["sPs"], ptr["ll"])
        ctx.synchronize()
        for k, a in out.items():
            ctx.d2h(a, ptr[k])
        ll = np.empty(2)
        ctx.d2h(ll, ptr["ll"])
        n_rccl, r_rccl = ctx.comm_count()
        # the unsharded pass on a context of its own
        sms, sPs, fms, fPs, ll1 = B.pkfs((host["P0"], Fs, Qs, host["H"].reshape(1, -1), np.array([[0.1]])), y, return_filtered=True,
                                        return_loglikelihood=True)
        err = max(float(np.max(np.abs(out["sms"] - sms[lo:hi]))), float(np.max(np.abs(out["sPs"] - sPs[lo:hi]))),
                  float(np.max(np.abs(out["fms"] - fms[lo:hi]))))
        q.put((rank, "ok", dict(ranks=n_rccl, rank=r_rccl, max_abs_err=err, ll=float(ll[0]), ll_unsharded=float(ll1),
                               library=B.Context.comm_library())))
        seg.close()
    except Exception as e:                          # noqa: BLE001
        q.put((rank, "error", repr(e)))


def main():
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    path = os.path.join(tempfile.mkdtemp(), "uid")
    procs = [mpc.Process(target=rank_main, args=(r, 2, path, 20000, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(150)
    res = []
    while not q.empty():
        res.append(q.get())
    for p in procs:
        if p.is_alive():
            p.terminate()
    print("exit codes", [p.exitcode for p in procs])
    for r in sorted(res):
        print(r)
    if not res:
        print("no rank reported: the rendezvous did not complete (watchdog) -- two ranks on one GPU are not possible here")
    return 0


if __name__ == "__main__":
    sys.exit(main())
