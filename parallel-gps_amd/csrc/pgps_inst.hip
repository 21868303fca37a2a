// pgps_inst.hip -- one translation unit per compiled (dtype, state dimension): the kernels of
// pgps_kernels.hip.h / pgps_discretise.hip.h instantiated for PGPS_INST_T, PGPS_INST_D, and the
// launch functions the C ABI dispatches to.  Split this way because the fully unrolled algebra
// for d >= 5 takes minutes to compile; the Makefile builds the units in parallel.
//
// With -DPGPS_NARROW -DPGPS_BLOCK=128 the unit holds a second build of the array-path scan only, with 128-lane
// workgroups (half the lanes per workgroup = half the scan tree per step at the same number of workgroups: c2 86.7 ->
// 82.6 us, RBF order 6 fp32 0.66 -> 0.59 ms, 2^14 .. 2^16 steps 32 -> 26 us; the fused kernels prefer 256 lanes).  Its
// kernels and its launch function carry their own names so that both builds link into one library.
#ifdef PGPS_NARROW
#define k_filter_reduce k_filter_reduce_n
#define k_filter_apply k_filter_apply_n
#define k_filter_single k_filter_single_n
#define k_smoother_reduce k_smoother_reduce_n
#define k_smoother_apply k_smoother_apply_n
#define k_seg_filter_total k_seg_filter_total_n
#define k_seg_smoother_total k_seg_smoother_total_n
#define launch_scan launch_scan_narrow
#include "pgps_kernels.hip.h"
#else
#include "pgps_discretise.hip.h"
#include "pgps_fused.hip.h"
#include "pgps_gpadj.hip.h"
#include "pgps_gradlti.h"
#include "pgps_kernels.hip.h"
#endif

#include <type_traits>

#ifndef PGPS_INST_T
#error "compile with -DPGPS_INST_T=<float|double> -DPGPS_INST_D=<d>"
#endif

// single-pass filter where the grid fits the chip: automatic choice (A/B: profiles/r03_experiments.txt)
#ifndef PGPS_SINGLE_PASS_AUTO
#define PGPS_SINGLE_PASS_AUTO false
#endif

namespace pgps {

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

template <typename T, int D>
static int carve_workspace(pgps_ctx* ctx, ScanArgs<T>& a) {
    const size_t nl = (size_t)a.nlanes, nb = (size_t)a.nblocks;
    size_t off = 0;
    const size_t o_spine = off;  off = align_up(off + nb * Dim<D>::NFILT * sizeof(T), 256);
    const size_t o_lpre = off;   off = align_up(off + nl * Dim<D>::NFILT * sizeof(T), 256);
    const size_t o_sspine = off; off = align_up(off + nb * Dim<D>::NSMTH * sizeof(T), 256);
    const size_t o_lsuf = off;   off = align_up(off + nl * Dim<D>::NSMTH * sizeof(T), 256);
    const size_t o_ll = off;     off = align_up(off + nb * sizeof(double), 256);
    const size_t o_status = off; off = align_up(off + 16, 256);
    const size_t o_seg = off;    off = align_up(off + (2 * (size_t)(D + D * D) + 2 * D * D) * sizeof(T), 256);
    const size_t o_flags = off;  off = align_up(off + (nb + 4 + 8 * 32) * sizeof(int), 256);       // (ticket +) the single-pass barrier's counter shards
    const size_t o_incl = off;   off = align_up(off + nb * Dim<D>::NMP * sizeof(T), 256);
    int rc = ensure(ctx, ctx->ws, off);
    if (rc) return rc;
    char* base = (char*)ctx->ws.p;
    a.spine = (T*)(base + o_spine);
    a.lpre = (T*)(base + o_lpre);
    a.sspine = (T*)(base + o_sspine);
    a.lsuf = (T*)(base + o_lsuf);
    a.llpart = (double*)(base + o_ll);
    a.status = ctx->status_word;
    (void)o_status;
    a.seg_ws = (T*)(base + o_seg);
    a.ticket = (int*)(base + o_flags);
    a.flags = a.ticket + 4;
    a.incl = (T*)(base + o_incl);
#ifdef PGPS_STAMPS
    rc = ensure(ctx, ctx->stamps, (size_t)3 * nb * 8 * sizeof(long long));
    if (rc) return rc;
    a.stamps = (long long*)ctx->stamps.p;
#endif
    return PGPS_OK;
}

#ifdef PGPS_NARROW
static_assert(kBlock == kBlockNarrow, "the narrow build is compiled with -DPGPS_BLOCK=128");
#endif

template <typename T, int D, int G, bool NT>
static int launch_scan_g(pgps_ctx* ctx, ScanArgs<T> a, Mode mode) {
    HIPCHK(ctx, hipSetDevice(ctx->device));
#ifdef PGPS_NARROW
    geometry_narrow(ctx, a.N, &a.Lc, &a.nblocks, D);
#else
    geometry(ctx, a.N, &a.Lc, &a.nblocks, D);
#endif
    a.nlanes = (long)a.nblocks * kBlock;
    a.nt = NT ? 1 : 0;
    a.shortcut = (ctx->shortcut != 0 && (long)kBlock * a.Lc >= 2048) ? 1 : 0;
    int rc = carve_workspace<T, D>(ctx, a);
    if (rc) return rc;
    const dim3 grid(a.nblocks), block(kBlock);
    hipStream_t s = ctx->stream;
    // Whole series: nothing reads the stitching operands, but they must point at memory -- the compiler turns the small
    // `k + 1 < N ? Fs[k + 1] : halo_FQ` choices of d <= 2 into loads of both sides and a select (k_smoother_reduce<T, 2>
    // faulted on a null halo_FQ in the stand-alone smoother).
    a.carry_in = a.seg_ws;
    a.halo_FQ = a.seg_ws + (D + D * D);
    a.carry_back = a.seg_ws + (D + D * D) + 2 * D * D;
    // the Kalman pass with its inputs through an LDS-DMA ring (FilterApplyDma): 128-lane build, d = 2, fp64, 4-step
    // sub-tiles, 16 or 32 steps per lane.  A measured experiment, off unless asked for (pgps_set_dma(ctx, 1)): the ring
    // fills while the spine is folded, but issuing its 64 pieces costs the prologue what the loop then saves -- the
    // streaming loops are bound by the CU's memory path (10 - 14 B/clk), not by latency (profiles/r03_experiments.txt)
    bool dma = false;
#ifdef PGPS_NARROW
    if constexpr (sizeof(T) == 8 && D == 2 && G == 4)
        dma = ctx->dma > 0 && (a.Lc == 16 || a.Lc == 32);
#endif
    // whole-series filter + smoother: the smoothing totals in innovation form (pgps_math.h smth_extend_u) -- the fold of a step into
    // the lane's total is one matrix product and a rank-one update instead of three matrix products, and no L = P - E F P
    // (-DPGPS_DFORM=0: the reference's (E, g, L) elements everywhere, as the segment protocol keeps them)
#ifndef PGPS_DFORM
#define PGPS_DFORM 1
#endif
    const bool dform = PGPS_DFORM != 0 && mode == MODE_PKFS && !dma;
    a.dform = dform ? 1 : 0;
    auto launch_apply = [&](auto smooth_tag) {
        constexpr bool SMOOTH = decltype(smooth_tag)::value;
        if constexpr (SMOOTH) {
            if (dform) {
                timed_launch(ctx, PGPS_K_FILTER_APPLY, k_filter_apply<T, D, true, G, NT, false, true>, grid, block, 0, a);
                return;
            }
        }
#ifdef PGPS_NARROW
        if constexpr (sizeof(T) == 8 && D == 2 && G == 4) {
            if (dma) {
                timed_launch(ctx, PGPS_K_FILTER_APPLY, k_filter_apply<T, D, SMOOTH, G, NT, true>, grid, block, 0, a);
                return;
            }
        }
#endif
        timed_launch(ctx, PGPS_K_FILTER_APPLY, k_filter_apply<T, D, SMOOTH, G, NT>, grid, block, 0, a);
    };
    (void)dma;
    if (mode == MODE_SEG_REDUCE || mode == MODE_SEG_FILTER || mode == MODE_SEG_SMOOTHER) {
        // one segment of a series sharded over GPUs: carry_in / halo_FQ / carry_back live in seg_ws
        a.seg_first = (a.rank == 0);
        a.seg_last = (a.rank == a.nranks - 1);
        a.carry_in = a.seg_ws;
        a.halo_FQ = a.seg_ws + (D + D * D);
        a.carry_back = a.seg_ws + (D + D * D) + 2 * D * D;
        const int pad = seg_rec_s_pad(D);
        if (mode == MODE_SEG_REDUCE) {
            timed_launch(ctx, PGPS_K_FILTER_REDUCE, k_filter_reduce<T, D, G>, grid, block, 0, a);
            hipLaunchKernelGGL((k_seg_filter_total<T, D>), dim3(1), block, 0, s, a);
        } else if (mode == MODE_SEG_FILTER) {
            // halo step of this segment = first step of the next rank, straight out of its record
            if (!a.seg_last)
                a.halo_FQ = a.gathered_f + (long)(a.rank + 1) * seg_rec_f_len(D) + Dim<D>::NFILT;
            launch_apply(std::true_type{});
            hipLaunchKernelGGL((k_seg_smoother_total<T, D>), dim3(1), block, 0, s, a, pad);
        } else timed_launch(ctx, PGPS_K_SMOOTHER_APPLY, k_smoother_apply<T, D, G, NT>, grid, block, 0, a);
        HIPCHK(ctx, hipGetLastError());
        return PGPS_OK;
    }
    if constexpr (G == 4 && D <= 2) {
        // single-pass filter: whole series on this GPU, 16 steps per lane (they live in registers), one grid barrier
        // inside -- so every workgroup must be resident: one wave per SIMD (the chunk's registers allow no more), i.e.
        // workgroups x waves <= 4 x CUs.  Automatic where that holds (pgps_set_single_pass(ctx, 0) turns it off).
        const bool fits = (long)a.nblocks * kWaves <= 4L * ctx->n_cu && a.Lc == 16;
        const bool want = ctx->single_pass < 0 ? PGPS_SINGLE_PASS_AUTO : ctx->single_pass != 0;
        if (want && fits && (mode == MODE_PKF || mode == MODE_PKFS)) {
            a.win = 0;
            a.dform = 0;                    // (k_filter_single builds the reference's elements)
            HIPCHK(ctx, hipMemsetAsync(a.flags, 0, 8 * 32 * sizeof(int), s));        // the barrier's 8 counter shards
            if (mode == MODE_PKFS) {
                timed_launch(ctx, PGPS_K_FILTER_APPLY, k_filter_single<T, D, true, 16, NT>, grid, block, 0, a);
            } else {
                timed_launch(ctx, PGPS_K_FILTER_APPLY, k_filter_single<T, D, false, 16, NT>, grid, block, 0, a);
                if (a.ll)
                    timed_launch(ctx, PGPS_K_LL_FINALIZE, k_ll_finalize, dim3(1), block, 0, (const double*)a.llpart,
                                 a.nblocks, a.ll);
            }
            if (mode == MODE_PKFS) {
                ScanArgs<T> b = a;
                timed_launch(ctx, PGPS_K_SMOOTHER_APPLY, k_smoother_apply<T, D, G, NT>, grid, block, 0, b);
            }
            HIPCHK(ctx, hipGetLastError());
            return PGPS_OK;
        }
    }
    if (mode == MODE_PKF || mode == MODE_PKFS) {
        timed_launch(ctx, PGPS_K_FILTER_REDUCE, k_filter_reduce<T, D, G>, grid, block, 0, a);
        if (mode == MODE_PKFS) launch_apply(std::true_type{}); else {
            a.ll_in_apply = (a.ll != nullptr && a.status != nullptr) ? 1 : 0;
            launch_apply(std::false_type{});
            if (a.ll && !a.ll_in_apply) {
                timed_launch(ctx, PGPS_K_LL_FINALIZE, k_ll_finalize, dim3(1), block, 0, (const double*)a.llpart, a.nblocks,
                             a.ll);
            }
        }
    }
    if (mode == MODE_PKS) {
        ScanArgs<T> b = a;
        b.ll = nullptr;
        timed_launch(ctx, PGPS_K_SMOOTHER_REDUCE, k_smoother_reduce<T, D>, grid, block, 0, b);
    }
    if (mode == MODE_PKS || mode == MODE_PKFS) {
        ScanArgs<T> b = a;
        if (mode == MODE_PKS) b.ll = nullptr;
        timed_launch(ctx, PGPS_K_SMOOTHER_APPLY, k_smoother_apply<T, D, G, NT>, grid, block, 0, b);
    }
    HIPCHK(ctx, hipGetLastError());
    return PGPS_OK;
}

// G = steps per lane per LDS-staged sub-tile (0 = direct global accesses); staging exists for
// d <= 3 only (StageCfg), G = 2 only in fp64 (a lane segment must be >= 16 bytes).
template <typename T, int D, int G>
static int launch_scan_nt(pgps_ctx* ctx, const ScanArgs<T>& a, Mode mode) {
    // streaming (non-temporal) output stores once one pass -- (7d^2+3d+1) scalars per step -- is well
    // beyond the 256 MiB Infinity Cache: -12 % on the smoother pass at N = 2^24, +5 % at 2^20
    const bool nt = (double)a.N * (7 * D * D + 3 * D + 1) * sizeof(T) > 512.0 * 1024 * 1024;
    if constexpr (G > 0) {
        if (nt) return launch_scan_g<T, D, G, true>(ctx, a, mode);
    }
    return launch_scan_g<T, D, G, false>(ctx, a, mode);
}

template <typename T, int D>
int launch_scan(pgps_ctx* ctx, ScanArgs<T> a, Mode mode) {
    if constexpr (D <= 2) {
        int g = ctx->stage_g < 0 ? 4 : ctx->stage_g;
        if (g == 4) return launch_scan_nt<T, D, 4>(ctx, a, mode);
        if constexpr (sizeof(T) == 8) {
            if (g == 2) return launch_scan_nt<T, D, 2>(ctx, a, mode);
        }
    } else if constexpr (D == 3) {
        // 144-byte lane segments: 2 steps of 72-byte fp64 records or 4 steps of 36-byte fp32 ones
        constexpr int g3 = (sizeof(T) == 8) ? 2 : 4;
        if (ctx->stage_g != 0) return launch_scan_nt<T, D, g3>(ctx, a, mode);
    } else if constexpr (D == 4 || (D == 6 && sizeof(T) == 4)) {
        // one step per sub-tile: 64- / 128-byte (d = 4) and 144-byte (d = 6 fp32) records
        if (ctx->stage_g != 0) return launch_scan_nt<T, D, 1>(ctx, a, mode);
    }
    return launch_scan_nt<T, D, 0>(ctx, a, mode);
}

#ifdef PGPS_NARROW
template int launch_scan<PGPS_INST_T, PGPS_INST_D>(pgps_ctx*, ScanArgs<PGPS_INST_T>, Mode);
#else
template <typename T, int D>
int launch_disc(pgps_ctx* ctx, long N, const T* F, const T* Pinf, const T* ts, T t0, T* Fs, T* Qs) {
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int block = 256;
    const long grid = (N + block - 1) / block;
    timed_launch(ctx, PGPS_K_DISCRETISE, k_discretise<T, D>, dim3((unsigned)grid), dim3(block), 0, N, F, Pinf, ts, t0, Fs,
                 Qs);
    HIPCHK(ctx, hipGetLastError());
    return PGPS_OK;
}

// ---- fused-discretisation ("gp") launches: d <= 3 ------------------------------------------------------
template <typename T, int D, bool NT>
static int launch_gp_nt(pgps_ctx* ctx, GpArgs<T> g, int want_filtered, int want_smoothed) {
    ScanArgs<T>& a = g.s;
    const dim3 grid(a.nblocks), block(kBlock);
    timed_launch(ctx, PGPS_K_FILTER_REDUCE, k_gp_reduce<T, D>, grid, block, 0, g);
    if (want_smoothed) {
        timed_launch(ctx, PGPS_K_FILTER_APPLY, k_gp_apply<T, D, true, NT>, grid, block, 0, g);
        if (g.qslot)
            timed_launch(ctx, PGPS_K_SMOOTHER_APPLY, k_gp_smooth<T, D, false, true>, grid, block, 0, g);
        else
            timed_launch(ctx, PGPS_K_SMOOTHER_APPLY, k_gp_smooth<T, D, NT>, grid, block, 0, g);
    } else {
        (void)want_filtered;
        a.ll_in_apply = (a.ll != nullptr && a.status != nullptr) ? 1 : 0;
        timed_launch(ctx, PGPS_K_FILTER_APPLY, k_gp_apply<T, D, false, NT>, grid, block, 0, g);
        if (a.ll && !a.ll_in_apply)
            timed_launch(ctx, PGPS_K_LL_FINALIZE, k_ll_finalize, dim3(1), block, 0, (const double*)a.llpart, a.nblocks,
                         a.ll);
    }
    HIPCHK(ctx, hipGetLastError());
    return PGPS_OK;
}

template <typename T, int D>
int launch_gp(pgps_ctx* ctx, GpArgs<T> g, int want_filtered, int want_smoothed) {
    if constexpr (D <= 3) {
        HIPCHK(ctx, hipSetDevice(ctx->device));
        ScanArgs<T>& a = g.s;
        geometry(ctx, a.N, &a.Lc, &a.nblocks);
        // a short series: ONE workgroup, ONE launch (k_gp_one) instead of three launches of a few workgroups
        const bool one = ctx->one_launch != 0 && ctx->chunk <= 0 &&
                         a.N <= (ctx->one_launch > 0 ? (long)ctx->one_launch : (long)kOneLaunchAuto);
        if (one) {
            long v = (a.N + kBlock - 1) / kBlock;
            v = (v + 3) / 4 * 4;
            a.Lc = (int)v;
            a.nblocks = 1;
        }
        a.nlanes = (long)a.nblocks * kBlock;
        a.seg_first = 1;
        a.seg_last = 1;
        a.shortcut = (ctx->shortcut != 0 && (long)kBlock * a.Lc >= 2048) ? 1 : 0;
        int rc = carve_workspace<T, D>(ctx, a);
        if (rc) return rc;
        if (one) {
            const dim3 grid1(1), block1(kBlock);
            if (!want_smoothed) timed_launch(ctx, PGPS_K_FILTER_APPLY, k_gp_one<T, D, 0>, grid1, block1, 0, g);
            else if (g.qslot) timed_launch(ctx, PGPS_K_FILTER_APPLY, k_gp_one<T, D, 2>, grid1, block1, 0, g);
            else timed_launch(ctx, PGPS_K_FILTER_APPLY, k_gp_one<T, D, 1>, grid1, block1, 0, g);
            HIPCHK(ctx, hipGetLastError());
            return PGPS_OK;
        }
        // bytes one pass moves: t, y in; (d^2 + d) filtered out and back in, (d^2 + d) smoothed out
        const double pass = (double)a.N * (2 + 3 * (D * D + D)) * sizeof(T);
        if (pass > 512.0 * 1024 * 1024) return launch_gp_nt<T, D, true>(ctx, g, want_filtered, want_smoothed);
        return launch_gp_nt<T, D, false>(ctx, g, want_filtered, want_smoothed);
    } else {
        (void)ctx; (void)g; (void)want_filtered; (void)want_smoothed;
        return PGPS_E_UNSUPPORTED_DIM;
    }
}

// ---- batched log-likelihood (B models over one series) ---------------------------------------------------
template <typename T, int D>
int launch_gp_batch(pgps_ctx* ctx, int B, GpBatchArgs<T> b) {
    if constexpr (D <= 3) {
        HIPCHK(ctx, hipSetDevice(ctx->device));
        // steps per lane: the serial part is the efficient one, so as long as the batch keeps the chip
        // covered (>= 1024 workgroups) use 16 steps per lane; halve towards 4 when B x N is small
        int lc = ctx->chunk;
        if (lc <= 0) {
            lc = 16;
            while (lc > 4 && (long)B * ((b.N + (long)kBlock * lc - 1) / ((long)kBlock * lc)) < 1024) lc /= 2;
            if (b.N < (long)kBlock * 4) lc = (int)((b.N + kBlock - 1) / kBlock);
            if (lc < 1) lc = 1;
        }
        b.Lc = lc;
        b.nblocks = (int)((b.N + (long)kBlock * lc - 1) / ((long)kBlock * lc));
        b.nlanes = (long)b.nblocks * kBlock;
        if (b.nblocks > 65535 * 16) return PGPS_E_INVALID;
        const size_t nb = (size_t)b.nblocks, nl = (size_t)b.nlanes, nB = (size_t)B;
        auto up = [](size_t x) { return (x + 255) / 256 * 256; };
        size_t off = 0;
        const size_t o_spine = off; off = up(off + nB * nb * Dim<D>::NFILT * sizeof(T));
        const size_t o_lpre = off;  off = up(off + nB * nl * Dim<D>::NFILT * sizeof(T));
        const size_t o_ll = off;    off = up(off + nB * nb * sizeof(double));
        int rc = ensure(ctx, ctx->ws, off);
        if (rc) return rc;
        char* base = (char*)ctx->ws.p;
        b.spine = (T*)(base + o_spine);
        b.lpre = (T*)(base + o_lpre);
        b.llpart = (double*)(base + o_ll);
        const dim3 grid(b.nblocks, B), block(kBlock);
        timed_launch(ctx, PGPS_K_FILTER_REDUCE, k_gpb_reduce<T, D>, grid, block, 0, b);
        timed_launch(ctx, PGPS_K_FILTER_APPLY, k_gpb_apply<T, D>, grid, block, 0, b);
        timed_launch(ctx, PGPS_K_LL_FINALIZE, k_gpb_finalize, dim3(B), block, 0, (const double*)b.llpart, b.nblocks, b.ll);
        HIPCHK(ctx, hipGetLastError());
        return PGPS_OK;
    } else {
        (void)ctx; (void)B; (void)b;
        return PGPS_E_UNSUPPORTED_DIM;
    }
}

// ---- fused path: log-likelihood and the model's adjoints (pgps_gpadj.hip.h), fp64 units of d <= 3 ---------------------
template <typename T, int D>
int launch_gp_adj(pgps_ctx* ctx, GpArgs<double> g, double* out) {
    if constexpr (D <= 3 && std::is_same<T, double>::value) {
        HIPCHK(ctx, hipSetDevice(ctx->device));
        ScanArgs<double>& a = g.s;
        geometry(ctx, a.N, &a.Lc, &a.nblocks);
        const bool one = ctx->one_launch != 0 && ctx->chunk <= 0 &&
                         a.N <= (ctx->one_launch > 0 ? (long)ctx->one_launch : (long)kOneLaunchAuto);
        if (one) {
            long v = (a.N + kBlock - 1) / kBlock;
            v = (v + 3) / 4 * 4;
            a.Lc = (int)v;
            a.nblocks = 1;
        }
        a.nlanes = (long)a.nblocks * kBlock;
        a.seg_first = 1;
        a.seg_last = 1;
        int rc = carve_workspace<double, D>(ctx, a);
        if (rc) return rc;
        constexpr int NX = D + Dim<D>::SYM, NST = gp_adj_nstat<D>();
        const size_t n_xs = (size_t)a.Lc * NX * (size_t)a.nlanes, n_gp = (size_t)a.nblocks * NST;
        rc = ensure(ctx, ctx->gadj, (n_xs + n_gp) * sizeof(double));
        if (rc) return rc;
        GpAdjArgs ga{};
        ga.g = g;
        ga.xs = (double*)ctx->gadj.p;
        ga.gpart = ga.xs + n_xs;
        ga.out = out;
        const dim3 grid(a.nblocks), block(kBlock);
        if (one) {
            timed_launch(ctx, PGPS_K_FILTER_APPLY, k_gp_gone<D>, dim3(1), block, 0, ga);
        } else {
            timed_launch(ctx, PGPS_K_FILTER_REDUCE, k_gp_reduce<double, D>, grid, block, 0, g);
            timed_launch(ctx, PGPS_K_FILTER_APPLY, k_gp_gfwd<D>, grid, block, 0, ga);
            timed_launch(ctx, PGPS_K_SMOOTHER_APPLY, k_gp_gback<D>, grid, block, 0, ga);
            timed_launch(ctx, PGPS_K_LL_FINALIZE, k_grad_lti_finalize, dim3(1 + NST), dim3(256), 0, (long)a.nblocks, (int)NST,
                         (const double*)a.llpart, (const double*)ga.gpart, out);
        }
        HIPCHK(ctx, hipGetLastError());
        return PGPS_OK;
    } else {
        (void)ctx; (void)g; (void)out;
        return PGPS_E_UNSUPPORTED_DIM;
    }
}
template int launch_gp_adj<PGPS_INST_T, PGPS_INST_D>(pgps_ctx*, GpArgs<double>, double*);

template int launch_gp_batch<PGPS_INST_T, PGPS_INST_D>(pgps_ctx*, int, GpBatchArgs<PGPS_INST_T>);
template int launch_gp<PGPS_INST_T, PGPS_INST_D>(pgps_ctx*, GpArgs<PGPS_INST_T>, int, int);
template int launch_scan<PGPS_INST_T, PGPS_INST_D>(pgps_ctx*, ScanArgs<PGPS_INST_T>, Mode);
template int launch_disc<PGPS_INST_T, PGPS_INST_D>(pgps_ctx*, long, const PGPS_INST_T*, const PGPS_INST_T*,
                                                   const PGPS_INST_T*, PGPS_INST_T, PGPS_INST_T*, PGPS_INST_T*);
#endif      // PGPS_NARROW

}  // namespace pgps
