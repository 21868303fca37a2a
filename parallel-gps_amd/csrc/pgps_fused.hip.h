// pgps_fused.hip.h -- the scan with the LTI discretisation fused in ("gp" entry points).
//
// For SDEs whose drift is  F = -lambda I + N  with N nilpotent (every Matern kernel, balanced or not:
// pssgp/kernels/matern/common.py:10-18) the transition matrix has the closed form
//     expm(dt F) = exp(-lambda dt) (I + dt N + dt^2 N^2 / 2 + ...)          (d terms)
// so  Fs[k], Qs[k] = Pinf - Fs[k] Pinf Fs[k]^T  (pssgp/kernels/base.py:29-47) cost one exp and a few
// dozen FMAs per step.  These kernels compute them in registers from (t_k, t_{k-1}) inside the three
// scan launches instead of reading the (N, d, d) arrays: a pass reads 2 scalars per step (t, y)
// instead of 2 d^2 + 1.  This is the StateSpaceGP path (model.py:92-117): times and observations in,
// log-likelihood and posterior out; Fs / Qs never exist in HBM.
//
// Structure, scratch and combine order are those of pgps_kernels.hip.h (k_filter_reduce /
// k_filter_apply / k_smoother_apply); only the source of (F, Q) differs.  Outputs go through the
// wave-private LDS staging so the stores stay coalesced; the filtered moments the smoother reads
// back are staged in the same way.  The discretisation arithmetic is fp64 whatever T is.
#pragma once

#include "pgps_kernels.hip.h"

namespace pgps {

template <typename T, int D>
__device__ __forceinline__ void lti_step(const GpModel<T>& m, T dt_in, T* F, T* Qf) {
    constexpr int MAT = D * D;
    const double dt = double(dt_in);
    const double e = exp(-m.lam * dt);
    if constexpr (D == 1) {
        F[0] = T(e);
        Qf[0] = T(-m.Pinf[0] * expm1(-2.0 * m.lam * dt));
    } else {
        double Fd[MAT], X[MAT];
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < D; ++j) {
                double v = (i == j ? 1.0 : 0.0) + dt * m.N1[i * D + j];
                if (D >= 3) v += dt * dt * m.N2[i * D + j];
                Fd[i * D + j] = e * v;
            }
        mat_mul<double, D>(Fd, m.Pinf, X);
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < D; ++j) {
                double acc = 0.0, acct = 0.0;
#pragma unroll
                for (int l = 0; l < D; ++l) { acc += X[i * D + l] * Fd[j * D + l]; acct += X[j * D + l] * Fd[i * D + l]; }
                Qf[i * D + j] = T(0.5 * (m.Pinf[i * D + j] + m.Pinf[j * D + i]) - 0.5 * (acc + acct));
                F[i * D + j] = T(Fd[i * D + j]);
            }
    }
}

template <typename T, int D>
__device__ __forceinline__ void gp_prior(const GpModel<T>& m, T* h, T* P0 /*sym*/) {
#pragma unroll
    for (int i = 0; i < D; ++i) h[i] = m.H[i];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = i; j < D; ++j) P0[symi<D>(i, j)] = T(0.5 * (m.Pinf[i * D + j] + m.Pinf[j * D + i]));
}

// ---------------------------------------------------------------------------------------------
// reduce
// ---------------------------------------------------------------------------------------------
// LDS of the fused kernels: declared once per kernel and handed to the bodies, so that the one-launch kernel (below), which
// runs the three bodies one after the other, pays for the largest of them and not for their sum
template <typename T, int D>
struct GpLds {
    static constexpr int G = 4;
    using GF = StageGeom<D * D * (int)sizeof(T), G>;
    using GM = StageGeom<D * (int)sizeof(T), G>;
    T lds[kWaves * Dim<D>::NFILT];
    double lds_ll[kWaves];
    __attribute__((aligned(16))) char stage[kWaves][GF::BYTES + GM::BYTES];
};

template <typename T, int D>
__device__ __forceinline__ void gp_reduce_body(const GpArgs<T>& g, T* lds) {
    constexpr int MAT = D * D, SYM = Dim<D>::SYM;
    using FE = FiltElem<T, D>;
    const ScanArgs<T>& a = g.s;
    T h[D], P0[SYM];
    gp_prior<T, D>(g.m, h, P0);
    const long gt = (long)blockIdx.x * kBlock + threadIdx.x;
    const long k0 = gt * a.Lc;
    const long k1 = min(a.N, k0 + a.Lc);
    FE agg;
    filt_identity(agg);
    if (k0 < k1) {
        T tprev = (k0 > 0) ? g.m.ts[k0 - 1] : g.m.t_prev;
        T tn = g.m.ts[k0], yn = a.ys[k0];
        for (long k = k0; k < k1; ++k) {
            const T t = tn, y = yn;
            if (k + 1 < k1) { tn = g.m.ts[k + 1]; yn = a.ys[k + 1]; }
            if (k == 0) {
                filt_first(agg, P0, y, h, a.R);
            } else {
                T F[MAT], Qf[MAT], Q[SYM];
                lti_step<T, D>(g.m, t - tprev, F, Qf);
                sym_from_full<T, D>(Qf, Q);
                filt_extend(agg, F, Q, y, h, a.R);
            }
            tprev = t;
        }
    }
    FE excl, total;
    block_scan_exclusive<FE, true>(agg, excl, total, lds);
    ws_store(a.lpre, a.nlanes, gt, excl);
    if (threadIdx.x == 0) rec_store(a.spine + (long)blockIdx.x * Dim<D>::NFILT, total);
}

template <typename T, int D>
__global__ __launch_bounds__(kBlock) void k_gp_reduce(const GpArgs<T> g) {
    __shared__ T lds[kWaves * Dim<D>::NFILT];
    gp_reduce_body<T, D>(g, lds);
}

// ---------------------------------------------------------------------------------------------
// apply (+ log-likelihood, + smoothing aggregates when SMOOTH); fms / fPs are written when non-null
// ---------------------------------------------------------------------------------------------
template <typename T, int D, bool SMOOTH, bool NT>
__device__ __forceinline__ void gp_apply_body(const GpArgs<T>& g, GpLds<T, D>& sh) {
    constexpr int MAT = D * D, SYM = Dim<D>::SYM, NF = Dim<D>::NFILT, G = 4;
    using FE = FiltElem<T, D>;
    using SE = SmthElem<T, D>;
    using MC = MeanCov<T, D>;
    using GF = StageGeom<MAT * (int)sizeof(T), G>;
    using GM = StageGeom<D * (int)sizeof(T), G>;
    const ScanArgs<T>& a = g.s;
    T* lds = sh.lds;
    double* lds_ll = sh.lds_ll;
    auto& stage = sh.stage;
    (void)NF;

    T h[D];
    MC s;
    gp_prior<T, D>(g.m, h, s.P);
#pragma unroll
    for (int i = 0; i < D; ++i) s.m[i] = T(0);

    const long gt = (long)blockIdx.x * kBlock + threadIdx.x;
    const long k0 = gt * a.Lc;
    const long k1 = min(a.N, k0 + a.Lc);
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    const long wbase = ((long)blockIdx.x * kBlock + wave * kWave) * a.Lc;
    const bool store = (a.fms != nullptr);
    const bool staged = store && (wbase + (long)kWave * a.Lc <= a.N) && (a.Lc % G == 0);

    FE left_part, lp;
    MC s_short;
    const bool shortcut = a.shortcut != 0 && blockIdx.x > 0 && carry_shortcut_filter<T, D>(a.spine, (int)blockIdx.x - 1, s_short);
    if (blockIdx.x > 0 && !shortcut) fold_spine_partial<FE>(a.spine, 0, (int)blockIdx.x, left_part);
    ws_load(a.lpre, a.nlanes, gt, lp);
    T tprev = T(0), tn = T(0), yn = T(0);
    if (k0 < k1) {
        tprev = (k0 > 0) ? g.m.ts[k0 - 1] : g.m.t_prev;
        tn = g.m.ts[k0];
        yn = a.ys[k0];
    }
    if (shortcut) {
        s = s_short;
    } else if (blockIdx.x > 0) {
        FE left;
        block_reduce_ordered(left_part, left, lds);
        filt_apply(s, left);
    }
    filt_apply(s, lp);

    LogLik ll;
    SE sagg;
    smth_identity(sagg);
    char* lP = stage[wave];
    char* lM = lP + GF::BYTES;
    const long pitchF = (long)a.Lc * MAT * sizeof(T), pitchM = (long)a.Lc * D * sizeof(T);
    char* gP = reinterpret_cast<char*>(a.fPs + wbase * MAT);
    char* gM = reinterpret_cast<char*>(a.fms + wbase * D);
    if (k0 < k1) {
        for (long k = k0; k < k1; ++k) {
            const T t = tn, y = yn;
            if (k + 1 < a.N && (k + 1 < k1 || SMOOTH)) tn = g.m.ts[k + 1];
            if (k + 1 < k1) yn = a.ys[k + 1];
            T F[MAT], Qf[MAT];
            lti_step<T, D>(g.m, t - tprev, F, Qf);
            tprev = t;
            filter_apply_step<T, D, SMOOTH>(a, k, k0, F, Qf, y, h, s, ll, sagg);
            if (store) {
                T Pf[MAT];
                full_from_sym<T, D>(s.P, Pf);
                if (staged) {
                    const int i = (int)((k - k0) % G);
                    if (i == 0) wave_lds_sync();
                    stage_put<GM, T, D>(lM, i, s.m);
                    stage_put<GF, T, MAT>(lP, i, Pf);
                    if (i == G - 1) {
                        const long sb = (k - k0) / G;
                        wave_lds_sync();
                        stage_drain<GM, NT>(gM + sb * GM::SEG, pitchM, lM);
                        stage_drain<GF, NT>(gP + sb * GF::SEG, pitchF, lP);
                    }
                } else {
                    store_rec<T, D>(a.fms + k * D, s.m);
                    store_rec<T, MAT>(a.fPs + k * MAT, Pf);
                }
            }
        }
        if (SMOOTH) {
            T F[MAT], Qf[MAT];
            const bool have_next = (k1 < a.N);
            if (have_next) lti_step<T, D>(g.m, tn - tprev, F, Qf);
            filter_tail_apply<T, D>(have_next, F, Qf, s, sagg);
        }
    }
    {
        const double v = ll.value();
        const double t = block_sum_double(v, lds_ll);
        if (threadIdx.x == 0) a.llpart[blockIdx.x] = t;
        if constexpr (!SMOOTH) {
            if (a.ll != nullptr && a.ll_in_apply) ll_finish(a.llpart, a.nblocks, a.ll, a.status + kLlTicketWord, lds_ll);
        }
    }
    if (SMOOTH) {
        SE excl, total;
        block_scan_exclusive<SE, false>(sagg, excl, total, lds);
        ws_store(a.lsuf, a.nlanes, gt, excl);
        if (threadIdx.x == 0) rec_store(a.sspine + (long)blockIdx.x * Dim<D>::NSMTH, total);
    }
    (void)lane;
}

template <typename T, int D, bool SMOOTH, bool NT>
__global__ __launch_bounds__(kBlock) void k_gp_apply(const GpArgs<T> g) {
    __shared__ GpLds<T, D> sh;
    gp_apply_body<T, D, SMOOTH, NT>(g, sh);
}

// ---------------------------------------------------------------------------------------------
// batched log-likelihood: B hyper-parameter settings over the same (ts, ys) in one pair of launches
// (HMC leapfrogs / grid search at the reference's realistic N of
// 1e3..1e5, where one series alone cannot fill the chip).  blockIdx.y selects the model; each model
// has its own slice of the scan scratch.  Same bodies, same arithmetic as the single-model launches.
// ---------------------------------------------------------------------------------------------
template <typename T, int D>
__device__ __forceinline__ GpArgs<T> gp_batch_select(const GpBatchArgs<T>& b) {
    const long m = blockIdx.y;
    const double* p = b.models + m * kGpModelStride;
    GpArgs<T> g{};
    g.s.N = b.N; g.s.Lc = b.Lc; g.s.nblocks = b.nblocks; g.s.nlanes = b.nlanes;
    g.s.seg_first = 1; g.s.seg_last = 1;
    g.s.ys = b.ys;
    g.s.R = T(p[31]);
    g.s.spine = b.spine + m * b.nblocks * Dim<D>::NFILT;
    g.s.lpre = b.lpre + m * b.nlanes * Dim<D>::NFILT;
    g.s.llpart = b.llpart + m * b.nblocks;
    g.m.lam = p[0];
#pragma unroll
    for (int i = 0; i < D * D; ++i) { g.m.N1[i] = p[1 + i]; g.m.N2[i] = p[10 + i]; g.m.Pinf[i] = p[19 + i]; }
#pragma unroll
    for (int i = 0; i < D; ++i) g.m.H[i] = T(p[28 + i]);
    g.m.ts = b.ts;
    g.m.t_prev = b.t_prev;
    return g;
}

template <typename T, int D>
__global__ __launch_bounds__(kBlock) void k_gpb_reduce(const GpBatchArgs<T> b) {
    const GpArgs<T> g = gp_batch_select<T, D>(b);
    __shared__ T lds[kWaves * Dim<D>::NFILT];
    gp_reduce_body<T, D>(g, lds);
}

template <typename T, int D>
__global__ __launch_bounds__(kBlock) void k_gpb_apply(const GpBatchArgs<T> b) {
    const GpArgs<T> g = gp_batch_select<T, D>(b);
    __shared__ GpLds<T, D> sh;
    gp_apply_body<T, D, false, false>(g, sh);
}

// one workgroup per model: ll[m] = sum of its block partials
static __global__ __launch_bounds__(kBlock) void k_gpb_finalize(const double* llpart, int nblocks, double* ll) {
    __shared__ double lds_ll[kWaves];
    const double* p = llpart + (long)blockIdx.x * nblocks;
    double v = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += kBlock) v += p[i];
    const double t = block_sum_double(v, lds_ll);
    if (threadIdx.x == 0) ll[blockIdx.x] = t;
}

// ---------------------------------------------------------------------------------------------
// smoother apply
// ---------------------------------------------------------------------------------------------
template <typename T, int D, bool NT, bool PROJ = false>
__device__ __forceinline__ void gp_smooth_body(const GpArgs<T>& g, GpLds<T, D>& sh) {
    constexpr int MAT = D * D, SYM = Dim<D>::SYM, NS = Dim<D>::NSMTH, G = 4;
    using SE = SmthElem<T, D>;
    using MC = MeanCov<T, D>;
    using GF = StageGeom<MAT * (int)sizeof(T), G>;
    using GM = StageGeom<D * (int)sizeof(T), G>;
    const ScanArgs<T>& a = g.s;
    T* lds = sh.lds;                    // (NFILT >= NSMTH scalars per wave)
    double* lds_ll = sh.lds_ll;
    auto& stage = sh.stage;
    (void)NS;

    const long gt = (long)blockIdx.x * kBlock + threadIdx.x;
    const long k0 = gt * a.Lc;
    const long k1 = min(a.N, k0 + a.Lc);
    const int wave = threadIdx.x / kWave;
    const long wbase = ((long)blockIdx.x * kBlock + wave * kWave) * a.Lc;
    const bool staged = (wbase + (long)kWave * a.Lc <= a.N) && (a.Lc % G == 0);
    char* lP = stage[wave];
    char* lM = lP + GF::BYTES;
    const long pitchF = (long)a.Lc * MAT * sizeof(T), pitchM = (long)a.Lc * D * sizeof(T);
    const int S = a.Lc / G;

    SE right_part, ls;
    MC s_short;
    const bool has_right = (int)blockIdx.x + 1 < a.nblocks;
    const bool shortcut = a.shortcut != 0 && has_right && carry_shortcut_smoother<T, D>(a.sspine, (int)blockIdx.x + 1, s_short);
    if (has_right && !shortcut) fold_spine_partial<SE>(a.sspine, (int)blockIdx.x + 1, a.nblocks, right_part);
    ws_load(a.lsuf, a.nlanes, gt, ls);
    V4 rP[GF::NV], rM[GM::NV];
    const char* gP = reinterpret_cast<const char*>(a.fPs + wbase * MAT);
    const char* gM = reinterpret_cast<const char*>(a.fms + wbase * D);
    if (staged) {
        stage_issue<GF>(gP + (long)(S - 1) * GF::SEG, pitchF, rP);
        stage_issue<GM>(gM + (long)(S - 1) * GM::SEG, pitchM, rM);
    }
    T tnext = T(0), tcur = T(0);
    if (k0 < k1) {
        tnext = (k1 < a.N) ? g.m.ts[k1] : T(0);
        tcur = g.m.ts[k1 - 1];
    }
    MC s;
#pragma unroll
    for (int i = 0; i < D; ++i) s.m[i] = T(0);
#pragma unroll
    for (int i = 0; i < SYM; ++i) s.P[i] = T(0);
    if (shortcut) {
        s = s_short;
    } else if ((int)blockIdx.x + 1 < a.nblocks) {
        SE right;
        block_reduce_ordered(right_part, right, lds);
        smth_apply(right, s);
    }
    smth_apply(ls, s);

    if (k0 < k1) {
        char* oP = reinterpret_cast<char*>(a.sPs + wbase * MAT);
        char* oM = reinterpret_cast<char*>(a.sms + wbase * D);
        for (long k = k1 - 1; k >= k0; --k) {
            const int i = (int)((k - k0) % G);
            const long sb = (k - k0) / G;
            if (staged && i == G - 1) {
                wave_lds_sync();
                stage_commit<GF>(lP, rP);
                stage_commit<GM>(lM, rM);
                if (sb > 0) {
                    stage_issue<GF>(gP + (sb - 1) * GF::SEG, pitchF, rP);
                    stage_issue<GM>(gM + (sb - 1) * GM::SEG, pitchM, rM);
                }
                wave_lds_sync();
            }
            T mk[D], Pk[MAT];
            if (staged) {
                stage_get<GM, T, D>(lM, i, mk);
                stage_get<GF, T, MAT>(lP, i, Pk);
            } else {
                load_rec<T, D>(a.fms + k * D, mk);
                load_rec<T, MAT>(a.fPs + k * MAT, Pk);
            }
            const T t = tcur;
            if (k > 0) tcur = g.m.ts[k - 1];
            const bool last = (k == a.N - 1);
            T F[MAT], Qf[MAT];
            if (!last) lti_step<T, D>(g.m, tnext - t, F, Qf);
            smoother_apply_step<T, D>(F, Qf, mk, Pk, last, s);
            tnext = t;
            if constexpr (PROJ) {
                // posterior of f = H x at the query rows only (pssgp/model.py:107-111)
                const int q = g.qslot[k];
                if (q >= 0) {
                    T mu = T(0), var = T(0);
#pragma unroll
                    for (int r = 0; r < D; ++r) {
                        mu += g.m.H[r] * s.m[r];
#pragma unroll
                        for (int c = 0; c < D; ++c) var += g.m.H[r] * g.m.H[c] * s.P[symi<D>(r < c ? r : c, r < c ? c : r)];
                    }
                    g.pmean[q] = mu;
                    g.pvar[q] = var;
                }
                continue;
            }
            T Pf[MAT];
            full_from_sym<T, D>(s.P, Pf);
            if (staged) {
                stage_put<GM, T, D>(lM, i, s.m);
                stage_put<GF, T, MAT>(lP, i, Pf);
                if (i == 0) {
                    wave_lds_sync();
                    stage_drain<GM, NT>(oM + sb * GM::SEG, pitchM, lM);
                    stage_drain<GF, NT>(oP + sb * GF::SEG, pitchF, lP);
                }
            } else {
                store_rec<T, D>(a.sms + k * D, s.m);
                store_rec<T, MAT>(a.sPs + k * MAT, Pf);
            }
        }
    }
    if (blockIdx.x == 0 && a.ll != nullptr) {
        double v = 0.0;
        for (int b = threadIdx.x; b < a.nblocks; b += kBlock) v += a.llpart[b];
        const double t = block_sum_double(v, lds_ll);
        if (threadIdx.x == 0) *a.ll = t;
    }
}

template <typename T, int D, bool NT, bool PROJ = false>
__global__ __launch_bounds__(kBlock) void k_gp_smooth(const GpArgs<T> g) {
    __shared__ GpLds<T, D> sh;
    gp_smooth_body<T, D, NT, PROJ>(g, sh);
}

// ---------------------------------------------------------------------------------------------
// ONE launch for a short series (round 3): the reference's own lengths are N = 200 .. 10^4 per call
// (pssgp/experiments/toy_models/mcmc.py:55, speed_and_stability.py:73), where three dependent launches of a few
// workgroups each cost more in launch gaps than in work.  A single workgroup runs the three bodies above one after
// the other -- its lanes own up to kOneLaunchSteps consecutive steps each, nblocks = 1, so there is no spine to fold and
// the "inter-workgroup" scratch (lpre, lsuf, the filtered moments) is written and read back by the same CU behind a
// workgroup barrier.  SMOOTH: 0 = log-likelihood (and filtered moments when asked for), 1 = + smoother, 2 = + smoother
// in projection mode (predict_f).
// ---------------------------------------------------------------------------------------------
template <typename T, int D, int SMOOTH>
__global__ __launch_bounds__(kBlock) void k_gp_one(const GpArgs<T> g) {
    __shared__ GpLds<T, D> sh;
    gp_reduce_body<T, D>(g, sh.lds);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    gp_apply_body<T, D, SMOOTH != 0, false>(g, sh);
    if constexpr (SMOOTH != 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        gp_smooth_body<T, D, false, SMOOTH == 2>(g, sh);
    } else {
        if (threadIdx.x == 0 && g.s.ll != nullptr) *g.s.ll = g.s.llpart[0];      // (this lane wrote the partial)
    }
}

}  // namespace pgps
