// pgps_grad.hip.h -- log-likelihood AND its gradient with respect to the hyper-parameters, fused path.
//
// The two log-likelihood launches of pgps_fused.hip.h (reduce, apply without stores) instantiated on
// dual numbers (pgps_dual.h): the model (lambda, N, N^2/2, Pinf, H, R) arrives as duals -- value and
// NP partial derivatives, formed on the host from the kernel's get_sde() -- and every filtering
// element, every application of the associative operator and every log-likelihood term carries its
// derivatives along.  Replaces what the reference gets from TensorFlow autodiff through
// tfp.math.scan_associative (tests/test_gp_vs_kfs.py:53-78; consumed by the L-BFGS / HMC drivers,
// pssgp/experiments/*).  SURVEY.md section 8f, rank 1 -- for the Matern family (d <= 2 here).
#pragma once

#include "pgps_kernels.hip.h"
#include "pgps_dual.h"

namespace pgps {

template <int NP>
__device__ __forceinline__ Dual<NP> wshfl_up(const Dual<NP>& x, int s) {
    Dual<NP> r;
    r.v = __shfl_up(x.v, s, kWave);
#pragma unroll
    for (int i = 0; i < NP; ++i) r.d[i] = __shfl_up(x.d[i], s, kWave);
    return r;
}
template <int NP>
__device__ __forceinline__ Dual<NP> wshfl_down(const Dual<NP>& x, int s) {
    Dual<NP> r;
    r.v = __shfl_down(x.v, s, kWave);
#pragma unroll
    for (int i = 0; i < NP; ++i) r.d[i] = __shfl_down(x.d[i], s, kWave);
    return r;
}

template <int NP>
struct GradModel {              // everything a dual: [value | d/dtheta_1 .. d/dtheta_NP]
    Dual<NP> lam;
    Dual<NP> N1[9];             // row-major d x d in the leading d*d entries
    Dual<NP> N2[9];             // N^2 / 2 (d = 3 only)
    Dual<NP> Pinf[9];
    Dual<NP> H[3];
    Dual<NP> R;
    const double* ts;
    const double* ys;
    double t_prev;
    long N;
    int Lc, nblocks;
    long nlanes;
    Dual<NP>* spine;            // (nblocks, NFILT)
    Dual<NP>* lpre;             // (NFILT, nlanes)
    Dual<NP>* llpart;           // (nblocks,)
    double* out;                // (1 + NP): ll, d ll / d theta
};

template <int NP, int D>
__device__ __forceinline__ void grad_prior(const GradModel<NP>& m, Dual<NP>* h, Dual<NP>* P0) {
#pragma unroll
    for (int i = 0; i < D; ++i) h[i] = m.H[i];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = i; j < D; ++j) P0[symi<D>(i, j)] = Dual<NP>(0.5) * (m.Pinf[i * D + j] + m.Pinf[j * D + i]);
}

// d = 3 differentiates one direction per model (registers): the directions run side by side, blockIdx.y picks the model
template <int NP>
struct GradPack { GradModel<NP> m[3]; };

template <int NP, int D>
__device__ __forceinline__ void grad_reduce_body(const GradModel<NP>& m) {
    using T = Dual<NP>;
    constexpr int MAT = D * D, SYM = Dim<D>::SYM;
    using FE = FiltElem<T, D>;
    __shared__ T lds[kWaves * Dim<D>::NFILT];
    T h[D], P0[SYM];
    grad_prior<NP, D>(m, h, P0);
    const long gt = (long)blockIdx.x * kBlock + threadIdx.x;
    const long k0 = gt * m.Lc;
    const long k1 = min(m.N, k0 + m.Lc);
    FE agg;
    filt_identity(agg);
    if (k0 < k1) {
        double tprev = (k0 > 0) ? m.ts[k0 - 1] : m.t_prev;
        double tn = m.ts[k0], yn = m.ys[k0];
        for (long k = k0; k < k1; ++k) {
            const double t = tn, y = yn;
            if (k + 1 < k1) { tn = m.ts[k + 1]; yn = m.ys[k + 1]; }
            if (k == 0) {
                filt_first(agg, P0, T(y), h, m.R);
            } else {
                T F[MAT], Q[SYM];
                lti_step_dual<NP, D>(m.lam, m.N1, m.N2, m.Pinf, t - tprev, F, Q);
                filt_extend(agg, F, Q, T(y), h, m.R);
            }
            tprev = t;
        }
    }
    FE excl, total;
    block_scan_exclusive<FE, true>(agg, excl, total, lds);
    ws_store(m.lpre, m.nlanes, gt, excl);
    if (threadIdx.x == 0) rec_store(m.spine + (long)blockIdx.x * Dim<D>::NFILT, total);
}
template <int NP, int D>
__global__ __launch_bounds__(kBlock) void k_grad_reduce(const GradModel<NP> m) { grad_reduce_body<NP, D>(m); }
template <int NP, int D>
__global__ __launch_bounds__(kBlock) void k_grad_reduce_pack(const GradPack<NP> p) { grad_reduce_body<NP, D>(p.m[blockIdx.y]); }

template <int NP, int D>
__device__ __forceinline__ void grad_apply_body(const GradModel<NP>& m, Dual<NP>* block_total = nullptr) {
    using T = Dual<NP>;
    constexpr int MAT = D * D, SYM = Dim<D>::SYM, NF = Dim<D>::NFILT;
    using FE = FiltElem<T, D>;
    using MC = MeanCov<T, D>;
    __shared__ T lds[kWaves * NF];
    __shared__ double lds_ll[kWaves];
    T h[D];
    MC s;
    grad_prior<NP, D>(m, h, s.P);
#pragma unroll
    for (int i = 0; i < D; ++i) s.m[i] = T(0.0);
    const long gt = (long)blockIdx.x * kBlock + threadIdx.x;
    const long k0 = gt * m.Lc;
    const long k1 = min(m.N, k0 + m.Lc);
    if (blockIdx.x > 0) {
        FE left;
        fold_spine<FE>(m.spine, 0, (int)blockIdx.x, left, lds);
        filt_apply(s, left);
    }
    {
        FE lp;
        ws_load(m.lpre, m.nlanes, gt, lp);
        filt_apply(s, lp);
    }
    LogLikDual<NP> ll;
    if (k0 < k1) {
        double tprev = (k0 > 0) ? m.ts[k0 - 1] : m.t_prev;
        double tn = m.ts[k0], yn = m.ys[k0];
        for (long k = k0; k < k1; ++k) {
            const double t = tn, y = yn;
            if (k + 1 < k1) { tn = m.ts[k + 1]; yn = m.ys[k + 1]; }
            T F[MAT], Q[SYM], mp[D], Pp[SYM], FP[MAT];
            lti_step_dual<NP, D>(m.lam, m.N1, m.N2, m.Pinf, t - tprev, F, Q);
            tprev = t;
            kf_step(s, F, Q, T(y), h, m.R, k == 0, ll, mp, Pp, FP);
        }
    }
    const T v = ll.value();
    T tot;
    tot.v = block_sum_double(v.v, lds_ll);
#pragma unroll
    for (int i = 0; i < NP; ++i) tot.d[i] = block_sum_double(v.d[i], lds_ll);
    if (threadIdx.x == 0) m.llpart[blockIdx.x] = tot;
    if (block_total) *block_total = tot;
}
template <int NP, int D>
__global__ __launch_bounds__(kBlock) void k_grad_apply(const GradModel<NP> m) { grad_apply_body<NP, D>(m); }
template <int NP, int D>
__global__ __launch_bounds__(kBlock) void k_grad_apply_pack(const GradPack<NP> p) { grad_apply_body<NP, D>(p.m[blockIdx.y]); }

// A series that fits ONE workgroup (the reference's own lengths): reduce, Kalman pass and the sum in one launch, one
// derivative direction per workgroup (blockIdx.y); direction p's workgroup writes d ll / d theta_p -- and direction 0's
// the log-likelihood -- where the caller wants them.  Three launch boundaries and the compaction kernel less.
template <int D>
__global__ __launch_bounds__(kBlock) void k_grad_one_pack(const GradPack<1> p, double* out) {
    const GradModel<1>& m = p.m[blockIdx.y];
    grad_reduce_body<1, D>(m);
    __builtin_amdgcn_s_waitcnt(0);              // this lane's scan record is in memory before it is read back
    __syncthreads();
    Dual<1> tot;
    grad_apply_body<1, D>(m, &tot);
    if (threadIdx.x == 0) {
        if (blockIdx.y == 0) out[0] = tot.v;
        out[1 + blockIdx.y] = tot.d[0];
    }
}

template <int NP>
__device__ __forceinline__ void grad_finalize_body(const Dual<NP>* llpart, int nblocks, double* out) {
    __shared__ double lds_ll[kWaves];
    for (int c = 0; c <= NP; ++c) {
        double v = 0.0;
        for (int b = threadIdx.x; b < nblocks; b += kBlock) v += (c == 0) ? llpart[b].v : llpart[b].d[c - 1];
        const double t = block_sum_double(v, lds_ll);
        if (threadIdx.x == 0) out[c] = t;
    }
}
template <int NP>
__global__ __launch_bounds__(kBlock) void k_grad_finalize(const Dual<NP>* llpart, int nblocks, double* out) {
    grad_finalize_body<NP>(llpart, nblocks, out);
}
template <int NP>
__global__ __launch_bounds__(kBlock) void k_grad_finalize_pack(const GradPack<NP> p) {
    const GradModel<NP>& m = p.m[blockIdx.y];
    grad_finalize_body<NP>(m.llpart, m.nblocks, m.out);
}

// d = 3: np single-direction passes left (ll, d ll / d theta_p) pairs; gather them as [ll, grad...]
static __global__ void k_grad_compact(const double* pairs, int np, double* out) {
    const int i = threadIdx.x;
    if (i == 0) out[0] = pairs[0];
    if (i >= 1 && i <= np) out[i] = pairs[2 * (i - 1) + 1];
}

}  // namespace pgps
