// pgps_gradb.hip -- log-likelihood and its EXACT gradient for composite kernels whose drift is block diagonal with
// blocks  F_b = -lam_b I + N_b,  N_b nilpotent: sums and products of Matern kernels (a sum is one block per summand; a
// product of Matern kernels is ONE block with lam = the sum of the factors' and N their Kronecker sum -- the
// Kronecker sum of commuting nilpotent matrices is nilpotent).  Balancing is a diagonal similarity and keeps that form.
// The transition matrix is then closed-form per block,  expm(dt F_b) = exp(-lam_b dt) sum_k dt^k N_b^k / k!  (k <= 3
// for state dimensions up to 6), Q = Pinf - F Pinf F^T, so the forward-mode dual numbers of pgps_grad.hip.h run through
// the same two launches (reduce, apply) at state dimensions 2 .. 6: what the reference gets from TensorFlow autodiff for
// `Matern32 + Matern52` and `Matern32 * Matern52` (tests/test_gp_vs_kfs.py:40-41,53-78).  One direction per pass (a
// dual filtering element at d = 6 is already 180 doubles per lane); compiled one unit per d (-DPGPS_GRADB_D).
#include "pgps_grad.hip.h"

#ifndef PGPS_GRADB_D
#error "compile with -DPGPS_GRADB_D=<d>"
#endif

namespace pgps {

constexpr int kGbMaxBlocks = 4;
constexpr int kGbMaxD = 6;

struct GradModelB {
    using T = Dual<1>;
    T lam[kGbMaxBlocks];
    int bid[kGbMaxD];                      // block of state i
    T N1[kGbMaxD * kGbMaxD];               // N      (block diagonal, row-major d x d)
    T N2[kGbMaxD * kGbMaxD];               // N^2 / 2
    T N3[kGbMaxD * kGbMaxD];               // N^3 / 6
    T Pinf[kGbMaxD * kGbMaxD];
    T H[kGbMaxD];
    T R;
    int nblk;
    const double* ts;
    const double* ys;
    double t_prev;
    long N;
    int Lc, nblocks;
    long nlanes;
    T* spine;
    T* lpre;
    T* llpart;
    double* out;                           // (2): ll, d ll / d theta_p
};

template <int D>
__device__ __forceinline__ void gb_step(const GradModelB& m, double dt, Dual<1>* F, Dual<1>* Q /*sym*/) {
    using T = Dual<1>;
    constexpr int MAT = D * D;
    T e[kGbMaxBlocks];
#pragma unroll
    for (int b = 0; b < kGbMaxBlocks; ++b) e[b] = b < m.nblk ? exp(-(m.lam[b] * T(dt))) : T(0.0);
    T X[MAT];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        T ei = e[0];
#pragma unroll
        for (int b = 1; b < kGbMaxBlocks; ++b) ei = (m.bid[i] == b) ? e[b] : ei;
#pragma unroll
        for (int j = 0; j < D; ++j) {
            T v = T(i == j ? 1.0 : 0.0);
            v += T(dt) * m.N1[i * D + j];
            v += T(dt * dt) * m.N2[i * D + j];
            v += T(dt * dt * dt) * m.N3[i * D + j];
            F[i * D + j] = ei * v;                // (entries outside the diagonal blocks are exact zeros: N is block diagonal)
        }
    }
    mat_mul<T, D>(F, m.Pinf, X);
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = i; j < D; ++j) {
            T acc = T(0.0), acct = T(0.0);
#pragma unroll
            for (int l = 0; l < D; ++l) { acc += X[i * D + l] * F[j * D + l]; acct += X[j * D + l] * F[i * D + l]; }
            Q[symi<D>(i, j)] = T(0.5) * (m.Pinf[i * D + j] + m.Pinf[j * D + i]) - T(0.5) * (acc + acct);
        }
}

template <int D>
__device__ __forceinline__ void gb_prior(const GradModelB& m, Dual<1>* h, Dual<1>* P0) {
#pragma unroll
    for (int i = 0; i < D; ++i) h[i] = m.H[i];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = i; j < D; ++j) P0[symi<D>(i, j)] = Dual<1>(0.5) * (m.Pinf[i * D + j] + m.Pinf[j * D + i]);
}

template <int D>
__global__ __launch_bounds__(kBlock) void k_gradb_reduce(const GradModelB m) {
    using T = Dual<1>;
    constexpr int MAT = D * D, SYM = Dim<D>::SYM;
    using FE = FiltElem<T, D>;
    __shared__ T lds[kWaves * Dim<D>::NFILT];
    T h[D], P0[SYM];
    gb_prior<D>(m, h, P0);
    const long gt = (long)blockIdx.x * kBlock + threadIdx.x;
    const long k0 = gt * m.Lc;
    const long k1 = min(m.N, k0 + m.Lc);
    FE agg;
    filt_identity(agg);
    if (k0 < k1) {
        double tprev = (k0 > 0) ? m.ts[k0 - 1] : m.t_prev;
        for (long k = k0; k < k1; ++k) {
            const double t = m.ts[k], y = m.ys[k];
            if (k == 0) {
                filt_first(agg, P0, T(y), h, m.R);
            } else {
                T F[MAT], Q[SYM];
                gb_step<D>(m, t - tprev, F, Q);
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

template <int D>
__global__ __launch_bounds__(kBlock) void k_gradb_apply(const GradModelB m) {
    using T = Dual<1>;
    constexpr int MAT = D * D, SYM = Dim<D>::SYM, NF = Dim<D>::NFILT;
    using FE = FiltElem<T, D>;
    using MC = MeanCov<T, D>;
    __shared__ T lds[kWaves * NF];
    __shared__ double lds_ll[kWaves];
    T h[D];
    MC s;
    gb_prior<D>(m, h, s.P);
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
    LogLikDual<1> ll;
    if (k0 < k1) {
        double tprev = (k0 > 0) ? m.ts[k0 - 1] : m.t_prev;
        for (long k = k0; k < k1; ++k) {
            const double t = m.ts[k], y = m.ys[k];
            T F[MAT], Q[SYM], mp[D], Pp[SYM], FP[MAT];
            gb_step<D>(m, t - tprev, F, Q);
            tprev = t;
            kf_step(s, F, Q, T(y), h, m.R, k == 0, ll, mp, Pp, FP);
        }
    }
    const T v = ll.value();
    T tot;
    tot.v = block_sum_double(v.v, lds_ll);
    tot.d[0] = block_sum_double(v.d[0], lds_ll);
    if (threadIdx.x == 0) m.llpart[blockIdx.x] = tot;
}

// model: (1 + np) consecutive rows [lam (kGbMaxBlocks) | N (d*d) | Pinf (d*d) | H (d) | R], row 0 the values, row p the
// partial derivatives with respect to hyper-parameter p; bsize: the nblk block sizes (sum = d).
// out_dev: 1 + np doubles, followed by 2 * np doubles of scratch (pass p leaves (ll, d ll / d theta_p) there).
template <int D>
int launch_gradb(pgps_ctx* ctx, long N, int nblk, const int* bsize, int np, const double* model, const double* ts, double t0,
                 const double* ys, double* out_dev) {
    using T = Dual<1>;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int stride = kGbMaxBlocks + 2 * D * D + D + 1;
    double* scratch = out_dev + 1 + np;
    for (int p = 0; p < np; ++p) {
        GradModelB m{};
        auto get = [&](int off) { T x(model[off]); x.d[0] = model[(p + 1) * stride + off]; return x; };
        m.nblk = nblk;
        for (int b = 0; b < kGbMaxBlocks; ++b) m.lam[b] = b < nblk ? get(b) : T(0.0);
        for (int i = 0, b = 0, left = bsize[0]; i < D; ++i) {
            while (left == 0 && b + 1 < nblk) { ++b; left = bsize[b]; }
            m.bid[i] = b;
            --left;
        }
        for (int i = 0; i < kGbMaxD * kGbMaxD; ++i) { m.N1[i] = T(0.0); m.N2[i] = T(0.0); m.N3[i] = T(0.0); m.Pinf[i] = T(0.0); }
        for (int i = 0; i < kGbMaxD; ++i) m.H[i] = T(0.0);
        for (int i = 0; i < D * D; ++i) { m.N1[i] = get(kGbMaxBlocks + i); m.Pinf[i] = get(kGbMaxBlocks + D * D + i); }
        for (int i = 0; i < D; ++i) m.H[i] = get(kGbMaxBlocks + 2 * D * D + i);
        m.R = get(kGbMaxBlocks + 2 * D * D + D);
        // N^2 / 2 and N^3 / 6 with their derivatives (product rule through the dual arithmetic)
        T P2[D * D];
        for (int i = 0; i < D; ++i)
            for (int j = 0; j < D; ++j) {
                T acc(0.0);
                for (int l = 0; l < D; ++l) acc += m.N1[i * D + l] * m.N1[l * D + j];
                P2[i * D + j] = acc;
                m.N2[i * D + j] = T(0.5) * acc;
            }
        for (int i = 0; i < D; ++i)
            for (int j = 0; j < D; ++j) {
                T acc(0.0);
                for (int l = 0; l < D; ++l) acc += P2[i * D + l] * m.N1[l * D + j];
                m.N3[i * D + j] = T(1.0 / 6.0) * acc;
            }
        m.ts = ts; m.ys = ys; m.t_prev = t0; m.N = N; m.out = scratch + 2 * p;
        // 8 steps per lane at most: the dual elements live in scratch memory at these sizes, the lane-serial part is
        // the cheap one, and the realistic series (1e3 .. 1e5 steps) then still spread over the chip
        long v = 8;
        while (v > 1 && (long)kBlock * v * 64 > N) v /= 2;
        m.Lc = (int)v;
        m.nblocks = (int)((N + (long)kBlock * m.Lc - 1) / ((long)kBlock * m.Lc));
        m.nlanes = (long)m.nblocks * kBlock;
        const size_t nb = (size_t)m.nblocks, nl = (size_t)m.nlanes;
        auto up = [](size_t x) { return (x + 255) / 256 * 256; };
        size_t off = 0;
        const size_t o_spine = off; off = up(off + nb * Dim<D>::NFILT * sizeof(T));
        const size_t o_lpre = off;  off = up(off + nl * Dim<D>::NFILT * sizeof(T));
        const size_t o_ll = off;    off = up(off + nb * sizeof(T));
        int rc = ensure(ctx, ctx->ws, off);
        if (rc) return rc;
        char* base = (char*)ctx->ws.p;
        m.spine = (T*)(base + o_spine); m.lpre = (T*)(base + o_lpre); m.llpart = (T*)(base + o_ll);
        const dim3 grid(m.nblocks), block(kBlock);
        timed_launch(ctx, PGPS_K_FILTER_REDUCE, k_gradb_reduce<D>, grid, block, 0, m);
        timed_launch(ctx, PGPS_K_FILTER_APPLY, k_gradb_apply<D>, grid, block, 0, m);
        timed_launch(ctx, PGPS_K_LL_FINALIZE, k_grad_finalize<1>, dim3(1), block, 0, (const T*)m.llpart, m.nblocks, m.out);
    }
    k_grad_compact<<<1, 64, 0, ctx->stream>>>(scratch, np, out_dev);
    HIPCHK(ctx, hipGetLastError());
    return PGPS_OK;
}

template int launch_gradb<PGPS_GRADB_D>(pgps_ctx*, long, int, const int*, int, const double*, const double*, double,
                                        const double*, double*);

}  // namespace pgps
