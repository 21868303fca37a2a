// pgps_gpadj.hip.h -- log-likelihood AND the adjoints of the model, fused (closed-form discretisation) path, d <= 3.
//
// The adjoint pass of DESIGN.md section 4l in the lane-chunk layout of pgps_fused.hip.h: what pgps_rcgrad.hip.h does on
// 16-lane rows for the general-LTI kernels, for the Matern family whose transition matrices are formed in registers from
// the time stamps.  Replaces the dual-number pass (pgps_grad.hip.h: one extra set of derivatives per hyper-parameter
// through every operator of the scan) by one filter pass and one reverse pass whatever the number of parameters --
// what the reference gets from TensorFlow autodiff through tfp.math.scan_associative (tests/test_gp_vs_kfs.py:53-78).
//
//   k_gp_reduce   (pgps_fused.hip.h)  chunk totals of the filtering elements, workgroup scan, spine
//   k_gp_gfwd     every lane filters its chunk from the state entering it (log-likelihood terms as in k_gp_apply), keeps
//                 the state ENTERING every step in a lane-major scratch, and folds the steps' ADJOINT elements
//                     E_k = A_k^T,  g_k = v_k r_k / s_k,  L_k = -v_k v_k^T / (2 s_k),   A_k = (I - K_k H) F_k,  v_k = F_k^T H^T
//                 (missing observation: E_k = F_k^T, g_k = 0, L_k = 0) into its chunk's total with the smoothing operator
//                 (parallel.py:176-184): with a_k = d ll / d m_k, B_k = d ll / d P_k and W = B - a a^T / 2 the reverse sweep
//                 of the Kalman filter reads  a_{k-1} = E_k a_k + g_k,  W_{k-1} = E_k W_k E_k^T + L_k;  workgroup suffix scan,
//                 spine.
//   k_gp_gback    the suffix of the totals applied to (0, 0) is (a, W) behind the lane's chunk; the lane walks its steps
//                 backwards, recomputes each step from the state kept by k_gp_gfwd and accumulates
//                     Abar += dt [mpbar mp^T + 2 Ppbar (Pp - Pinf)]     Ubar += ubar
//                     Hbar += sbar u + Pp ubar - rbar mp                Rbar += sbar
//                 (oracle/np_grad.py states the recursion); workgroup sums -> gpart.
//   k_grad_lti_finalize (pgps_gradlti.h)  out = [ll | Abar (d^2) | Ubar (d) | Hbar (d) | Rbar], chunk partials in a fixed order.
// A short series runs the three bodies in ONE launch of one workgroup (k_gp_gone), as k_gp_one does.
// fp64 only (the host contracts the adjoints with the model's derivatives: pssgp/_backend.py contract_grad_stats).
#pragma once

#include "pgps_fused.hip.h"

namespace pgps {

struct GpAdjArgs {
    GpArgs<double> g;
    double* xs;             // ((d + sym) Lc, nlanes): the filtered state entering every step, lane-major
    double* gpart;          // (nblocks, d^2 + 2 d + 1)
    double* out;            // (1 + d^2 + 2 d + 1)
};

template <int D>
constexpr int gp_adj_nstat() { return D * D + 2 * D + 1; }

// one step's quantities from the state entering it
template <int D>
struct AdjStep {
    double F[D * D], mp[D], Pp[Dim<D>::SYM], u[D], K[D];
    double S, r, inv;
    bool obs;
};

template <int D>
__device__ __forceinline__ void adj_step(const GpModel<double>& m, double dt, const MeanCov<double, D>& s, double y,
                                         const double* h, double R, AdjStep<D>& o) {
    constexpr int MAT = D * D, SYM = Dim<D>::SYM;
    double Qf[MAT], Q[SYM], FP[MAT];
    lti_step<double, D>(m, dt, o.F, Qf);
    sym_from_full<double, D>(Qf, Q);
    mat_vec<double, D>(o.F, s.m, o.mp);
    predict_cov<double, D>(o.F, s.P, Q, FP, o.Pp);
    sym_vec<double, D>(o.Pp, h, o.u);
    double S = R, mu = 0.0;
#pragma unroll
    for (int i = 0; i < D; ++i) { S += h[i] * o.u[i]; mu += h[i] * o.mp[i]; }
    o.obs = !is_nan(y);
    o.S = S;
    o.inv = o.obs ? recip(S) : 0.0;
    o.r = o.obs ? y - mu : 0.0;
#pragma unroll
    for (int i = 0; i < D; ++i) o.K[i] = o.u[i] * o.inv;
}

// ---------------------------------------------------------------------------------------------
// forward: filter, log-likelihood, kept states, adjoint elements
// ---------------------------------------------------------------------------------------------
template <int D>
__device__ __forceinline__ void gp_gfwd_body(const GpAdjArgs& ga, GpLds<double, D>& sh) {
    using T = double;
    constexpr int MAT = D * D, SYM = Dim<D>::SYM, NX = D + SYM;
    using FE = FiltElem<T, D>;
    using SE = SmthElem<T, D>;
    using MC = MeanCov<T, D>;
    const GpArgs<T>& g = ga.g;
    const ScanArgs<T>& a = g.s;
    T* lds = sh.lds;
    double* lds_ll = sh.lds_ll;

    T h[D];
    MC s;
    gp_prior<T, D>(g.m, h, s.P);
#pragma unroll
    for (int i = 0; i < D; ++i) s.m[i] = T(0);

    const long gt = (long)blockIdx.x * kBlock + threadIdx.x;
    const long k0 = gt * a.Lc;
    const long k1 = min(a.N, k0 + a.Lc);

    FE left_part, lp;
    if (blockIdx.x > 0) fold_spine_partial<FE>(a.spine, 0, (int)blockIdx.x, left_part);
    ws_load(a.lpre, a.nlanes, gt, lp);
    if (blockIdx.x > 0) {
        FE left;
        block_reduce_ordered(left_part, left, lds);
        filt_apply(s, left);
    }
    filt_apply(s, lp);

    LogLik ll;
    SE agg;
    smth_identity(agg);
    if (k0 < k1) {
        T tprev = (k0 > 0) ? g.m.ts[k0 - 1] : g.m.t_prev;
        T tn = g.m.ts[k0], yn = a.ys[k0];
        for (long k = k0; k < k1; ++k) {
            const T t = tn, y = yn;
            if (k + 1 < k1) { tn = g.m.ts[k + 1]; yn = a.ys[k + 1]; }
            // the state entering the step: what the reverse pass starts the step from
            {
                double* x = ga.xs + (long)(k - k0) * NX * a.nlanes + gt;
#pragma unroll
                for (int i = 0; i < D; ++i) x[(long)i * a.nlanes] = s.m[i];
#pragma unroll
                for (int i = 0; i < SYM; ++i) x[(long)(D + i) * a.nlanes] = s.P[i];
            }
            AdjStep<D> st;
            adj_step<D>(g.m, t - tprev, s, y, h, a.R, st);
            tprev = t;
            if (st.obs) ll.add(st.r, st.S);
            // filtered state of the step
#pragma unroll
            for (int i = 0; i < D; ++i) s.m[i] = st.mp[i] + st.K[i] * st.r;
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = i; j < D; ++j) s.P[symi<D>(i, j)] = st.Pp[symi<D>(i, j)] - st.u[i] * st.u[j] * st.inv;
            // adjoint element of the step, folded into the chunk's total on its right (time order)
            SE e, r;
            T v[D];
#pragma unroll
            for (int j = 0; j < D; ++j) {
                T acc = T(0);
#pragma unroll
                for (int i = 0; i < D; ++i) acc += h[i] * st.F[i * D + j];
                v[j] = acc;
            }
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j) e.E[j * D + i] = st.F[i * D + j] - st.K[i] * v[j];      // E = A^T
#pragma unroll
            for (int j = 0; j < D; ++j) e.g[j] = v[j] * (st.r * st.inv);
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = i; j < D; ++j) e.L[symi<D>(i, j)] = T(-0.5) * v[i] * v[j] * st.inv;
            smth_combine(agg, e, r);
            agg = r;
        }
    }
    {
        const double v = ll.value();
        const double t = block_sum_double(v, lds_ll);
        if (threadIdx.x == 0) a.llpart[blockIdx.x] = t;
    }
    SE excl, total;
    block_scan_exclusive<SE, false>(agg, excl, total, lds);
    ws_store(a.lsuf, a.nlanes, gt, excl);
    if (threadIdx.x == 0) rec_store(a.sspine + (long)blockIdx.x * Dim<D>::NSMTH, total);
    (void)MAT;
}

template <int D>
__global__ __launch_bounds__(kBlock) void k_gp_gfwd(const GpAdjArgs ga) {
    __shared__ GpLds<double, D> sh;
    gp_gfwd_body<D>(ga, sh);
}

// ---------------------------------------------------------------------------------------------
// backward: the reverse sweep
// ---------------------------------------------------------------------------------------------
template <int D>
__device__ __forceinline__ void gp_gback_body(const GpAdjArgs& ga, GpLds<double, D>& sh) {
    using T = double;
    constexpr int MAT = D * D, SYM = Dim<D>::SYM, NX = D + SYM, NST = gp_adj_nstat<D>();
    using SE = SmthElem<T, D>;
    using MC = MeanCov<T, D>;
    const GpArgs<T>& g = ga.g;
    const ScanArgs<T>& a = g.s;
    T* lds = sh.lds;
    double* lds_ll = sh.lds_ll;

    T h[D], Pinf[SYM];
    gp_prior<T, D>(g.m, h, Pinf);
    const long gt = (long)blockIdx.x * kBlock + threadIdx.x;
    const long k0 = gt * a.Lc;
    const long k1 = min(a.N, k0 + a.Lc);

    SE right_part, ls;
    if ((int)blockIdx.x + 1 < a.nblocks) fold_spine_partial<SE>(a.sspine, (int)blockIdx.x + 1, a.nblocks, right_part);
    ws_load(a.lsuf, a.nlanes, gt, ls);
    MC z;                       // (a, W) behind the chunk
#pragma unroll
    for (int i = 0; i < D; ++i) z.m[i] = T(0);
#pragma unroll
    for (int i = 0; i < SYM; ++i) z.P[i] = T(0);
    if ((int)blockIdx.x + 1 < a.nblocks) {
        SE right;
        block_reduce_ordered(right_part, right, lds);
        smth_apply(right, z);
    }
    smth_apply(ls, z);
    T av[D], B[SYM];
#pragma unroll
    for (int i = 0; i < D; ++i) av[i] = z.m[i];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = i; j < D; ++j) B[symi<D>(i, j)] = z.P[symi<D>(i, j)] + T(0.5) * av[i] * av[j];

    T st_[NST];
#pragma unroll
    for (int i = 0; i < NST; ++i) st_[i] = T(0);
    T* Abar = st_;
    T* Ubar = st_ + MAT;
    T* Hbar = st_ + MAT + D;
    T& Rbar = st_[MAT + 2 * D];

    if (k0 < k1) {
        T tcur = g.m.ts[k1 - 1];
        for (long k = k1 - 1; k >= k0; --k) {
            const T t = tcur;
            const T tp = (k > 0) ? g.m.ts[k - 1] : g.m.t_prev;
            tcur = tp;
            const T y = a.ys[k];
            MC s;
            {
                const double* x = ga.xs + (long)(k - k0) * NX * a.nlanes + gt;
#pragma unroll
                for (int i = 0; i < D; ++i) s.m[i] = x[(long)i * a.nlanes];
#pragma unroll
                for (int i = 0; i < SYM; ++i) s.P[i] = x[(long)(D + i) * a.nlanes];
            }
            const T dt = t - tp;
            AdjStep<D> q;
            adj_step<D>(g.m, dt, s, y, h, a.R, q);
            T BK[D], ubar[D], mpbar[D], Ppbar[SYM];
            sym_vec<T, D>(B, q.K, BK);
            T aK = T(0), KBK = T(0);
#pragma unroll
            for (int i = 0; i < D; ++i) { aK += av[i] * q.K[i]; KBK += q.K[i] * BK[i]; }
            const T ri = q.r * q.inv;
            const T sbar = q.obs ? (-aK * ri + KBK - T(0.5) * q.inv + T(0.5) * ri * ri) : T(0);
            const T rbar = q.obs ? (aK - ri) : T(0);
#pragma unroll
            for (int i = 0; i < D; ++i) {
                ubar[i] = q.obs ? (av[i] * ri - T(2) * BK[i] + sbar * h[i]) : T(0);
                mpbar[i] = av[i] - rbar * h[i];
            }
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = i; j < D; ++j) Ppbar[symi<D>(i, j)] = B[symi<D>(i, j)] + T(0.5) * (ubar[i] * h[j] + h[i] * ubar[j]);
            // Abar += dt [mpbar mp^T + 2 Ppbar (Pp - Pinf)]
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    T acc = mpbar[i] * q.mp[j];
#pragma unroll
                    for (int l = 0; l < D; ++l) {
                        const int il = symi<D>(i < l ? i : l, i < l ? l : i), lj = symi<D>(l < j ? l : j, l < j ? j : l);
                        acc += T(2) * Ppbar[il] * (q.Pp[lj] - Pinf[lj]);
                    }
                    Abar[i * D + j] += dt * acc;
                }
            T Ppu[D];
            sym_vec<T, D>(q.Pp, ubar, Ppu);
#pragma unroll
            for (int i = 0; i < D; ++i) {
                Ubar[i] += ubar[i];
                Hbar[i] += sbar * q.u[i] + Ppu[i] - rbar * q.mp[i];
            }
            Rbar += sbar;
            // (a, B) of the step before: F^T mpbar, sym(F^T Ppbar F)
            T X[MAT];
#pragma unroll
            for (int i = 0; i < D; ++i) {
                T acc = T(0);
#pragma unroll
                for (int l = 0; l < D; ++l) acc += q.F[l * D + i] * mpbar[l];
                av[i] = acc;
            }
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    T acc = T(0);
#pragma unroll
                    for (int l = 0; l < D; ++l) acc += Ppbar[symi<D>(i < l ? i : l, i < l ? l : i)] * q.F[l * D + j];
                    X[i * D + j] = acc;                                         // Ppbar F
                }
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = i; j < D; ++j) {
                    T acc = T(0), acct = T(0);
#pragma unroll
                    for (int l = 0; l < D; ++l) { acc += q.F[l * D + i] * X[l * D + j]; acct += q.F[l * D + j] * X[l * D + i]; }
                    B[symi<D>(i, j)] = T(0.5) * (acc + acct);
                }
        }
    }
#pragma unroll
    for (int i = 0; i < NST; ++i) {
        const double t = block_sum_double(st_[i], lds_ll);
        if (threadIdx.x == 0) ga.gpart[(long)blockIdx.x * NST + i] = t;
    }
}

template <int D>
__global__ __launch_bounds__(kBlock) void k_gp_gback(const GpAdjArgs ga) {
    __shared__ GpLds<double, D> sh;
    gp_gback_body<D>(ga, sh);
}

// ONE launch of one workgroup for a short series (nblocks = 1: no spines to fold, the scratch is written and read back
// by the same CU behind a workgroup barrier)
template <int D>
__global__ __launch_bounds__(kBlock) void k_gp_gone(const GpAdjArgs ga) {
    __shared__ GpLds<double, D> sh;
    gp_reduce_body<double, D>(ga.g, sh.lds);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    gp_gfwd_body<D>(ga, sh);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    gp_gback_body<D>(ga, sh);
    constexpr int NST = gp_adj_nstat<D>();
    if (threadIdx.x == 0) {                     // (this lane wrote the partials)
        ga.out[0] = ga.g.s.llpart[0];
#pragma unroll
        for (int i = 0; i < NST; ++i) ga.out[1 + i] = ga.gpart[i];
    }
}

}  // namespace pgps
