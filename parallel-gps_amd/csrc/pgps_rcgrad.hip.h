// pgps_rcgrad.hip.h -- the two level-1 kernels of the adjoint log-likelihood gradient (pgps_gradlti.h) in the
// row-cooperative family: fp64, 2 <= d <= 16, one 16-lane DPP row per chain, lane j holds column j of every operand
// (pgps_rc.hip.h).  Compiled into the fp64 units of pgps_rc_inst.hip.
//
// Reference semantics: the value differentiated is pkf's log-likelihood, pssgp/kalman/parallel.py:135-151 (equal to the
// sequential filter's, sequential.py:11-47); the reference differentiates it with tf.GradientTape
// (tests/test_gp_vs_kfs.py:53-78).  The checker is oracle/np_grad.py.
#pragma once

#include "pgps_gradlti.h"
#include "pgps_rc.hip.h"

namespace pgps {
namespace rc {

// ====================================================================================================
// forward: the Kalman pass of rc_apply1 (filter only, implicit process noise, filtered moments stored) with the adjoint
// element of every step folded into the chain's total under the smoothing operator:
//   element of step k:  E = A^T = F^T - v K^T,  g = v r / s,  L = -v v^T / (2 s)      v = (H F)^T, K = Pp H^T / s
//   total <- total (x) element:  E' = Ec E,  g' = Ec g + gc,  L' = Ec L Ec^T + Lc
// Steps at or beyond N run as F = I, y missing: the identity element, state unchanged.
// ====================================================================================================
template <int D>
__global__ __launch_bounds__(64) void rc_gapply1(const GradLtiArgs a) {
    using Real = double;
    __shared__ Real tl[4 * kPatch];
    const int lane = threadIdx.x & 15, row = threadIdx.x >> 4;
    Real* patch = patch_init(tl, row);
    constexpr int dd = D * D;
    const long kw = (long)blockIdx.x * 4 * a.Lw;
    const long c = (long)blockIdx.x * 4 + row;
    const long k0 = c * a.Lw;
    Io<D, Real> io;
    io.init(lane, row, a.Lw);
    const bool lv = io.lv, cv = c < a.nchunk;
    Real h[D];
#pragma unroll
    for (int i = 0; i < D; ++i) h[i] = a.H[i];
    const Real hl = lv ? a.H[lane] : Real(0.0);
    Real Pinf[D];
#pragma unroll
    for (int i = 0; i < D; ++i) Pinf[i] = lv ? Real(0.5) * (a.Pinf[i * D + lane] + a.Pinf[lane * D + i]) : Real(0.0);
    // state entering the chain: (b, C) of the inclusive prefix of the chain before (A = 0 there); the prior (0, Pinf) for chain 0
    Real m = Real(0.0), P[D];
    {
        const bool pr = cv && c > 0;
        const Real* rec = a.pre + (pr ? c - 1 : 0) * nfilt(D);
        m = (pr && lv) ? rec[3 * dd + lane] : Real(0.0);
#pragma unroll
        for (int i = 0; i < D; ++i) P[i] = !lv ? Real(0.0) : (pr ? rec[dd + i * D + lane] : Pinf[i]);
    }
    Real Ec[D], L[D], g = Real(0.0);
#pragma unroll
    for (int i = 0; i < D; ++i) { Ec[i] = (i == lane) ? Real(1.0) : Real(0.0); L[i] = Real(0.0); }
    LogLik ll;
    Real Fc[D], Fr[D], y;
    zero<D>(Fc); zero<D>(Fr);
    auto load = [&](int s) {
        const long ku = kw + s, k = k0 + s;
        const long kc = ku < a.N ? ku : a.N - 1;
        const bool real = k < a.N;
        io.template mat_slow<false>(a.Fs + kc * dd, real, Real(1.0), Fc);
        io.template mat_slow<true>(a.Fs + kc * dd, real, Real(1.0), Fr);
        y = __builtin_nan("");
        if (real) y = a.ys[k];
    };
    load(0);
    for (int s = 0; s < a.Lw; ++s) {
        const long ku = kw + s, k = k0 + s;
        // predict: Pp = Pinf + F (P - Pinf) F^T
        Real Pm[D], FP[D], Pp[D];
#pragma unroll
        for (int i = 0; i < D; ++i) Pm[i] = P[i] - Pinf[i];
        zero<D>(FP); mm<D>(FP, Fc, Pm);
        copy<D>(Pp, Pinf); mm<D>(Pp, FP, Fr);
        const Real mp = mvr<D>(Fr, m, Real(0.0));
        const Real v = dot_h<D>(Fc, h);                     // (H F)_lane
        const Real yk = y;
        // E' = Ec F^T - (Ec v) K^T needs F in row layout: before the next step's loads overwrite it
        Real E2[D];
        zero<D>(E2); mm<D>(E2, Ec, Fr);
        if (s + 1 < a.Lw) load(s + 1);
        symmetrise<D>(Pp, patch, lane);
        const bool obs = !(yk != yk);
        const Real u = dot_h<D>(Pp, h);
        const Real S = rowsum<D>(hl * u, a.R), mu = rowsum<D>(hl * mp, Real(0.0));
        if (obs) ll.add((double)yk - (double)mu, (double)S);
        const Real inv = obs ? Real(1.0) / S : Real(0.0);
        const Real res = obs ? yk - mu : Real(0.0);
        const Real K = u * inv;
        // fold the adjoint element
        Real Er[D];
        transpose<D>(Ec, Er, patch, lane);
        const Real w = mvr<D>(Er, v, Real(0.0));            // Ec v
        g = mvr<D>(Er, v * (res * inv), g);
        rank1<D>(L, w, Real(-0.5) * inv * w);
        rank1<D>(E2, w, -K);
        copy<D>(Ec, E2);
        // update
        m = mp + u * (inv * res);
        copy<D>(P, Pp); rank1<D>(P, u, -u * inv);
        const bool st = k < a.N;
        io.template st_mat<false>(a.fPs + ku * dd, st, P);
        io.template st_vec<false>(a.fms + ku * D, st, m);
    }
    if (cv) {
        if (lv) {
            Real* rec = a.sagg + c * nsmth(D);
#pragma unroll
            for (int i = 0; i < D; ++i) { rec[i * D + lane] = Ec[i]; rec[dd + i * D + lane] = L[i]; }
            rec[2 * dd + lane] = g;
        }
        if (lane == 0) a.llpart[c] = ll.value();
    }
}

// ====================================================================================================
// backward: from (a, W) behind the chain (the suffix of the next chain's total; zero behind the last one) down the
// chain's steps, recomputing every step's predict from the stored filtered moments of the step before.
//   B = W + a a^T / 2;   sbar, rbar, ubar, mpbar, Ppbar as in oracle/np_grad.py;
//   Abar += dt [mpbar mp^T + 2 Ppbar (Pp - Pinf)],  Ubar += ubar,  Hbar += sbar u + Pp ubar - rbar mp,  Rbar += sbar
//   a <- F^T mpbar,   W <- F^T Ppbar F - a a^T / 2
// Steps at or beyond N: F = I, y missing, dt = 0 -- nothing accumulates and (a, W) pass through.
// ====================================================================================================
template <int D>
__global__ __launch_bounds__(64) void rc_gback1(const GradLtiArgs a) {
    using Real = double;
    __shared__ Real tl[4 * kPatch];
    const int lane = threadIdx.x & 15, row = threadIdx.x >> 4;
    Real* patch = patch_init(tl, row);
    constexpr int dd = D * D;
    const long kw = (long)blockIdx.x * 4 * a.Lw;
    const long c = (long)blockIdx.x * 4 + row;
    const long k0 = c * a.Lw;
    Io<D, Real> io;
    io.init(lane, row, a.Lw);
    const bool lv = io.lv, cv = c < a.nchunk;
    Real h[D];
#pragma unroll
    for (int i = 0; i < D; ++i) h[i] = a.H[i];
    const Real hl = lv ? a.H[lane] : Real(0.0);
    Real Pinf[D];
#pragma unroll
    for (int i = 0; i < D; ++i) Pinf[i] = lv ? Real(0.5) * (a.Pinf[i * D + lane] + a.Pinf[lane * D + i]) : Real(0.0);
    Real av = Real(0.0), W[D];
    {
        const bool nx = c + 1 < a.nchunk;
        const Real* rec = a.suf + (nx ? c + 1 : 0) * nsmth(D);
        av = (nx && lv) ? rec[2 * dd + lane] : Real(0.0);
#pragma unroll
        for (int i = 0; i < D; ++i) W[i] = (nx && lv) ? rec[dd + i * D + lane] : Real(0.0);
    }
    Real Ab[D], Ub = Real(0.0), Hb = Real(0.0), Rb = Real(0.0);
    zero<D>(Ab);
    Real Fc[D], Fr[D], Pq[D], mq = Real(0.0), y = Real(0.0), dt = Real(0.0);
    zero<D>(Fc); zero<D>(Fr); zero<D>(Pq);
    auto load = [&](int s) {
        const long ku = kw + s, k = k0 + s;
        const long kc = ku < a.N ? ku : a.N - 1;
        const bool real = k < a.N, prev = real && k > 0;
        io.template mat_slow<false>(a.Fs + kc * dd, real, Real(1.0), Fc);
        io.template mat_slow<true>(a.Fs + kc * dd, real, Real(1.0), Fr);
        // filtered moments of the step before (the prior before the first step; anything finite beyond the end)
        const long kq = (ku >= 1 ? ku : 1) - 1, kqc = kq < a.N ? kq : a.N - 1;
        const Real* bP = a.fPs + (ku >= 1 ? kqc : -1) * dd;         // (row 0 of workgroup 0 at s = 0 is not `prev`: never read)
        const Real* bm = a.fms + (ku >= 1 ? kqc : -1) * D;
        io.template mat_slow<false>(bP, prev, Real(0.0), Pq);
        mq = io.vec(bm, prev);
        if (!prev) {
#pragma unroll
            for (int i = 0; i < D; ++i) Pq[i] = Pinf[i];
        }
        y = __builtin_nan("");
        dt = Real(0.0);
        if (real) {
            y = a.ys[k];
            dt = a.ts[k] - (k > 0 ? a.ts[k - 1] : a.t0);
        }
    };
    load(a.Lw - 1);
    for (int s = a.Lw - 1; s >= 0; --s) {
        // predict of this step from the stored moments of the step before
        Real Pm[D], FP[D], Pp[D];
#pragma unroll
        for (int i = 0; i < D; ++i) Pm[i] = Pq[i] - Pinf[i];
        zero<D>(FP); mm<D>(FP, Fc, Pm);
        copy<D>(Pp, Pinf); mm<D>(Pp, FP, Fr);
        const Real mp = mvr<D>(Fr, mq, Real(0.0));
        const Real yk = y, dtk = dt;
        symmetrise<D>(Pp, patch, lane);
        const bool obs = !(yk != yk);
        const Real u = dot_h<D>(Pp, h);
        const Real S = rowsum<D>(hl * u, a.R), mu = rowsum<D>(hl * mp, Real(0.0));
        const Real inv = obs ? Real(1.0) / S : Real(0.0);
        const Real res = obs ? yk - mu : Real(0.0);
        const Real K = u * inv;
        // adjoint of the update
        Real B[D];
        copy<D>(B, W); rank1<D>(B, av, Real(0.5) * av);
        const Real BK = mvr<D>(B, K, Real(0.0));            // (B symmetric: its columns are its rows)
        const Real kap = rowsum<D>(K * BK, Real(0.0)), aK = rowsum<D>(av * K, Real(0.0));
        const Real ri = res * inv;
        const Real sbar = obs ? (-aK * ri + kap - Real(0.5) * inv + Real(0.5) * ri * ri) : Real(0.0);
        const Real rbar = aK - ri;
        const Real ubar = av * ri - Real(2.0) * BK + sbar * hl;
        const Real mpbar = av - rbar * hl;
        Real Ppb[D];
        copy<D>(Ppb, B);
        rank1<D>(Ppb, ubar, Real(0.5) * hl);
        rank1<D>(Ppb, hl, Real(0.5) * ubar);
        // the model's adjoints
        Real X[D];
#pragma unroll
        for (int i = 0; i < D; ++i) X[i] = (Real(2.0) * dtk) * (Pp[i] - Pinf[i]);
        mm<D>(Ab, Ppb, X);
        rank1<D>(Ab, mpbar, dtk * mp);
        Ub += ubar;
        Hb += sbar * u + mvr<D>(Pp, ubar, Real(0.0)) - rbar * mp;
        Rb += sbar;
        // through the predict: a <- F^T mpbar, W <- F^T Ppbar F - a a^T / 2
        Real T[D], Bn[D];
        zero<D>(T); mm<D>(T, Ppb, Fc);
        zero<D>(Bn); mm<D>(Bn, Fr, T);
        av = mvr<D>(Fc, mpbar, Real(0.0));
        if (s > 0) load(s - 1);
        symmetrise<D>(Bn, patch, lane);
        copy<D>(W, Bn); rank1<D>(W, av, Real(-0.5) * av);
    }
    if (cv && lv) {
        Real* rec = a.gpart + c * (long)(dd + 2 * D + 1);
#pragma unroll
        for (int i = 0; i < D; ++i) rec[i * D + lane] = Ab[i];
        rec[dd + lane] = Ub;
        rec[dd + D + lane] = Hb;
        if (lane == 0) rec[dd + 2 * D] = Rb;
    }
}

template <int D>
int launch_rc_grad(pgps_ctx* ctx, const GradLtiArgs& a, int phase) {
    const dim3 blk(64), g1((unsigned)((a.nchunk + 3) / 4));
    switch (phase) {
        case 0: timed_launch(ctx, PGPS_K_FILTER_APPLY, rc_gapply1<D>, g1, blk, 0u, a); break;
        case 1: timed_launch(ctx, PGPS_K_SMOOTHER_APPLY, rc_gback1<D>, g1, blk, 0u, a); break;
        default:
            hipLaunchKernelGGL(k_grad_lti_finalize, dim3((unsigned)(1 + grad_lti_nstat(D))), dim3(256), 0, ctx->stream, (long)a.nchunk,
                               grad_lti_nstat(D), (const double*)a.llpart, (const double*)a.gpart, a.out);
            break;
    }
    HIPCHK(ctx, hipGetLastError());
    return PGPS_OK;
}

}  // namespace rc
}  // namespace pgps
