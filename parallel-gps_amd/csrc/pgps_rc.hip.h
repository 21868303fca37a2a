// pgps_rc.hip.h -- the "row-cooperative" scan family: fp64, state dimensions up to 16 (runtime d).
// Included by pgps_wc.hip (it shares that file's LDS combine routines for the upper scan levels).
//
// Between the lane-chunk family (one lane owns whole d x d operands, d <= 6) and the wave-cooperative one
// (64 lanes share LDS-resident operands, d <= 32) sits the case the c5 config (d = 11) and RBF order 15 live in:
// operands too large for one lane, yet small enough that LDS tiles starve the fp64 pipes (two LDS reads per
// 2 x 2 tile step).  Here ONE 16-LANE DPP ROW owns a chain of consecutive time steps and LANE j HOLDS COLUMN j
// of every operand in registers (DP doubles per matrix, DP in {8, 12, 16}); a wavefront runs four chains.
// A product Z = X Y is DP^2 instructions per lane, every one a v_fmac_f64_dpp whose first operand is lane k's
// register broadcast to the row (row_newbcast:k) -- no LDS, no shuffles, no extra moves:
//     Z_i(lane j) += bcast_k(X_i) * Y_k(lane j)          (pgps_rc_asm.h, generated)
// Products with a transposed right operand take that operand in row layout (lane j holds row j), which for the
// per-step inputs is simply a second load of the same 8 d^2 bytes; only three or four operands per step go
// through a 2 KB LDS patch to be transposed.  Rank-one updates, matrix-vector products and the Gauss-Jordan
// elimination of the smoother gain are the same broadcast-fmac pattern.  MFMA is not used.
//
// Scan structure (same algebra as pgps_math.h, reference pssgp/kalman/parallel.py:13-196):
//   rc_reduce1   chain = chunk of Lw steps: filt_extend per step           -> chunk totals (A, b, C, J, eta)
//   ks_filter    Kogge-Stone inclusive scan of the totals (wc::combine in LDS, ceil(log2 nchunk) launches)
//   rc_apply1    Kalman pass from the prefix's (b, C) [every prefix has A = 0]: fms, fPs, log-lik partials;
//                per step the smoothing element (E, g, L) is built ONCE, stored (E -> sPs, g -> sms, L ->
//                workspace) and folded into the chunk's smoothing total
//   ks_smoother  Kogge-Stone inclusive suffix scan of the smoothing totals (wc::scombine)
//   rc_smooth1   backward pass  sm = E sm' + g,  sP = E sP' E^T + L  from the stored elements (two products
//                per step instead of predict + gain solve + two products), overwriting sms / sPs in place
#pragma once

#include "pgps_rc_asm.h"

namespace pgps {
namespace rc {

constexpr int kLdT = 17;                    // leading dimension of a transpose patch (16 lanes + 1)
constexpr int kPatch = 16 * kLdT;           // doubles per row patch

template <int DP> struct Ops;
#define PGPS_RC_OPS(DPV)                                                                                              \
    template <> struct Ops<DPV> {                                                                                     \
        static __device__ __forceinline__ void rows4(double& a0, double& a1, double& a2, double& a3, double x0,       \
                                                     double x1, double x2, double x3, const double* y) {             \
            rows4_##DPV(a0, a1, a2, a3, x0, x1, x2, x3, y);                                                           \
        }                                                                                                             \
        static __device__ __forceinline__ void mv(double& a0, double& a1, double v, const double* x) { mv_##DPV(a0, a1, v, x); } \
        static __device__ __forceinline__ void rank1(double* z, double p, double q) { rank1_##DPV(z, p, q); }        \
    };
PGPS_RC_OPS(8)
PGPS_RC_OPS(12)
PGPS_RC_OPS(16)
#undef PGPS_RC_OPS

// z += X Y  (z pre-loaded with the addend; z must not alias x or y)
template <int DP>
__device__ __forceinline__ void mm(double* z, const double* x, const double* y) {
#pragma unroll
    for (int i = 0; i < DP; i += 4) Ops<DP>::rows4(z[i], z[i + 1], z[i + 2], z[i + 3], x[i], x[i + 1], x[i + 2], x[i + 3], y);
}
template <int DP>
__device__ __forceinline__ void zero(double* z) {
#pragma unroll
    for (int i = 0; i < DP; ++i) z[i] = 0.0;
}
template <int DP>
__device__ __forceinline__ void copy(double* z, const double* x) {
#pragma unroll
    for (int i = 0; i < DP; ++i) z[i] = x[i];
}
// (X v)_lane + add, X in ROW layout (lane i holds row i), v distributed (lane k holds v_k); also the
// row-wide sum  sum_k v_k x_k + add  when x is replicated (every lane gets the same value)
template <int DP>
__device__ __forceinline__ double mvr(const double* xr, double v, double add) {
    double a0 = add, a1 = 0.0;
    Ops<DP>::mv(a0, a1, v, xr);
    return a0 + a1;
}
// sum_i h_i X[i][lane]: (X^T h)_lane, = (X h)_lane for symmetric X
template <int DP>
__device__ __forceinline__ double dot_h(const double* x, const double* h) {
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int i = 0; i < DP; i += 2) { a0 = __builtin_fma(h[i], x[i], a0); a1 = __builtin_fma(h[i + 1], x[i + 1], a1); }
    return a0 + a1;
}
template <int DP>
__device__ __forceinline__ void rank1(double* z, double p, double q) { Ops<DP>::rank1(z, p, q); }

// xt = X^T through the row's LDS patch (lanes >= DP get zeros)
template <int DP>
__device__ __forceinline__ void transpose(const double* x, double* xt, double* patch, int lane) {
#pragma unroll
    for (int i = 0; i < DP; ++i) patch[i * kLdT + lane] = x[i];
    wc::sync();
#pragma unroll
    for (int i = 0; i < DP; ++i) { const double v = patch[lane * kLdT + i]; xt[i] = lane < DP ? v : 0.0; }
    wc::sync();
}
template <int DP>
__device__ __forceinline__ void symmetrise(double* x, double* patch, int lane) {
    double xt[DP];
    transpose<DP>(x, xt, patch, lane);
#pragma unroll
    for (int i = 0; i < DP; ++i) x[i] = 0.5 * (x[i] + xt[i]);
}

template <int K>
__device__ __forceinline__ double bcast(double x) { return __builtin_amdgcn_update_dpp(x, x, 0x150 + K, 0xf, 0xf, true); }

// Gauss-Jordan without pivoting (M symmetric positive definite, identity on the padding): B <- M^-1 B.
// Row operations in column layout: row_r -= M[r][c] * row_c / M[c][c], the factor M[r][c] being lane c's
// register r broadcast to the row.  M is destroyed.
template <int DP, int C>
struct GjStep {
    static __device__ __forceinline__ void run(double* M, double* B) {
        const double inv = 1.0 / bcast<C>(M[C]);
        const double mc = M[C] * inv, bv = B[C] * inv;
        gj8<C>(M, B, -mc, -bv);
        if constexpr (DP == 12) gj4<C>(M + 8, B + 8, -mc, -bv);
        if constexpr (DP == 16) gj8<C>(M + 8, B + 8, -mc, -bv);
        M[C] = mc;                       // the pivot row itself (whatever the block did to it is discarded)
        B[C] = bv;
        if constexpr (C + 1 < DP) GjStep<DP, C + 1>::run(M, B);
    }
};

struct RcArgs {
    long N;
    int d, Lw;
    long nchunk;
    const double *P0, *H;
    double R;
    const double *Fs, *Qs, *ys;
    double *fms, *fPs, *sms, *sPs;
    double* agg1;               // (nchunk, nfilt) chunk totals
    const double* pre;          // (nchunk, nfilt) inclusive prefixes of agg1
    double* sagg1;              // (nchunk, nsmth) smoothing totals
    const double* suf;          // (nchunk, nsmth) inclusive suffixes of sagg1
    double* Lws;                // (N, d, d) the smoothing elements' L
    double* llpart;             // (nchunk,)
};

// one step's inputs in both layouts; inactive steps get `fdiag` on the diagonal of F and `qdiag` on Q's
template <int DP>
__device__ __forceinline__ void load_step(const RcArgs& a, long k, bool real, double fdiag, double qdiag, int lane,
                                          double* Fc, double* Fr, double* Q) {
    const int d = a.d;
    const bool lv = lane < d;
    const double* Fg = a.Fs + k * (long)d * d;
    const double* Qg = a.Qs + k * (long)d * d;
#pragma unroll
    for (int i = 0; i < DP; ++i) {
        const bool in = real && lv && i < d;
        const double diag = (i == lane && lv) ? 1.0 : 0.0;
        Fc[i] = in ? Fg[i * d + lane] : fdiag * diag;
        Fr[i] = in ? Fg[lane * d + i] : fdiag * diag;
        Q[i] = in ? 0.5 * (Qg[i * d + lane] + Qg[lane * d + i]) : qdiag * diag;
    }
}

// ====================================================================================================
// level 1: reduce -- filt_extend over the chunk (pgps_math.h filt_extend, parallel.py:46-72,100-118)
// ====================================================================================================
template <int DP>
__global__ __launch_bounds__(64) void rc_reduce1(const RcArgs a) {
    __shared__ double tl[4 * kPatch];
    const int lane = threadIdx.x & 15, row = threadIdx.x >> 4;
    double* patch = tl + row * kPatch;
    const int d = a.d, dd = d * d;
    const long c = (long)blockIdx.x * 4 + row;
    const long k0 = c * a.Lw, k1 = min(a.N, k0 + a.Lw);
    const bool lv = lane < d;
    double h[DP];
#pragma unroll
    for (int i = 0; i < DP; ++i) h[i] = i < d ? a.H[i] : 0.0;
    double A[DP], C[DP], J[DP], b = 0.0, eta = 0.0;
#pragma unroll
    for (int i = 0; i < DP; ++i) {
        // chunk 0 starts from (0, 0, P0, 0, 0) and takes step 0 with F = I, Q = 0: that is filt_first
        A[i] = (c != 0 && i == lane && lv) ? 1.0 : 0.0;
        C[i] = (c == 0 && lv && i < d) ? 0.5 * (a.P0[i * d + lane] + a.P0[lane * d + i]) : 0.0;
        J[i] = 0.0;
    }
    double Fc[DP], Fr[DP], Q[DP];
    load_step<DP>(a, k0, k0 < k1 && k0 != 0, 1.0, 0.0, lane, Fc, Fr, Q);
    double y = k0 < k1 ? a.ys[k0] : __builtin_nan("");
    for (int s = 0; s < a.Lw; ++s) {
        double Ap[DP], FC[DP], Cp[DP];
        zero<DP>(Ap); mm<DP>(Ap, Fc, A);
        zero<DP>(FC); mm<DP>(FC, Fc, C);
        copy<DP>(Cp, Q); mm<DP>(Cp, FC, Fr);
        const double bp = mvr<DP>(Fr, b, 0.0);
        const double yk = y;
        {   // next step's inputs: their registers are free from here on
            const long kn = k0 + s + 1;
            const bool real = (s + 1 < a.Lw) && kn < k1;
            load_step<DP>(a, real ? kn : 0, real, 1.0, 0.0, lane, Fc, Fr, Q);
            y = real ? a.ys[kn] : __builtin_nan("");
        }
        symmetrise<DP>(Cp, patch, lane);
        const double u = dot_h<DP>(Cp, h), v = dot_h<DP>(Ap, h);
        const double S = mvr<DP>(h, u, a.R), hb = mvr<DP>(h, bp, 0.0);
        const bool obs = !(yk != yk);
        const double inv = obs ? 1.0 / S : 0.0;
        const double res = obs ? yk - hb : 0.0;
        copy<DP>(A, Ap); rank1<DP>(A, u, -v * inv);
        copy<DP>(C, Cp); rank1<DP>(C, u, -u * inv);
        rank1<DP>(J, v, v * inv);
        b = bp + u * (inv * res);
        eta += v * (res * inv);
    }
    if (c < a.nchunk) {
        double* rec = a.agg1 + c * wc::nfilt(d);
#pragma unroll
        for (int i = 0; i < DP; ++i)
            if (lv && i < d) { rec[i * d + lane] = A[i]; rec[dd + i * d + lane] = C[i]; rec[2 * dd + i * d + lane] = J[i]; }
        if (lv) { rec[3 * dd + lane] = b; rec[3 * dd + d + lane] = eta; }
    }
}

// ====================================================================================================
// level 1: apply -- Kalman pass, log-likelihood, smoothing elements and the chunk's smoothing total
// (kf_step / smth_element / smth_combine of pgps_math.h; parallel.py:135-151, 155-184)
// ====================================================================================================
template <int DP, bool SMOOTH>
__global__ __launch_bounds__(64) void rc_apply1(const RcArgs a) {
    __shared__ double tl[4 * kPatch];
    const int lane = threadIdx.x & 15, row = threadIdx.x >> 4;
    double* patch = tl + row * kPatch;
    const int d = a.d, dd = d * d;
    const long c = (long)blockIdx.x * 4 + row;
    const long k0 = c * a.Lw, k1 = min(a.N, k0 + a.Lw);
    const bool lv = lane < d, cv = c < a.nchunk;
    double h[DP];
#pragma unroll
    for (int i = 0; i < DP; ++i) h[i] = i < d ? a.H[i] : 0.0;
    // state entering the chunk: (b, C) of the inclusive prefix of the chunk before (A = 0 there); prior for chunk 0
    double m, P[DP];
    {
        const double* rec = a.pre + (cv && c > 0 ? c - 1 : 0) * wc::nfilt(d);
        const bool pr = cv && c > 0;
        m = (pr && lv) ? rec[3 * dd + lane] : 0.0;
#pragma unroll
        for (int i = 0; i < DP; ++i) {
            const bool in = lv && i < d;
            P[i] = !in ? 0.0 : pr ? rec[dd + i * d + lane] : (c == 0 ? 0.5 * (a.P0[i * d + lane] + a.P0[lane * d + i]) : 0.0);
        }
    }
    double Ec[DP], Er[DP], L[DP], g = 0.0;     // smoothing total of the steps seen so far (E in both layouts)
    if (SMOOTH) {
#pragma unroll
        for (int i = 0; i < DP; ++i) { Ec[i] = (i == lane && lv) ? 1.0 : 0.0; Er[i] = Ec[i]; L[i] = 0.0; }
    }
    LogLik ll;
    double Fc[DP], Fr[DP], Q[DP];
    // steps at or beyond N run with F = 0, Q = I: the element built from them is (0, m, P), i.e. the last
    // element of the series (parallel.py:155-156), and a total whose E is 0 absorbs whatever follows unchanged
    load_step<DP>(a, k0 < a.N ? k0 : 0, k0 < a.N, 0.0, 1.0, lane, Fc, Fr, Q);
    const int iters = SMOOTH ? a.Lw + 1 : a.Lw;
    for (int s = 0; s < iters; ++s) {
        const long k = k0 + s;
        // predict
        double FP[DP], Pp[DP];
        zero<DP>(FP); mm<DP>(FP, Fc, P);
        copy<DP>(Pp, Q); mm<DP>(Pp, FP, Fr);
        const double mp = mvr<DP>(Fr, m, 0.0);
        {
            const long kn = k + 1;
            const bool real = (s + 1 < iters) && kn < a.N;
            load_step<DP>(a, real ? kn : 0, real, 0.0, 1.0, lane, Fc, Fr, Q);
        }
        symmetrise<DP>(Pp, patch, lane);
        if (SMOOTH && s > 0) {
            // element of step k-1: W = Pp^-1 F P = E^T (i.e. E in row layout), g = m - E mp, L = P - E F P
            double M[DP], W[DP];
#pragma unroll
            for (int i = 0; i < DP; ++i) { M[i] = (i == lane && !lv) ? 1.0 : Pp[i]; W[i] = FP[i]; }
            GjStep<DP, 0>::run(M, W);
            const double gn = m - mvr<DP>(W, mp, 0.0);
            double En[DP], Ln[DP], T[DP];
            transpose<DP>(W, En, patch, lane);
            zero<DP>(T); mm<DP>(T, En, FP);
#pragma unroll
            for (int i = 0; i < DP; ++i) Ln[i] = P[i] - T[i];
            if (k - 1 < k1) {
#pragma unroll
                for (int i = 0; i < DP; ++i)
                    if (lv && i < d) { a.sPs[(k - 1) * dd + i * d + lane] = En[i]; a.Lws[(k - 1) * dd + i * d + lane] = Ln[i]; }
                if (lv) a.sms[(k - 1) * d + lane] = gn;
            }
            // total <- total (x) element:  E = Ea En, g = Ea gn + ga, L = Ea Ln Ea^T + La
            double E2[DP];
            zero<DP>(E2); mm<DP>(E2, Ec, En);
            zero<DP>(T); mm<DP>(T, Ec, Ln);
            g = mvr<DP>(Er, gn, g);
            mm<DP>(L, T, Er);
            symmetrise<DP>(L, patch, lane);
            copy<DP>(Ec, E2);
            transpose<DP>(E2, Er, patch, lane);
        }
        if (s < a.Lw) {
            const bool upd = k < k1;
            const double y = upd ? a.ys[k] : __builtin_nan("");
            const bool obs = !(y != y);
            double u = dot_h<DP>(Pp, h);
            double S = mvr<DP>(h, u, a.R), mu = mvr<DP>(h, mp, 0.0);
            if (obs) ll.add(y - mu, S);
            double mb = mp;
            if (blockIdx.x == 0 && s == 0) {
                // first step of the series (chain 0 only): the update uses the prior itself (parallel.py:24-30),
                // the likelihood term above used F0 P0 F0^T + Q0 (parallel.py:136-141)
                const bool first = (c == 0);
                const double u0 = dot_h<DP>(P, h);
                const double S0 = mvr<DP>(h, u0, a.R), mu0 = mvr<DP>(h, m, 0.0);
#pragma unroll
                for (int i = 0; i < DP; ++i) Pp[i] = first ? P[i] : Pp[i];
                mb = first ? m : mp;
                u = first ? u0 : u;
                S = first ? S0 : S;
                mu = first ? mu0 : mu;
            }
            const double inv = obs ? 1.0 / S : 0.0;
            const double res = obs ? y - mu : 0.0;
            m = mb + u * (inv * res);
            copy<DP>(P, Pp); rank1<DP>(P, u, -u * inv);
            if (upd) {
#pragma unroll
                for (int i = 0; i < DP; ++i)
                    if (lv && i < d) a.fPs[k * dd + i * d + lane] = P[i];
                if (lv) a.fms[k * d + lane] = m;
            }
        }
    }
    if (cv) {
        if (SMOOTH) {
            double* rec = a.sagg1 + c * wc::nsmth(d);
#pragma unroll
            for (int i = 0; i < DP; ++i)
                if (lv && i < d) { rec[i * d + lane] = Ec[i]; rec[dd + i * d + lane] = L[i]; }
            if (lv) rec[2 * dd + lane] = g;
        }
        if (lane == 0) a.llpart[c] = ll.value();
    }
}

// ====================================================================================================
// level 1: smoother -- sm = E sm' + g, sP = E sP' E^T + L from the stored elements (parallel.py:176-184)
// ====================================================================================================
template <int DP>
__device__ __forceinline__ void load_elem(const RcArgs& a, long k, bool real, int lane, double* Ec, double* Er, double* L,
                                          double& g) {
    const int d = a.d;
    const bool lv = lane < d;
    const double* Eg = a.sPs + k * (long)d * d;
    const double* Lg = a.Lws + k * (long)d * d;
#pragma unroll
    for (int i = 0; i < DP; ++i) {
        const bool in = real && lv && i < d;
        const double diag = (i == lane && lv) ? 1.0 : 0.0;
        Ec[i] = in ? Eg[i * d + lane] : diag;       // inactive steps: the identity element
        Er[i] = in ? Eg[lane * d + i] : diag;
        L[i] = in ? Lg[i * d + lane] : 0.0;
    }
    g = (real && lv) ? a.sms[k * d + lane] : 0.0;
}

template <int DP>
__global__ __launch_bounds__(64) void rc_smooth1(const RcArgs a) {
    __shared__ double tl[4 * kPatch];
    const int lane = threadIdx.x & 15, row = threadIdx.x >> 4;
    double* patch = tl + row * kPatch;
    const int d = a.d, dd = d * d;
    const long c = (long)blockIdx.x * 4 + row;
    const long k0 = c * a.Lw, k1 = min(a.N, k0 + a.Lw);
    const bool lv = lane < d;
    // smoothed moments of the first step after the chunk: (g, L) of the inclusive suffix of the next chunk
    // (E = 0 there: every suffix contains the last element); nothing (0, 0) after the last chunk, whose own
    // last element has E = 0
    double sm, sP[DP];
    {
        const bool nx = c + 1 < a.nchunk;
        const double* rec = a.suf + (nx ? c + 1 : 0) * wc::nsmth(d);
        sm = (nx && lv) ? rec[2 * dd + lane] : 0.0;
#pragma unroll
        for (int i = 0; i < DP; ++i) sP[i] = (nx && lv && i < d) ? rec[dd + i * d + lane] : 0.0;
    }
    double Ec[DP], Er[DP], L[DP], g;
    {
        const long k = k0 + a.Lw - 1;
        load_elem<DP>(a, k < k1 ? k : 0, k < k1, lane, Ec, Er, L, g);
    }
    for (int s = a.Lw - 1; s >= 0; --s) {
        const long k = k0 + s;
        double T[DP], nP[DP];
        zero<DP>(T); mm<DP>(T, Ec, sP);
        copy<DP>(nP, L); mm<DP>(nP, T, Er);
        sm = mvr<DP>(Er, sm, g);
        {
            const long kn = k - 1;
            const bool real = s > 0 && kn < k1;
            load_elem<DP>(a, real ? kn : 0, real, lane, Ec, Er, L, g);
        }
        symmetrise<DP>(nP, patch, lane);
        copy<DP>(sP, nP);
        if (k < k1) {
#pragma unroll
            for (int i = 0; i < DP; ++i)
                if (lv && i < d) a.sPs[k * dd + i * d + lane] = sP[i];
            if (lv) a.sms[k * d + lane] = sm;
        }
    }
}

// ====================================================================================================
// upper levels: Kogge-Stone scans over the chunk totals, one wave per record, the combine in LDS (pgps_wc.hip)
// ====================================================================================================
template <int DP>
__global__ __launch_bounds__(64) void ks_filter(int d, long n, long stride, const double* in, double* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using GE = wc::Geo<DP>;
    const long c = blockIdx.x;
    const int nf = wc::nfilt(d), dk = (d + 3) & ~3;
    if (c < stride) {
        for (int e = threadIdx.x; e < nf; e += 64) out[c * nf + e] = in[c * nf + e];
        return;
    }
    wc::Pool<double> pool(reinterpret_cast<double*>(smem));
    double* e1 = pool.take(GE::NFL); double* e2 = pool.take(GE::NFL); double* o = pool.take(GE::NFL);
    double* M = pool.take(GE::MSZ); double* rhs = pool.take(DP * GE::NRC); double* X = pool.take(GE::MSZ);
    double* vt = pool.take(DP);
    wc::filt_g2l<double, DP>(d, in + (c - stride) * nf, e1);
    wc::filt_g2l<double, DP>(d, in + c * nf, e2);
    wc::sync();
    wc::combine<double, DP>(d, dk, e1, e2, o, M, rhs, X, vt);
    wc::filt_l2g<double, DP>(d, o, out + c * nf);
}

template <int DP>
__global__ __launch_bounds__(64) void ks_smoother(int d, long n, long stride, const double* in, double* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using GE = wc::Geo<DP>;
    const long c = blockIdx.x;
    const int ns = wc::nsmth(d), dk = (d + 3) & ~3;
    if (c + stride >= n) {
        for (int e = threadIdx.x; e < ns; e += 64) out[c * ns + e] = in[c * ns + e];
        return;
    }
    wc::Pool<double> pool(reinterpret_cast<double*>(smem));
    double* ea = pool.take(GE::NSL); double* eb = pool.take(GE::NSL); double* o = pool.take(GE::NSL);
    double* X = pool.take(GE::MSZ);
    wc::smth_g2l<double, DP>(d, in + c * ns, ea);
    wc::smth_g2l<double, DP>(d, in + (c + stride) * ns, eb);
    wc::sync();
    wc::scombine<double, DP>(dk, ea, eb, o, X);
    wc::smth_l2g<double, DP>(d, o, out + c * ns);
}

}  // namespace rc

// ---- host side ----------------------------------------------------------------------------------------
static inline size_t rc_align(size_t x) { return (x + 255) / 256 * 256; }

template <int DP>
static int launch_scan_rc_dp(pgps_ctx* ctx, rc::RcArgs a, Mode mode, double* aggA, double* aggB, double* saggA,
                             double* saggB, double* ll) {
    using namespace rc;
    using GE = wc::Geo<DP>;
    const int d = a.d;
    const size_t l_ksf = (3 * GE::NFL + 2 * GE::MSZ + (size_t)DP * GE::NRC + DP + 64) * sizeof(double);
    const size_t l_kss = (3 * GE::NSL + GE::MSZ + 64) * sizeof(double);
    HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(ks_filter<DP>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)l_ksf));
    HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(ks_smoother<DP>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)l_kss));
    const dim3 blk(64), g1((unsigned)((a.nchunk + 3) / 4)), gk((unsigned)a.nchunk);
    a.agg1 = aggA;
    timed_launch(ctx, PGPS_K_FILTER_REDUCE, rc_reduce1<DP>, g1, blk, 0u, a);
    double *src = aggA, *dst = aggB;
    for (long s = 1; s < a.nchunk; s *= 2) {
        timed_launch(ctx, PGPS_K_FILTER_REDUCE, ks_filter<DP>, gk, blk, (unsigned)l_ksf, d, (long)a.nchunk, s,
                     (const double*)src, dst);
        double* t = src; src = dst; dst = t;
    }
    a.pre = src;
    if (mode == MODE_PKFS) {
        a.sagg1 = saggA;
        timed_launch(ctx, PGPS_K_FILTER_APPLY, rc_apply1<DP, true>, g1, blk, 0u, a);
        src = saggA; dst = saggB;
        for (long s = 1; s < a.nchunk; s *= 2) {
            timed_launch(ctx, PGPS_K_SMOOTHER_REDUCE, ks_smoother<DP>, gk, blk, (unsigned)l_kss, d, (long)a.nchunk, s,
                         (const double*)src, dst);
            double* t = src; src = dst; dst = t;
        }
        a.suf = src;
        timed_launch(ctx, PGPS_K_SMOOTHER_APPLY, rc_smooth1<DP>, g1, blk, 0u, a);
    } else {
        timed_launch(ctx, PGPS_K_FILTER_APPLY, rc_apply1<DP, false>, g1, blk, 0u, a);
    }
    if (ll)
        timed_launch(ctx, PGPS_K_LL_FINALIZE, wc::wc_ll_finalize, dim3(1), blk, 0u, (const double*)a.llpart, (long)a.nchunk, ll);
    HIPCHK(ctx, hipGetLastError());
    return PGPS_OK;
}

// fp64, d <= 16, pkf / pkfs on one device
int launch_scan_rc(pgps_ctx* ctx, ScanArgs<double> sa, int d, Mode mode) {
    if (mode != MODE_PKF && mode != MODE_PKFS) return PGPS_E_UNSUPPORTED_DIM;
    if (d < 1 || d > 16) return PGPS_E_UNSUPPORTED_DIM;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    rc::RcArgs a{};
    a.N = sa.N; a.d = d;
    if (ctx->chunk > 0) {
        a.Lw = ctx->chunk;
    } else {
        // four chains per wave, one wave per SIMD: 4096 chains fill the chip
        long lw = (sa.N + 4095) / 4096;
        a.Lw = (int)(lw < 8 ? 8 : lw > 512 ? 512 : lw);
    }
    a.nchunk = (sa.N + a.Lw - 1) / a.Lw;
    a.P0 = sa.P0; a.H = sa.H; a.R = sa.R; a.Fs = sa.Fs; a.Qs = sa.Qs; a.ys = sa.ys;
    a.fms = sa.fms; a.fPs = sa.fPs; a.sms = sa.sms; a.sPs = sa.sPs;
    const size_t dd = (size_t)d * d, nf = wc::nfilt(d), ns = wc::nsmth(d), nc = (size_t)a.nchunk;
    size_t off = 0;
    const size_t o_aggA = off;  off = rc_align(off + nc * nf * sizeof(double));
    const size_t o_aggB = off;  off = rc_align(off + nc * nf * sizeof(double));
    const size_t o_sagA = off;  off = rc_align(off + nc * ns * sizeof(double));
    const size_t o_sagB = off;  off = rc_align(off + nc * ns * sizeof(double));
    const size_t o_ll = off;    off = rc_align(off + nc * sizeof(double));
    const size_t o_L = off;     if (mode == MODE_PKFS) off = rc_align(off + (size_t)sa.N * dd * sizeof(double));
    int rcode = ensure(ctx, ctx->ws, off);
    if (rcode) return rcode;
    char* base = (char*)ctx->ws.p;
    a.llpart = (double*)(base + o_ll);
    a.Lws = (double*)(base + o_L);
    double* aggA = (double*)(base + o_aggA); double* aggB = (double*)(base + o_aggB);
    double* sagA = (double*)(base + o_sagA); double* sagB = (double*)(base + o_sagB);
    if (d <= 8) return launch_scan_rc_dp<8>(ctx, a, mode, aggA, aggB, sagA, sagB, sa.ll);
    if (d <= 12) return launch_scan_rc_dp<12>(ctx, a, mode, aggA, aggB, sagA, sagB, sa.ll);
    return launch_scan_rc_dp<16>(ctx, a, mode, aggA, aggB, sagA, sagB, sa.ll);
}

}  // namespace pgps
