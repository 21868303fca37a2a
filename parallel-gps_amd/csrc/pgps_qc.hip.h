// pgps_qc.hip.h -- the "quad-cooperative" level-1 kernels: fp32, state dimensions 5..8 (config c3: RBF order 6).
//
// Between the lane-chunk family (one lane owns whole d x d operands: at d = 6 its fp32 kernels hold 256 VGPRs + 256
// AGPRs, one wave per SIMD, and spend two thirds of the reduce pass in scan trees over 65 536 lanes) and the row-
// cooperative one (a 16-lane DPP row per chain: 10 of 16 lanes idle at d = 6) sits this layout: ONE QUAD (4 lanes) owns
// a chain of consecutive time steps, LANE q HOLDS COLUMNS 2q, 2q+1 of every operand (2 d registers per matrix), a
// wavefront runs sixteen chains.  A product Z = X Y is 2 d^2 instructions per lane, each a v_fmac_f32 whose first
// operand is a quad_perm broadcast of the register that holds X[i][k] in lane k / 2 (DPP: no LDS, no shuffle):
//     Z[i][2q+j] += bcast_{k/2}( X[i][k] ) * Y[k][2q+j]
// A transposed LEFT operand costs nothing (X^T[i][k] = X[k][i] sits in lane i / 2: another broadcast); a transposed
// RIGHT operand is needed in row layout, which for the per-step inputs is a second load and for symmetric matrices
// (covariances) is the column layout itself.  The algebra is arranged so that nothing else is ever transposed:
//   * the smoothing gain comes out of the elimination as W = Pp^-1 (F P) = E^T in column layout = E in ROW layout,
//     which is what both its products want (E X as a transposed-left product, X E^T as a row-layout right operand);
//   * the chain's smoothing total is carried TRANSPOSED (Tt = E_tot^T): Tt' = W Tt, L' = L_tot + Tt^T (L_n Tt).
// Vectors (means, b, eta, g) are replicated in the four lanes; a matrix-vector product is computed where its rows or
// columns live and spread with d broadcasts.  Lanes whose columns do not exist (lane 3 at d <= 6; the second column
// of lane (d-1)/2 at odd d) compute on a duplicate of column 0: nothing ever broadcasts from them and their stores are
// suppressed, so they need no zeroing.  No LDS at all; d = 6: 12 registers per matrix.
//
// The kernels speak the row-cooperative family's protocol (pgps_rc.hip.h: chain totals [A | C | J | b | eta] and
// [E | L | g] as records in scratch, prefixes / suffixes of them by the upper-level scans, RcArgsT), so the host driver,
// the scans and the segment protocol of that family carry them (pgps_wc.hip scan_rc): they replace rc_reduce1 /
// rc_apply1 / rc_smooth1 for fp32 series at 5 <= d <= 8 in the filter and filter + smoother calls.
// Reference semantics: pssgp/kalman/parallel.py:13-196 (elements, operators, pkf, pks), as pgps_math.h.
#pragma once

#include <hip/hip_runtime.h>

#include "pgps_internal.h"
#include "pgps_math.h"

namespace pgps {
namespace qc {

constexpr int kChains = 16;                 // chains (quads) per wavefront
#ifndef PGPS_QC_WAVES
#define PGPS_QC_WAVES 2                      // waves per SIMD the kernels are compiled for (register budget 512 / waves)
#endif

// A lane's two columns of a matrix: m[i] = (M[i][2q], M[i][2q+1]) -- the pair sits in adjacent registers, so that one
// v_pk_fma_f32 updates both columns from one broadcast operand.
typedef float V2 __attribute__((ext_vector_type(2)));

// lane p of the quad's value in every lane of the quad
template <int P>
__device__ __forceinline__ float qb(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), P * 0x55, 0xf, 0xf, true));
}
__device__ __forceinline__ float qbi(float x, int p) {      // p is a compile-time constant after unrolling
    switch (p) {
        case 0: return qb<0>(x);
        case 1: return qb<1>(x);
        case 2: return qb<2>(x);
        default: return qb<3>(x);
    }
}
__device__ __forceinline__ V2 fma2(float a, V2 b, V2 c) { return __builtin_elementwise_fma(V2{a, a}, b, c); }
__device__ __forceinline__ float comp(V2 v, int j) { return j ? v.y : v.x; }

// z[i] += sum_k X(i, k) * yk[k],  X(i, k) = element (i, k) of the left operand, given in column layout:
//   LT = false: the operand is x itself        (X(i, k) = component k & 1 of x[i] in lane k / 2)
//   LT = true : the operand is x^T             (X(i, k) = component i & 1 of x[k] in lane i / 2)
// yk[k] = the right operand's elements (k, own columns): its column layout, or -- for a product with the right operand
// transposed -- that operand's ROW layout.  d^2 quad broadcasts + d^2 v_pk_fma_f32 per lane.
template <int D, bool LT>
__device__ __forceinline__ void mm(V2 (&z)[D], const V2 (&x)[D], const V2 (&yk)[D]) {
    auto X = [&](int i, int k) { return LT ? qbi(comp(x[k], i & 1), i >> 1) : qbi(comp(x[i], k & 1), k >> 1); };
#pragma unroll
    for (int k = 0; k < D; ++k) {
        // two broadcasts share a register pair (v_pk_fma_f32 takes its scalar operand as half of a 64-bit pair: one
        // broadcast per pair would waste the other half -- 2 d^2 registers of a product's operands in flight)
#pragma unroll
        for (int i = 0; i + 1 < D; i += 2) {
            const V2 xp = V2{X(i, k), X(i + 1, k)};
            z[i] = __builtin_elementwise_fma(V2{xp.x, xp.x}, yk[k], z[i]);
            z[i + 1] = __builtin_elementwise_fma(V2{xp.y, xp.y}, yk[k], z[i + 1]);
        }
        if constexpr (D & 1) z[D - 1] = fma2(X(D - 1, k), yk[k], z[D - 1]);
        // vector-ALU instructions stay on their side of this line (memory and scalar ones may cross): left alone, the
        // scheduler hoists the d^2 broadcasts of a product far ahead of their use and the kernels need twice the registers
        __builtin_amdgcn_sched_barrier(0x0014 | 0x0380 | 0x0060);
    }
}
template <int D>
__device__ __forceinline__ void zero2(V2 (&z)[D]) {
#pragma unroll
    for (int i = 0; i < D; ++i) z[i] = V2{0.0f, 0.0f};
}
template <int D>
__device__ __forceinline__ void copy2(V2 (&z)[D], const V2 (&x)[D]) {
#pragma unroll
    for (int i = 0; i < D; ++i) z[i] = x[i];
}
// a vector that lives distributed (lane i / 2 holds element i in component i & 1 of loc) -> replicated in every lane
template <int D>
__device__ __forceinline__ void spread(V2 loc, float (&v)[D]) {
#pragma unroll
    for (int i = 0; i < D; ++i) v[i] = qbi(comp(loc, i & 1), i >> 1);
}
// own columns of a matrix (column layout) dotted with a replicated vector: (M^T w)[own columns]
template <int D>
__device__ __forceinline__ V2 coldot(const V2 (&m)[D], const float (&w)[D]) {
    V2 a = V2{0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < D; ++i) a = fma2(w[i], m[i], a);
    return a;
}
// element idx (lane-dependent) of a replicated vector, without dynamic register indexing
template <int D>
__device__ __forceinline__ float pick(const float (&v)[D], int idx) {
    float r = v[0];
#pragma unroll
    for (int i = 1; i < D; ++i) r = (idx == i) ? v[i] : r;
    return r;
}
template <int D>
__device__ __forceinline__ float dot(const float (&a)[D], const float (&b)[D], float add) {
    float s = add;
#pragma unroll
    for (int i = 0; i < D; ++i) s = __builtin_fmaf(a[i], b[i], s);
    return s;
}
// z[i] += p[i] * qv  (rank one: p replicated, qv = the other vector's own-column elements)
template <int D>
__device__ __forceinline__ void rank1(V2 (&z)[D], const float (&p)[D], V2 qv) {
#pragma unroll
    for (int i = 0; i < D; ++i) z[i] = fma2(p[i], qv, z[i]);
}

// B <- M^-1 B for symmetric positive definite M (Gauss-Jordan, no pivoting), both in column layout; M is destroyed.
// Row operations: row_r -= M[r][c] * row_c / M[c][c]; the factors are column c, i.e. broadcasts from lane c / 2.
template <int D>
__device__ __forceinline__ void spd_solve(V2 (&m)[D], V2 (&b)[D]) {
#pragma unroll
    for (int c = 0; c < D; ++c) {
        const float inv = recip(qbi(comp(m[c], c & 1), c >> 1));
        const V2 mc = m[c] * inv, bc = b[c] * inv;
#pragma unroll
        for (int r = 0; r < D; ++r) {
            if (r == c) continue;
            const float f = -qbi(comp(m[r], c & 1), c >> 1);
            m[r] = fma2(f, mc, m[r]);
            b[r] = fma2(f, bc, b[r]);
        }
        m[c] = mc; b[c] = bc;
    }
}

// ---- addressing: lane q of a quad works on columns / rows cj = 2q + j (j = 0, 1); where such a column does not exist
// it works on a duplicate of column 0 (always inside the record) and never stores ------------------------------------
template <int D>
struct Lane {
    int cj[2];
    bool ok[2];
    bool lead;                              // lane 0 of the quad: writes the replicated vectors
    __device__ __forceinline__ void init(int tid) {
        const int q = tid & 3;
        ok[0] = 2 * q < D; ok[1] = 2 * q + 1 < D;
        // (a lane without columns duplicates columns 0 and 1: its paired accesses stay inside the record; the second
        // column of the last lane at odd d duplicates column 0)
        cj[0] = ok[0] ? 2 * q : 0; cj[1] = ok[1] ? 2 * q + 1 : (ok[0] ? 0 : (D > 1 ? 1 : 0));
        lead = q == 0;
    }
    // r[k] = (M[c0][k], M[c1][k]): rows of a row-major record.  Both rows exist: 2 d consecutive floats (whole
    // 16-byte pieces at even d), re-paired in registers; otherwise element by element.
    __device__ __forceinline__ void rows(const float* rec, V2 (&r)[D]) const {
        if constexpr (D % 2 == 0) {
            // (at even d every lane has both rows or -- the duplicate of rows 0, 1 -- none)
            using F4 = __attribute__((ext_vector_type(4))) float;
            const float* p = rec + cj[0] * D;
            float f[2 * D];
            if constexpr ((2 * D) % 4 == 0) {
#pragma unroll
                for (int v = 0; v < 2 * D / 4; ++v) {
                    const F4 x = *reinterpret_cast<const F4*>(p + 4 * v);
                    f[4 * v] = x.x; f[4 * v + 1] = x.y; f[4 * v + 2] = x.z; f[4 * v + 3] = x.w;
                }
            } else {
#pragma unroll
                for (int v = 0; v < 2 * D; ++v) f[v] = p[v];
            }
#pragma unroll
            for (int k = 0; k < D; ++k) r[k] = V2{f[k], f[D + k]};
        } else {
            const float* p0 = rec + cj[0] * D;
            const float* p1 = rec + cj[1] * D;
#pragma unroll
            for (int k = 0; k < D; ++k) r[k] = V2{p0[k], p1[k]};
        }
    }
    // c[i] = (M[i][c0], M[i][c1]): columns (adjacent in memory where both exist: one 8-byte load per row at even d)
    __device__ __forceinline__ void cols(const float* rec, V2 (&c)[D]) const {
        if constexpr (D % 2 == 0) {
#pragma unroll
            for (int i = 0; i < D; ++i) c[i] = *reinterpret_cast<const V2*>(rec + i * D + cj[0]);
        } else {
#pragma unroll
            for (int i = 0; i < D; ++i) c[i] = V2{rec[i * D + cj[0]], rec[i * D + cj[1]]};
        }
    }
    __device__ __forceinline__ void st_cols(float* rec, bool pred, const V2 (&c)[D]) const {
        if constexpr (D % 2 == 0) {
            if (pred && ok[0]) {
#pragma unroll
                for (int i = 0; i < D; ++i) *reinterpret_cast<V2*>(rec + i * D + cj[0]) = c[i];
            }
        } else if (pred && ok[0]) {
#pragma unroll
            for (int i = 0; i < D; ++i) rec[i * D + cj[0]] = c[i].x;
            if (ok[1]) {
#pragma unroll
                for (int i = 0; i < D; ++i) rec[i * D + cj[1]] = c[i].y;
            }
        }
    }
    // rec[cj][k] = c[k].{x,y}: the transpose of a column-layout matrix, row-major (once per chain: element by element)
    __device__ __forceinline__ void st_cols_t(float* rec, bool pred, const V2 (&c)[D]) const {
        if (pred && ok[0]) {
#pragma unroll
            for (int k = 0; k < D; ++k) rec[cj[0] * D + k] = c[k].x;
            if (ok[1]) {
#pragma unroll
                for (int k = 0; k < D; ++k) rec[cj[1] * D + k] = c[k].y;
            }
        }
    }
    __device__ __forceinline__ void st_vec(float* p, bool pred, const float (&v)[D]) const {
        if (pred && lead) {
#pragma unroll
            for (int i = 0; i < D; ++i) p[i] = v[i];
        }
    }
};
template <int D>
__device__ __forceinline__ void ld_vec(const float* p, float (&v)[D]) {
#pragma unroll
    for (int i = 0; i < D; ++i) v[i] = p[i];
}
// symmetric part of a row-major matrix, columns cj: c[i] = (M[i][cj] + M[cj][i]) / 2
template <int D>
__device__ __forceinline__ void ld_sym_cols(const Lane<D>& ln, const float* rec, V2 (&c)[D]) {
#pragma unroll
    for (int i = 0; i < D; ++i)
        c[i] = V2{0.5f * (rec[i * D + ln.cj[0]] + rec[ln.cj[0] * D + i]), 0.5f * (rec[i * D + ln.cj[1]] + rec[ln.cj[1] * D + i])};
}
template <int D>
__device__ __forceinline__ void ident2(const Lane<D>& ln, float dg, V2 (&z)[D]) {
#pragma unroll
    for (int i = 0; i < D; ++i) z[i] = V2{(i == ln.cj[0]) ? dg : 0.0f, (i == ln.cj[1]) ? dg : 0.0f};
}

__host__ __device__ inline int nfilt(int d) { return 3 * d * d + 2 * d; }      // [A | C | J | b | eta]
__host__ __device__ inline int nsmth(int d) { return 2 * d * d + d; }          // [E | L | g]

// FAST waves: all sixteen chains inside the series, the series' first step not among them, and the step after the last
// chain still inside the series -- no per-step predicates
__device__ __forceinline__ bool wave_fast(const rc::RcArgsT<float>& a) {
    const long wq = (a.N - 1) / ((long)kChains * a.Lw);
    return blockIdx.x >= 1 && (long)blockIdx.x < wq;
}

// ====================================================================================================
// level 1: reduce -- filt_extend over the chain (pgps_math.h filt_extend, parallel.py:46-72,100-118)
// ====================================================================================================
template <int D, bool FAST>
__device__ __forceinline__ void reduce1_body(const rc::RcArgsT<float>& a) {
    constexpr int dd = D * D;
    Lane<D> ln;
    ln.init(threadIdx.x);
    const long c = (long)blockIdx.x * kChains + (threadIdx.x >> 2);
    const long k0 = c * a.Lw, k1 = min(a.N, k0 + a.Lw);
    float h[D];
    ld_vec<D>(a.H, h);
    // chain 0 of the series starts from (0, 0, P0, 0, 0) and takes step 0 with F = I, Q = 0: that is filt_first
    const bool head = (c == 0 && a.seg_first);
    V2 A[D], C[D], J[D];
    float b[D], eta[D];
    ident2<D>(ln, head ? 0.0f : 1.0f, A);
    zero2<D>(C); zero2<D>(J);
    if (head) ld_sym_cols<D>(ln, a.P0, C);
#pragma unroll
    for (int i = 0; i < D; ++i) { b[i] = 0.0f; eta[i] = 0.0f; }
    V2 Fr[D], Fc[D], Q[D];
    float y;
    auto load = [&](int s) {
        const long k = k0 + s;
        const long kc = k < a.N ? k : a.N - 1;
        ln.rows(a.Fs + kc * dd, Fr);
        ln.cols(a.Fs + kc * dd, Fc);
        ln.cols(a.Qs + kc * dd, Q);
        y = a.ys[kc];
        if constexpr (!FAST) {
            const bool real = k < k1 && !(k == 0 && a.seg_first);
            if (!real) { ident2<D>(ln, 1.0f, Fr); ident2<D>(ln, 1.0f, Fc); zero2<D>(Q); }
            if (k >= k1) y = __builtin_nanf("");
        }
    };
    load(0);
    for (int s = 0; s < a.Lw; ++s) {
        V2 Ap[D], FC[D], Cp[D];
        float bp[D];
        zero2<D>(Ap); mm<D, false>(Ap, Fc, A);
        zero2<D>(FC); mm<D, false>(FC, Fc, C);
        copy2<D>(Cp, Q); mm<D, false>(Cp, FC, Fr);
        spread<D>(coldot<D>(Fr, b), bp);                 // (F b)[own rows], spread
        const float yk = y;
        load(s + 1 < a.Lw ? s + 1 : s);         // (clamped, not skipped: no branch inside the step)
        float u[D], v[D];
        const V2 ul = coldot<D>(Cp, h), vl = coldot<D>(Ap, h);
        spread<D>(ul, u); spread<D>(vl, v);
        const float S = dot<D>(h, u, a.R), hb = dot<D>(h, bp, 0.0f);
        const bool obs = !(yk != yk);
        const float inv = obs ? recip(S) : 0.0f;
        const float res = obs ? yk - hb : 0.0f;
        copy2<D>(A, Ap); rank1<D>(A, u, vl * (-inv));
        copy2<D>(C, Cp); rank1<D>(C, u, ul * (-inv));
        rank1<D>(J, v, vl * inv);
        const float ri = res * inv;
#pragma unroll
        for (int i = 0; i < D; ++i) { b[i] = __builtin_fmaf(u[i], ri, bp[i]); eta[i] = __builtin_fmaf(v[i], ri, eta[i]); }
    }
    if (c < a.nchunk) {
        float* rec = a.agg1 + c * nfilt(D);
        ln.st_cols(rec, true, A); ln.st_cols(rec + dd, true, C); ln.st_cols(rec + 2 * dd, true, J);
        ln.st_vec(rec + 3 * dd, true, b); ln.st_vec(rec + 3 * dd + D, true, eta);
    }
}

template <int D>
__global__ __launch_bounds__(64, PGPS_QC_WAVES) void q_reduce1(const rc::RcArgsT<float> a) {
    if (wave_fast(a)) reduce1_body<D, true>(a);
    else reduce1_body<D, false>(a);
}

// ====================================================================================================
// level 1: apply -- Kalman pass, log-likelihood, smoothing elements and the chain's smoothing total
// (kf_step / smth_element / smth_combine of pgps_math.h; parallel.py:135-151, 155-184)
// ====================================================================================================
template <int D, bool SMOOTH, bool FAST, bool STORE>
__device__ __forceinline__ void apply1_body(const rc::RcArgsT<float>& a) {
    constexpr int dd = D * D;
    Lane<D> ln;
    ln.init(threadIdx.x);
    const long c = (long)blockIdx.x * kChains + (threadIdx.x >> 2);
    const long k0 = c * a.Lw, k1 = min(a.N, k0 + a.Lw);
    const bool cv = c < a.nchunk;
    float h[D];
    ld_vec<D>(a.H, h);
    // state entering the chain: (b, C) of the inclusive prefix of the chain before (A = 0 there); the prior for chain 0
    // (a later segment of a sharded series enters its first chain with the carry-in of the ranks before it)
    float m[D];
    V2 P[D];
    {
        const bool pr = cv && (c > 0 || !a.seg_first);
        const float* rec = c > 0 ? a.pre + (cv ? c - 1 : 0) * nfilt(D) : (a.seg_first ? a.pre : a.carry);
        if (pr) { ld_vec<D>(rec + 3 * dd, m); ld_sym_cols<D>(ln, rec + dd, P); }
        else {
#pragma unroll
            for (int i = 0; i < D; ++i) m[i] = 0.0f;
            if (c == 0) ld_sym_cols<D>(ln, a.P0, P); else zero2<D>(P);
        }
    }
    // smoothing total of the steps seen so far, E transposed: Tt = E_tot^T (column layout), Ls, gt
    V2 Tt[D], Ls[D];
    float gt[D];
    if (SMOOTH) {
        ident2<D>(ln, 1.0f, Tt); zero2<D>(Ls);
#pragma unroll
        for (int i = 0; i < D; ++i) gt[i] = 0.0f;
    }
    LogLik ll;
    V2 Fr[D], Fc[D], Q[D];
    float y;
    // steps at or beyond N run with F = 0, Q = I: the element built from them is (0, m, P), i.e. the last element of the
    // series (parallel.py:155-156), and a total whose E is 0 absorbs whatever follows unchanged
    auto load = [&](int s) {
        const long k = k0 + s;
        const long kc = k < a.N ? k : a.N - 1;
        const float* pF = a.Fs + kc * dd;
        const float* pQ = a.Qs + kc * dd;
        if constexpr (!FAST) {
            // step N of a segment that is not the last: the first step of the next rank (halo), out of its record
            if (k == a.N && a.halo_F != nullptr) { pF = a.halo_F; pQ = a.halo_Q; }
        }
        ln.rows(pF, Fr);
        ln.cols(pF, Fc);
        ln.cols(pQ, Q);
        y = a.ys[kc];
        if constexpr (!FAST) {
            const bool real = k < a.N || (k == a.N && a.halo_F != nullptr);
            if (!real) { zero2<D>(Fr); zero2<D>(Fc); ident2<D>(ln, 1.0f, Q); }
            if (!(s < a.Lw && k < k1)) y = __builtin_nanf("");
        }
    };
    load(0);
    const int nsteps = SMOOTH ? a.Lw + 1 : a.Lw;
    for (int s = 0; s < nsteps; ++s) {
        const long k = k0 + s;
        const bool last = SMOOTH && s == a.Lw;          // the step after the chain: builds the last element, does not filter
        // predict
        V2 FP[D], Pp[D];
        float mp[D];
        zero2<D>(FP); mm<D, false>(FP, Fc, P);
        copy2<D>(Pp, Q); mm<D, false>(Pp, FP, Fr);
        spread<D>(coldot<D>(Fr, m), mp);
        const float yk = y;
        load(s + 1 < nsteps ? s + 1 : s);       // (clamped, not skipped: no branch inside the step)
        if (SMOOTH && s > 0) {
            // element of step k-1: W = Pp^-1 F P = E^T (E in row layout), g = m - E mp, L = P - E F P
            V2 M[D], W[D];
            copy2<D>(M, Pp); copy2<D>(W, FP);
            spd_solve<D>(M, W);
            float gn[D];
            spread<D>(V2{pick<D>(m, ln.cj[0]), pick<D>(m, ln.cj[1])} - coldot<D>(W, mp), gn);      // m - E mp
            V2 Ln[D], T[D];
            zero2<D>(T); mm<D, true>(T, W, FP);         // E (F P)
#pragma unroll
            for (int i = 0; i < D; ++i) Ln[i] = P[i] - T[i];
            {
                const bool st = FAST || (k - 1 < k1);
                ln.st_cols(a.Es + (k - 1) * dd, st, W);          // W = E^T, row-major: what the smoother's products take
                ln.st_cols(a.Lws + (k - 1) * dd, st, Ln);
                ln.st_vec(a.gs + (k - 1) * D, st, gn);
            }
            // total <- total (x) element, with Tt = E_tot^T:  Tt' = W Tt;  g' = Tt^T gn + gt;  L' = Tt^T (Ln Tt) + Ls
            // Beyond the end of a segment that is NOT the last of its series there is nothing to fold: the F = 0 steps
            // would put E = 0 into a total that the ranks after this one still have to extend.
            const bool fold = FAST || a.seg_last || (k - 1 < a.N);
            V2 X[D], T2[D], L2[D];
            float g2[D];
            zero2<D>(X); mm<D, false>(X, Ln, Tt);
            copy2<D>(L2, Ls); mm<D, true>(L2, Tt, X);
            spread<D>(coldot<D>(Tt, gn), g2);
            zero2<D>(T2); mm<D, false>(T2, W, Tt);
            if (fold) {
                copy2<D>(Ls, L2); copy2<D>(Tt, T2);
#pragma unroll
                for (int i = 0; i < D; ++i) gt[i] += g2[i];
            }
        }
        if (!last) {
            const bool upd = FAST || k < k1;
            const bool obs = !(yk != yk);
            float u[D];
            V2 ul = coldot<D>(Pp, h);
            spread<D>(ul, u);
            float S = dot<D>(h, u, a.R), mu = dot<D>(h, mp, 0.0f);
            if (obs) ll.add((double)yk - (double)mu, (double)S);
            if (!FAST && blockIdx.x == 0 && s == 0 && c == 0 && a.seg_first) {
                // first step of the series: the update uses the prior itself (parallel.py:24-30), the likelihood term
                // above used F0 P0 F0^T + Q0 (parallel.py:136-141)
                copy2<D>(Pp, P);
#pragma unroll
                for (int i = 0; i < D; ++i) mp[i] = m[i];
                ul = coldot<D>(Pp, h);
                spread<D>(ul, u);
                S = dot<D>(h, u, a.R); mu = dot<D>(h, mp, 0.0f);
            }
            const float inv = obs ? recip(S) : 0.0f;
            const float ri = (obs ? yk - mu : 0.0f) * inv;
#pragma unroll
            for (int i = 0; i < D; ++i) m[i] = __builtin_fmaf(u[i], ri, mp[i]);
            copy2<D>(P, Pp); rank1<D>(P, u, ul * (-inv));
            if constexpr (STORE) {
                ln.st_cols(a.fPs + k * dd, upd, P);
                ln.st_vec(a.fms + k * D, upd, m);
            }
        }
    }
    if (cv) {
        if (SMOOTH) {
            float* rec = a.sagg1 + c * nsmth(D);
            ln.st_cols_t(rec, true, Tt);                 // E_tot = Tt^T: Tt's columns are E_tot's rows
            ln.st_cols(rec + dd, true, Ls);
            ln.st_vec(rec + 2 * dd, true, gt);
        }
        if (ln.lead) a.llpart[c] = ll.value();
    }
}

template <int D, bool SMOOTH, bool STORE>
__global__ __launch_bounds__(64, PGPS_QC_WAVES) void q_apply1(const rc::RcArgsT<float> a) {
    if (wave_fast(a)) apply1_body<D, SMOOTH, true, STORE>(a);
    else apply1_body<D, SMOOTH, false, STORE>(a);
}

// ====================================================================================================
// level 1: smoother -- sm = E sm' + g, sP = E sP' E^T + L from the stored elements (parallel.py:176-184)
// ====================================================================================================
template <int D, bool FAST>
__device__ __forceinline__ void smooth1_body(const rc::RcArgsT<float>& a) {
    constexpr int dd = D * D;
    Lane<D> ln;
    ln.init(threadIdx.x);
    const long c = (long)blockIdx.x * kChains + (threadIdx.x >> 2);
    const long k0 = c * a.Lw, k1 = min(a.N, k0 + a.Lw);
    // smoothed moments of the first step after the chain: (g, L) of the inclusive suffix of the next chain (E = 0
    // there: every suffix contains the last element); nothing (0, 0) after the last chain
    float sm[D];
    V2 sP[D];
    {
        const bool inner = c + 1 < a.nchunk;
        const bool nx = inner || (c + 1 == a.nchunk && a.carry_back != nullptr);
        const float* rec = inner ? a.suf + (c + 1) * nsmth(D) : (a.carry_back ? a.carry_back : a.suf);
        if (nx) { ld_vec<D>(rec + 2 * dd, sm); ld_sym_cols<D>(ln, rec + dd, sP); }
        else {
#pragma unroll
            for (int i = 0; i < D; ++i) sm[i] = 0.0f;
            zero2<D>(sP);
        }
    }
    V2 W[D], L[D];
    float g[D];
    // stored element of the chain's step; steps outside the chain run as the identity element (I, 0, 0)
    auto load = [&](int s) {
        const long k = k0 + s;
        const long kc = k < a.N ? k : a.N - 1;
        ln.cols(a.Es + kc * dd, W);              // (q_apply1 stored W = E^T)
        ln.cols(a.Lws + kc * dd, L);
        ld_vec<D>(a.gs + kc * D, g);
        if constexpr (!FAST) {
            if (!(k < k1)) {
                ident2<D>(ln, 1.0f, W); zero2<D>(L);
#pragma unroll
                for (int i = 0; i < D; ++i) g[i] = 0.0f;
            }
        }
    };
    load(a.Lw - 1);
    for (int s = a.Lw - 1; s >= 0; --s) {
        const long k = k0 + s;
        V2 T[D], nP[D];
        zero2<D>(T); mm<D, true>(T, W, sP);             // E sP
        copy2<D>(nP, L); mm<D, false>(nP, T, W);        // + (E sP) E^T
        const V2 sl = coldot<D>(W, sm) + V2{pick<D>(g, ln.cj[0]), pick<D>(g, ln.cj[1])};      // (E sm + g)[own rows]
        load(s > 0 ? s - 1 : 0);
        spread<D>(sl, sm);
        copy2<D>(sP, nP);
        const bool st = FAST || k < k1;
        ln.st_cols(a.sPs + k * dd, st, sP);
        ln.st_vec(a.sms + k * D, st, sm);
    }
}

template <int D>
__global__ __launch_bounds__(64, PGPS_QC_WAVES) void q_smooth1(const rc::RcArgsT<float> a) {
    if (wave_fast(a)) smooth1_body<D, true>(a);
    else smooth1_body<D, false>(a);
}

// ---- host side ---------------------------------------------------------------------------------------
// phase 0: reduce, 1: apply + smoothing elements, 2: apply only, 3: smoother
template <int D>
int launch_qc_level1(pgps_ctx* ctx, const rc::RcArgsT<float>& a, int phase) {
    const dim3 blk(64), g1((unsigned)((a.nchunk + kChains - 1) / kChains));
    const bool st = a.store_f != 0;
    switch (phase) {
        case 0: timed_launch(ctx, PGPS_K_FILTER_REDUCE, q_reduce1<D>, g1, blk, 0u, a); break;
        case 1:
            if (st) timed_launch(ctx, PGPS_K_FILTER_APPLY, q_apply1<D, true, true>, g1, blk, 0u, a);
            else timed_launch(ctx, PGPS_K_FILTER_APPLY, q_apply1<D, true, false>, g1, blk, 0u, a);
            break;
        case 2:
            if (st) timed_launch(ctx, PGPS_K_FILTER_APPLY, q_apply1<D, false, true>, g1, blk, 0u, a);
            else timed_launch(ctx, PGPS_K_FILTER_APPLY, q_apply1<D, false, false>, g1, blk, 0u, a);
            break;
        case 3: timed_launch(ctx, PGPS_K_SMOOTHER_APPLY, q_smooth1<D>, g1, blk, 0u, a); break;
        default: return PGPS_E_INVALID;
    }
    HIPCHK(ctx, hipGetLastError());
    return PGPS_OK;
}

}  // namespace qc
}  // namespace pgps
