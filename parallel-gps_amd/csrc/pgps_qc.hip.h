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
// next step's inputs requested right after the predict (their registers are free again from there: the loads fly under the
// rest of the step) or at the end of the step (36 registers less alive through the element and total phases)
#ifndef PGPS_QC_PREFETCH_EARLY
#define PGPS_QC_PREFETCH_EARLY 1
#endif
constexpr bool kPrefetchEarly = PGPS_QC_PREFETCH_EARLY != 0;
// Waves per SIMD each kernel is compiled for (register budget 512 / waves; PGPS_QC_WAVES forces one value on all).  The
// filter + smoothing-element pass holds a dozen 2 d-register matrices and spills on 256 registers at every d (d = 6:
// 100 B a lane inside the step loop, 450 us at 2^20 steps against 240-270 us on 512); the reduce pass fits 256 up to d = 6
// (232 registers, no spills), the smoother and the filter-only pass fit everywhere (116 / 154 at d = 6) -- a grid of 1024
// waves puts one on every SIMD either way, so the smaller budgets cost nothing.  profiles/r03_experiments.txt.
#ifdef PGPS_QC_WAVES
constexpr int waves_reduce(int) { return PGPS_QC_WAVES; }
constexpr int waves_apply(int, bool) { return PGPS_QC_WAVES; }
constexpr int waves_smooth(int) { return PGPS_QC_WAVES; }
#else
constexpr int waves_reduce(int d) { return d <= 6 ? 2 : 1; }
constexpr int waves_apply(int, bool smooth) { return smooth ? 1 : 2; }
constexpr int waves_smooth(int) { return 2; }
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
// INIT: z = X Y instead of z += X Y (the first term is a v_pk_mul_f32: no zeroing of the accumulators)
template <int D, bool LT, bool INIT = false>
__device__ __forceinline__ void mm(V2 (&z)[D], const V2 (&x)[D], const V2 (&yk)[D]) {
    auto X = [&](int i, int k) { return LT ? qbi(comp(x[k], i & 1), i >> 1) : qbi(comp(x[i], k & 1), k >> 1); };
#pragma unroll
    for (int k = 0; k < D; ++k) {
        // two broadcasts share a register pair (v_pk_fma_f32 takes its scalar operand as half of a 64-bit pair: one
        // broadcast per pair would waste the other half -- 2 d^2 registers of a product's operands in flight)
#pragma unroll
        for (int i = 0; i + 1 < D; i += 2) {
            const V2 xp = V2{X(i, k), X(i + 1, k)};
            if (INIT && k == 0) {
                z[i] = V2{xp.x, xp.x} * yk[k];
                z[i + 1] = V2{xp.y, xp.y} * yk[k];
            } else {
                z[i] = __builtin_elementwise_fma(V2{xp.x, xp.x}, yk[k], z[i]);
                z[i + 1] = __builtin_elementwise_fma(V2{xp.y, xp.y}, yk[k], z[i + 1]);
            }
        }
        if constexpr (D & 1) {
            const float xl = X(D - 1, k);
            z[D - 1] = (INIT && k == 0) ? V2{xl, xl} * yk[k] : fma2(xl, yk[k], z[D - 1]);
        }
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

// ---- whole records as 16-byte pieces through LDS ----------------------------------------------------------------
// A lane's own accesses to its chain's record are 8 bytes wide and sixteen chains apart: every memory instruction of a
// step touches sixteen cache lines for 512 useful bytes, and the kernels are bound by the number of such instructions
// (the row-cooperative family measured the same: pgps_rc.hip.h Io).  FAST waves at even d therefore move the sixteen
// records of a step as 16-byte pieces, lane t of instruction v carrying piece v * 64 + t (piece j of chain r:
// r * NPC + j), through an LDS slot that holds the records back to back; the lanes pick their columns / rows out of it.
// d = 6: 3 instructions per matrix instead of 6 (columns) or 12 (rows + columns); vectors: 1 instead of 6.
#ifndef PGPS_QC_WIDE
#define PGPS_QC_WIDE 1
#endif
constexpr unsigned kOob = 0x7ffff000u;
// (every kernel that calls this is ONE wave per workgroup and its LDS is that wave's own: the LDS executes a wave's
// instructions in issue order, so only the compiler has to be held to program order -- wavefront scope.  Until round 5 these
// were workgroup-scope fences, which on gfx950 also drain the wave's vector-memory operations (s_waitcnt vmcnt(0)): the
// prefetched next step and the previous step's stores were waited for at every LDS hand-over.  -DPGPS_LDS_SYNC_WORKGROUP
// restores that for A/B runs.  LDS-DMA fetches are waited for explicitly where they are consumed: dma_wait_all.)
__device__ __forceinline__ void wsync() {
#ifdef PGPS_LDS_SYNC_WORKGROUP
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}
template <int D>
struct Wide {
    static constexpr bool kOn = (D % 2 == 0) && (PGPS_QC_WIDE != 0);
    static constexpr int REC = D * D * 4, NPC = REC / 16, NV = (kChains * NPC + 63) / 64, SLOT = NV * 1024;
    static constexpr int VREC = D * 4, NPV = VREC / 8, VSLOT = 512;        // vectors: 8-byte pieces, one instruction
    using V4 = __attribute__((ext_vector_type(4))) unsigned int;
    using U2 = __attribute__((ext_vector_type(2))) unsigned int;
    unsigned wg[NV], wv, lrec, lvec, span_m, span_v;
    int tid;
    __device__ __forceinline__ void init(int t, int Lw) {
        tid = t & 63;
        const unsigned pitch = (unsigned)Lw * (unsigned)REC, vpitch = (unsigned)Lw * (unsigned)VREC;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int q = v * 64 + tid;
            wg[v] = q < kChains * NPC ? (unsigned)(q / NPC) * pitch + (unsigned)(q % NPC) * 16u : kOob;
        }
        wv = tid < kChains * NPV ? (unsigned)(tid / NPV) * vpitch + (unsigned)(tid % NPV) * 8u : kOob;
        lrec = (unsigned)(tid >> 2) * (unsigned)REC;
        lvec = (unsigned)(tid >> 2) * (unsigned)VREC;
        span_m = (unsigned)(kChains - 1) * pitch + (unsigned)REC;
        span_v = (unsigned)(kChains - 1) * vpitch + (unsigned)VREC;
    }
    static __device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, unsigned bytes) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
    }
    // the bytes an access may touch from its base: everything the sixteen chains span (FAST waves), or what is left of
    // the array (`left` steps from the base's one to the end; none at all when the base is beyond it)
    template <bool FAST>
    __device__ __forceinline__ unsigned limit(long left) const {
        if constexpr (FAST) return span_m;
        const long b = left * (long)REC;
        return b <= 0 ? 0u : (b < (long)span_m ? (unsigned)b : span_m);
    }
    template <bool FAST>
    __device__ __forceinline__ unsigned vlimit(long left) const {
        if constexpr (FAST) return span_v;
        const long b = left * (long)VREC;
        return b <= 0 ? 0u : (b < (long)span_v ? (unsigned)b : span_v);
    }
    // base = the record of the wave's first chain at the step
    __device__ __forceinline__ void load(const float* base, unsigned lim, V4 (&r)[NV]) const {
        const __amdgpu_buffer_rsrc_t rs = rsrc(base, lim);
#pragma unroll
        for (int v = 0; v < NV; ++v) r[v] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)wg[v], 0, 0);
    }
    __device__ __forceinline__ void commit(char* slot, const V4 (&r)[NV]) const {
#pragma unroll
        for (int v = 0; v < NV; ++v) *reinterpret_cast<V4*>(slot + (v * 64 + tid) * 16) = r[v];
    }
    __device__ __forceinline__ void cols(const char* slot, const Lane<D>& ln, V2 (&c)[D]) const {
        const char* p = slot + lrec + ln.cj[0] * 4;
#pragma unroll
        for (int i = 0; i < D; ++i) c[i] = *reinterpret_cast<const V2*>(p + i * D * 4);
    }
    __device__ __forceinline__ void rows(const char* slot, const Lane<D>& ln, V2 (&r)[D]) const {
        const char* p0 = slot + lrec + ln.cj[0] * D * 4;
        const char* p1 = slot + lrec + ln.cj[1] * D * 4;
#pragma unroll
        for (int k = 0; k < D; ++k) r[k] = V2{*reinterpret_cast<const float*>(p0 + 4 * k), *reinterpret_cast<const float*>(p1 + 4 * k)};
    }
    __device__ __forceinline__ void put_cols(char* slot, const Lane<D>& ln, const V2 (&c)[D]) const {
        if (ln.ok[0]) {
            char* p = slot + lrec + ln.cj[0] * 4;
#pragma unroll
            for (int i = 0; i < D; ++i) *reinterpret_cast<V2*>(p + i * D * 4) = c[i];
        }
    }
    __device__ __forceinline__ void drain(float* base, unsigned lim, const char* slot) const {
        const __amdgpu_buffer_rsrc_t rs = rsrc(base, lim);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const V4 x = *reinterpret_cast<const V4*>(slot + (v * 64 + tid) * 16);
            __builtin_amdgcn_raw_buffer_store_b128(x, rs, (int)wg[v], 0, 0);
        }
    }
    __device__ __forceinline__ U2 vload(const float* base, unsigned lim) const {
        return __builtin_amdgcn_raw_buffer_load_b64(rsrc(base, lim), (int)wv, 0, 0);
    }
    __device__ __forceinline__ void vcommit(char* vslot, U2 x) const { *reinterpret_cast<U2*>(vslot + tid * 8) = x; }
    __device__ __forceinline__ void vget(const char* vslot, float (&v)[D]) const {
#pragma unroll
        for (int i = 0; i < D / 2; ++i) {
            const V2 t = *reinterpret_cast<const V2*>(vslot + lvec + 8 * i);
            v[2 * i] = t.x; v[2 * i + 1] = t.y;
        }
    }
    __device__ __forceinline__ void vput(char* vslot, const Lane<D>& ln, const float (&v)[D]) const {
        if (ln.lead) {
#pragma unroll
            for (int i = 0; i < D / 2; ++i) *reinterpret_cast<V2*>(vslot + lvec + 8 * i) = V2{v[2 * i], v[2 * i + 1]};
        }
    }
    __device__ __forceinline__ void vdrain(float* base, unsigned lim, const char* vslot) const {
        const U2 x = *reinterpret_cast<const U2*>(vslot + tid * 8);
        __builtin_amdgcn_raw_buffer_store_b64(x, rsrc(base, lim), (int)wv, 0, 0);
    }
};

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
// Edge waves (the first and the last ones of a series) run the same road as the others -- whole records through LDS, the
// buffer range cut at the end of the array so that what lies beyond reads as zeros and is not stored -- plus a handful of
// selects per step; only odd d (records that are not whole 16-byte pieces) goes lane by lane.
template <int D, bool FAST>
__device__ __forceinline__ void reduce1_body(const rc::RcArgsT<float>& a, char* lds) {
    constexpr int dd = D * D;
    using Wd = Wide<D>;
    constexpr bool WD = Wd::kOn;
    Wd w;
    typename Wd::V4 rf[Wd::NV], rq[Wd::NV];
    const long kw = (long)blockIdx.x * kChains * a.Lw;      // first step of the wave's first chain
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
    // steps outside the chain, and the series' first one, run as the identity transition without a measurement
    auto fix = [&](int s) {
        if constexpr (!FAST) {
            const long k = k0 + s;
            const bool real = k < k1 && !(k == 0 && a.seg_first);
            if (!real) { ident2<D>(ln, 1.0f, Fr); ident2<D>(ln, 1.0f, Fc); zero2<D>(Q); }
        }
    };
    auto load_y = [&](int s) {
        const long k = k0 + s;
        y = a.ys[FAST || k < a.N ? k : a.N - 1];
        if constexpr (!FAST) { if (k >= k1) y = __builtin_nanf(""); }
    };
    auto load = [&](int s) {                    // lane by lane (odd d)
        const long k = k0 + s;
        const long kc = k < a.N ? k : a.N - 1;
        ln.rows(a.Fs + kc * dd, Fr);
        ln.cols(a.Fs + kc * dd, Fc);
        ln.cols(a.Qs + kc * dd, Q);
        load_y(s);
        fix(s);
    };
    auto issue = [&](int s) {                   // whole records, a step ahead
        const unsigned lim = w.template limit<FAST>(a.N - (kw + s));
        w.load(a.Fs + (kw + s) * dd, lim, rf); w.load(a.Qs + (kw + s) * dd, lim, rq);
    };
    if constexpr (WD) { w.init(threadIdx.x, a.Lw); issue(0); load_y(0); }
    else load(0);
    for (int s = 0; s < a.Lw; ++s) {
        const int sn = s + 1 < a.Lw ? s + 1 : s;        // (clamped, not skipped: no branch inside the step)
        if constexpr (WD) {
            wsync();
            w.commit(lds, rf); w.commit(lds + Wd::SLOT, rq);
            wsync();
            w.rows(lds, ln, Fr); w.cols(lds, ln, Fc); w.cols(lds + Wd::SLOT, ln, Q);
            fix(s);
            issue(sn);
            __builtin_amdgcn_sched_barrier(0);
        }
        V2 Ap[D], FC[D], Cp[D];
        float bp[D];
        mm<D, false, true>(Ap, Fc, A);
        mm<D, false, true>(FC, Fc, C);
        copy2<D>(Cp, Q); mm<D, false>(Cp, FC, Fr);
        spread<D>(coldot<D>(Fr, b), bp);                 // (F b)[own rows], spread
        const float yk = y;
        if constexpr (WD) load_y(sn);
        else load(sn);
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
__global__ __launch_bounds__(64, waves_reduce(D)) void q_reduce1(const rc::RcArgsT<float> a) {
    __shared__ __attribute__((aligned(16))) char lds[Wide<D>::kOn ? 2 * Wide<D>::SLOT : 16];
    if (wave_fast(a)) reduce1_body<D, true>(a, lds);
    else reduce1_body<D, false>(a, lds);
}

// ====================================================================================================
// level 1: apply -- Kalman pass, log-likelihood, smoothing elements and the chain's smoothing total
// (kf_step / smth_element / smth_combine of pgps_math.h; parallel.py:135-151, 155-184)
// ====================================================================================================
template <int D, bool SMOOTH, bool FAST, bool STORE>
__device__ __forceinline__ void apply1_body(const rc::RcArgsT<float>& a, char* lds) {
    constexpr int dd = D * D;
    using Wd = Wide<D>;
    constexpr bool WD = Wd::kOn;
    Wd w;
    typename Wd::V4 rf[Wd::NV], rq[Wd::NV];
    const long kw = (long)blockIdx.x * kChains * a.Lw;
    // LDS: [F | Q] in, [W | Ln | P] out, [gn | m] vectors out
    char* const sF = lds; char* const sQ = lds + Wd::SLOT;
    char* const sW = lds + 2 * Wd::SLOT; char* const sL = lds + 3 * Wd::SLOT; char* const sP_ = lds + 4 * Wd::SLOT;
    char* const sg = lds + 5 * Wd::SLOT; char* const sm_ = sg + Wd::VSLOT;
    Lane<D> ln;
    ln.init(threadIdx.x);
    const long c = (long)blockIdx.x * kChains + (threadIdx.x >> 2);
    const long k0 = c * a.Lw, k1 = min(a.N, k0 + a.Lw);
    const bool cv = c < a.nchunk;
    float h[D];
    ld_vec<D>(a.H, h);
    // state entering the chain: (b, C) of the inclusive prefix of the chain before (A = 0 there); the prior for chain 0
    // (a later segment of a sharded series enters its first chain with the carry-in of the ranks before it).
    // The mean is carried twice: replicated (m) and as the lane's own elements (ml) -- what `m - E mp` takes.
    float m[D];
    V2 ml;
    V2 P[D];
    {
        const bool pr = cv && (c > 0 || !a.seg_first);
        const float* rec = c > 0 ? a.pre + (cv ? c - 1 : 0) * nfilt(D) : (a.seg_first ? a.pre : a.carry);
        if (pr) { ld_vec<D>(rec + 3 * dd, m); ld_sym_cols<D>(ln, rec + dd, P); ml = V2{rec[3 * dd + ln.cj[0]], rec[3 * dd + ln.cj[1]]}; }
        else {
#pragma unroll
            for (int i = 0; i < D; ++i) m[i] = 0.0f;
            ml = V2{0.0f, 0.0f};
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
    auto fix = [&](int s) {
        if constexpr (!FAST) {
            const long k = k0 + s;
            // step N of a segment that is not the last: the first step of the next rank (halo), out of its record
            const bool halo = k == a.N && a.halo_F != nullptr;
            if (halo) { ln.rows(a.halo_F, Fr); ln.cols(a.halo_F, Fc); ln.cols(a.halo_Q, Q); }
            if (!(k < a.N || halo)) { zero2<D>(Fr); zero2<D>(Fc); ident2<D>(ln, 1.0f, Q); }
        }
    };
    auto load_y = [&](int s) {
        const long k = k0 + s;
        y = a.ys[(FAST && s < a.Lw) || k < a.N ? k : a.N - 1];
        if constexpr (!FAST) { if (!(s < a.Lw && k < k1)) y = __builtin_nanf(""); }
    };
    auto load = [&](int s) {                    // lane by lane (odd d)
        const long k = k0 + s;
        const long kc = k < a.N ? k : a.N - 1;
        ln.rows(a.Fs + kc * dd, Fr);
        ln.cols(a.Fs + kc * dd, Fc);
        ln.cols(a.Qs + kc * dd, Q);
        load_y(s);
        fix(s);
    };
    auto issue = [&](int s) {
        const unsigned lim = w.template limit<FAST>(a.N - (kw + s));
        w.load(a.Fs + (kw + s) * dd, lim, rf); w.load(a.Qs + (kw + s) * dd, lim, rq);
    };
    const int nsteps = SMOOTH ? a.Lw + 1 : a.Lw;
    if constexpr (WD) { w.init(threadIdx.x, a.Lw); issue(0); load_y(0); }
    else load(0);
    for (int s = 0; s < nsteps; ++s) {
        const long k = k0 + s;
        const bool last = SMOOTH && s == a.Lw;          // the step after the chain: builds the last element, does not filter
        const int sn = s + 1 < nsteps ? s + 1 : s;      // (clamped, not skipped: no branch inside the step)
        if constexpr (WD) {
            wsync();
            w.commit(sF, rf); w.commit(sQ, rq);
            wsync();
            w.rows(sF, ln, Fr); w.cols(sF, ln, Fc); w.cols(sQ, ln, Q);
            fix(s);
            issue(sn);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- phase A: predict; element of step k-1 (W = Pp^-1 F P = E^T, i.e. E in row layout; g = m - E mp; L = P - E F P)
        V2 FP[D], Pp[D];
        float mp[D];
        mm<D, false, true>(FP, Fc, P);
        copy2<D>(Pp, Q); mm<D, false>(Pp, FP, Fr);
        const V2 mpl = coldot<D>(Fr, m);                // (F m)[own rows]
        spread<D>(mpl, mp);
        const float yk = y;
        if constexpr (WD) load_y(sn);
        else if constexpr (kPrefetchEarly) load(sn);
        V2 W[D], Ln[D];
        float gn[D];
        const bool elem = SMOOTH && s > 0;
        if (elem) {
            V2 M[D];
            copy2<D>(M, Pp); copy2<D>(W, FP);
            spd_solve<D>(M, W);
            spread<D>(ml - coldot<D>(W, mp), gn);       // m - E mp
            V2 T[D];
            mm<D, true, true>(T, W, FP);         // E (F P)
#pragma unroll
            for (int i = 0; i < D; ++i) Ln[i] = P[i] - T[i];
            if constexpr (WD) {
                const long left = a.N - (kw + s - 1);
                wsync();
                w.put_cols(sW, ln, W); w.put_cols(sL, ln, Ln); w.vput(sg, ln, gn);
                wsync();
                w.drain(a.Es + (kw + s - 1) * dd, w.template limit<FAST>(left), sW);      // W = E^T, row-major: what the
                w.drain(a.Lws + (kw + s - 1) * dd, w.template limit<FAST>(left), sL);     // smoother's products take
                w.vdrain(a.gs + (kw + s - 1) * D, w.template vlimit<FAST>(left), sg);
            } else {
                const bool st = FAST || (k - 1 < k1);
                ln.st_cols(a.Es + (k - 1) * dd, st, W);
                ln.st_cols(a.Lws + (k - 1) * dd, st, Ln);
                ln.st_vec(a.gs + (k - 1) * D, st, gn);
            }
        }
        // (the three phases of a step are kept apart: interleaved by the scheduler, their operands -- a dozen 2 d-register
        // matrices -- are all alive at once and the kernel spills)
        __builtin_amdgcn_sched_barrier(0);
        // ---- phase B: Kalman update
        if (!last) {
            const bool upd = FAST || k < k1;
            const bool obs = !(yk != yk);
            float u[D];
            V2 ul = coldot<D>(Pp, h);
            spread<D>(ul, u);
            float S = dot<D>(h, u, a.R), mu = dot<D>(h, mp, 0.0f);
            if (obs) ll.add((double)yk - (double)mu, (double)S);
            V2 mq = mpl;
            if (!FAST && blockIdx.x == 0 && s == 0 && c == 0 && a.seg_first) {
                // first step of the series: the update uses the prior itself (parallel.py:24-30), the likelihood term
                // above used F0 P0 F0^T + Q0 (parallel.py:136-141)
                copy2<D>(Pp, P);
#pragma unroll
                for (int i = 0; i < D; ++i) mp[i] = m[i];
                mq = ml;
                ul = coldot<D>(Pp, h);
                spread<D>(ul, u);
                S = dot<D>(h, u, a.R); mu = dot<D>(h, mp, 0.0f);
            }
            const float inv = obs ? recip(S) : 0.0f;
            const float ri = (obs ? yk - mu : 0.0f) * inv;
#pragma unroll
            for (int i = 0; i < D; ++i) m[i] = __builtin_fmaf(u[i], ri, mp[i]);
            ml = fma2(ri, ul, mq);
            copy2<D>(P, Pp); rank1<D>(P, u, ul * (-inv));
            if constexpr (STORE && WD) {
                const long left = a.N - (kw + s);
                wsync();
                w.put_cols(sP_, ln, P); w.vput(sm_, ln, m);
                wsync();
                w.drain(a.fPs + (kw + s) * dd, w.template limit<FAST>(left), sP_);
                w.vdrain(a.fms + (kw + s) * D, w.template vlimit<FAST>(left), sm_);
            } else if constexpr (STORE) {
                ln.st_cols(a.fPs + k * dd, upd, P);
                ln.st_vec(a.fms + k * D, upd, m);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- phase C: total <- total (x) element, with Tt = E_tot^T:  L' = Ls + Tt^T (Ln Tt);  g' = gt + Tt^T gn;  Tt' = W Tt
        if (elem) {
            V2 X[D];
            float g2[D];
            mm<D, false, true>(X, Ln, Tt);
            spread<D>(coldot<D>(Tt, gn), g2);
            // Beyond the end of a segment that is NOT the last of its series there is nothing to fold: the F = 0 steps
            // would put E = 0 into a total that the ranks after this one still have to extend.  (A whole quad takes the
            // branch or none of it does: the broadcasts inside stay within the quad.)
            if (FAST || a.seg_last || (k - 1 < a.N)) {
                mm<D, true>(Ls, Tt, X);
#pragma unroll
                for (int i = 0; i < D; ++i) gt[i] += g2[i];
                mm<D, false, true>(X, W, Tt);
                copy2<D>(Tt, X);
            }
        }
        if constexpr (!WD && !kPrefetchEarly) load(sn);
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
__global__ __launch_bounds__(64, waves_apply(D, SMOOTH)) void q_apply1(const rc::RcArgsT<float> a) {
    __shared__ __attribute__((aligned(16))) char lds[Wide<D>::kOn ? 5 * Wide<D>::SLOT + 2 * Wide<D>::VSLOT : 16];
    if (wave_fast(a)) apply1_body<D, SMOOTH, true, STORE>(a, lds);
    else apply1_body<D, SMOOTH, false, STORE>(a, lds);
}

// ====================================================================================================
// level 1: smoother -- sm = E sm' + g, sP = E sP' E^T + L from the stored elements (parallel.py:176-184)
// ====================================================================================================
template <int D, bool FAST>
__device__ __forceinline__ void smooth1_body(const rc::RcArgsT<float>& a, char* lds) {
    constexpr int dd = D * D;
    using Wd = Wide<D>;
    constexpr bool WD = Wd::kOn;
    Wd w;
    typename Wd::V4 rw[Wd::NV], rl[Wd::NV];
    typename Wd::U2 rg;
    const long kw = (long)blockIdx.x * kChains * a.Lw;
    char* const sW = lds; char* const sL = lds + Wd::SLOT; char* const sO = lds + 2 * Wd::SLOT;
    char* const sg = lds + 3 * Wd::SLOT; char* const so = sg + Wd::VSLOT;
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
    V2 W[D], L[D], gl;
    // stored element of the chain's step; steps outside the chain run as the identity element (I, 0, 0)
    auto fix = [&](int s) {
        if constexpr (!FAST) {
            if (!(k0 + s < k1)) { ident2<D>(ln, 1.0f, W); zero2<D>(L); gl = V2{0.0f, 0.0f}; }
        }
    };
    auto load = [&](int s) {                    // lane by lane (odd d)
        const long k = k0 + s;
        const long kc = k < a.N ? k : a.N - 1;
        ln.cols(a.Es + kc * dd, W);              // (q_apply1 stored W = E^T)
        ln.cols(a.Lws + kc * dd, L);
        gl = V2{a.gs[kc * D + ln.cj[0]], a.gs[kc * D + ln.cj[1]]};
        fix(s);
    };
    auto issue = [&](int s) {
        const long left = a.N - (kw + s);
        w.load(a.Es + (kw + s) * dd, w.template limit<FAST>(left), rw); w.load(a.Lws + (kw + s) * dd, w.template limit<FAST>(left), rl);
        rg = w.vload(a.gs + (kw + s) * D, w.template vlimit<FAST>(left));
    };
    if constexpr (WD) { w.init(threadIdx.x, a.Lw); issue(a.Lw - 1); }
    else load(a.Lw - 1);
    for (int s = a.Lw - 1; s >= 0; --s) {
        const long k = k0 + s;
        const int sn = s > 0 ? s - 1 : 0;
        if constexpr (WD) {
            wsync();
            w.commit(sW, rw); w.commit(sL, rl); w.vcommit(sg, rg);
            wsync();
            w.cols(sW, ln, W); w.cols(sL, ln, L);
            gl = *reinterpret_cast<const V2*>(sg + w.lvec + ln.cj[0] * 4);      // (columns in pairs at even d)
            fix(s);
            issue(sn);
            __builtin_amdgcn_sched_barrier(0);
        }
        V2 T[D], nP[D];
        mm<D, true, true>(T, W, sP);             // E sP
        copy2<D>(nP, L); mm<D, false>(nP, T, W);        // + (E sP) E^T
        const V2 sl = coldot<D>(W, sm) + gl;            // (E sm + g)[own rows]
        if constexpr (!WD) load(sn);
        spread<D>(sl, sm);
        copy2<D>(sP, nP);
        if constexpr (WD) {
            const long left = a.N - (kw + s);
            wsync();
            w.put_cols(sO, ln, sP); w.vput(so, ln, sm);
            wsync();
            w.drain(a.sPs + (kw + s) * dd, w.template limit<FAST>(left), sO);
            w.vdrain(a.sms + (kw + s) * D, w.template vlimit<FAST>(left), so);
        } else {
            const bool st = FAST || k < k1;
            ln.st_cols(a.sPs + k * dd, st, sP);
            ln.st_vec(a.sms + k * D, st, sm);
        }
    }
}

template <int D>
__global__ __launch_bounds__(64, waves_smooth(D)) void q_smooth1(const rc::RcArgsT<float> a) {
    __shared__ __attribute__((aligned(16))) char lds[Wide<D>::kOn ? 3 * Wide<D>::SLOT + 2 * Wide<D>::VSLOT : 16];
    if (wave_fast(a)) smooth1_body<D, true>(a, lds);
    else smooth1_body<D, false>(a, lds);
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
