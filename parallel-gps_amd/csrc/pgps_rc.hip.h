// pgps_rc.hip.h -- the "row-cooperative" scan family: fp64 and fp32, state dimensions 2..16, one instantiation per
// (scalar type, d).
// Compiled by pgps_rc_inst.hip (-DPGPS_RC_D=d); the host driver that strings the launches together is in pgps_wc.hip
// (scan_rc / scan_rc_seg / launch_disc_rc).
//
// Between the lane-chunk family (one lane owns whole d x d operands, d <= 6) and the wave-cooperative one
// (64 lanes share LDS-resident operands, d <= 32) sits the case the c5 config (d = 11) and RBF order 15 live in:
// operands too large for one lane, yet small enough that LDS tiles starve the fp64 pipes (two LDS reads per
// 2 x 2 tile step).  Here ONE 16-LANE DPP ROW owns a chain of consecutive time steps and LANE j HOLDS COLUMN j
// of every operand in registers (D doubles per matrix); a wavefront runs four chains.
// A product Z = X Y is D^2 instructions per lane, every one a v_fmac_f64_dpp whose first operand is lane k's
// register broadcast to the row (row_newbcast:k) -- no LDS, no shuffles, no extra moves:
//     Z_i(lane j) += bcast_k(X_i) * Y_k(lane j)          (pgps_rc_asm.h, generated)
// Products with a transposed right operand take that operand in row layout (lane j holds row j), which for the
// per-step inputs is simply a second load of the same 8 d^2 bytes; only three or four operands per step go
// through a 2 KB LDS patch to be transposed.  Rank-one updates, matrix-vector products, row sums and the
// Gauss-Jordan eliminations are the same broadcast-fmac pattern.  Lanes >= D of a row hold zeros (their loads are
// masked off by EXEC and nothing else ever writes them).  MFMA is not used.
//
// Kernels (same algebra as pgps_math.h, reference pssgp/kalman/parallel.py:13-196):
//   rc_discretise   Fs, Qs from the time stamps (Pade scaling and squaring, one row per step)
//   rc_reduce1      chain of Lw steps: filt_extend per step                    -> chain totals (A, b, C, J, eta)
//   rc_ks_filter    one Kogge-Stone step over the totals, one row per record (general operator, pivoted elimination);
//                   with a fixed left operand: a segment's carry-in combined into every prefix
//   rc_apply1       Kalman pass from the prefix's (b, C) [every prefix has A = 0]: fms, fPs, log-lik partials;
//                   per step the smoothing element (E, g, L) is built ONCE, stored and folded into the chain's
//                   smoothing total
//   rc_selem1       the element part of rc_apply1 from GIVEN filtered moments (stand-alone pks)
//   rc_ks_smoother  one Kogge-Stone step over the smoothing totals (suffix direction)
//   rc_smooth1      backward pass  sm = E sm' + g,  sP = E sP' E^T + L  from the stored elements (two products
//                   per step instead of predict + gain solve + two products); PROJ: only H sm, H sP H^T at marked steps
//   rc_seg_carry_f / _s   carry records of a segment from the gathered totals of the other ranks (multi-GPU)
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

#include "pgps_internal.h"
#include "pgps_math.h"
#include "pgps_rc_asm.h"

namespace pgps {
namespace rc {

template <typename T> struct Ident { using type = T; };
template <typename T> using Id = typename Ident<T>::type;      // a parameter that takes no part in deducing T

constexpr int kLdT = 17;                    // leading dimension of a transpose patch (16 lanes + 1)
constexpr int kPatch = 16 * kLdT;           // doubles per row patch

// (every kernel that calls this is ONE wave per workgroup and its LDS is that wave's own: the LDS executes a wave's
// instructions in issue order, so only the compiler has to be held to program order -- wavefront scope.  Until round 5 these
// were workgroup-scope fences, which on gfx950 also drain the wave's vector-memory operations (s_waitcnt vmcnt(0)): the
// prefetched next step and the previous step's stores were waited for at every LDS hand-over.  -DPGPS_LDS_SYNC_WORKGROUP
// restores that for A/B runs.  LDS-DMA fetches are waited for explicitly where they are consumed: dma_wait_all.)
__device__ __forceinline__ void sync() {
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
__host__ __device__ inline int nfilt(int d) { return 3 * d * d + 2 * d; }      // [A | C | J | b | eta], as pgps_wc.hip
__host__ __device__ inline int nsmth(int d) { return 2 * d * d + d; }          // [E | L | g]

// z += X Y  (z pre-loaded with the addend; z must not alias x or y)
template <int D, typename Real>
__device__ __forceinline__ void mm(Real* z, const Real* x, const Real* y) {
#pragma unroll
    for (int i = 0; i + 4 <= D; i += 4) Asm<Real, D>::rows4(z + i, x + i, y);
    constexpr int R = D % 4, I0 = D - R;
    if constexpr (R == 1) Asm<Real, D>::rows1(z + I0, x + I0, y);
    if constexpr (R == 2) Asm<Real, D>::rows2(z + I0, x + I0, y);
    if constexpr (R == 3) Asm<Real, D>::rows3(z + I0, x + I0, y);
}
template <int D, typename Real>
__device__ __forceinline__ void zero(Real* z) {
#pragma unroll
    for (int i = 0; i < D; ++i) z[i] = Real(0.0);
}
template <int D, typename Real>
__device__ __forceinline__ void copy(Real* z, const Real* x) {
#pragma unroll
    for (int i = 0; i < D; ++i) z[i] = x[i];
}
// (X v)_lane + add, X in ROW layout (lane i holds row i), v distributed (lane k holds v_k); also the
// row-wide sum  sum_k v_k x_k + add  when x is replicated (every lane gets the same value)
template <int D, typename Real>
__device__ __forceinline__ Real mvr(const Real* xr, Id<Real> v, Id<Real> add) {
    Real a0 = add, a1 = Real(0.0);
    Asm<Real, D>::mv(a0, a1, v, xr);
    return a0 + a1;
}
// add + sum_k v_k over the row (every lane gets it)
template <int D, typename Real>
__device__ __forceinline__ Real rowsum(Real v, Id<Real> add) {
    Real a0 = add, a1 = Real(0.0);
    Asm<Real, D>::rsum(a0, a1, v, Real(1.0));
    return a0 + a1;
}
// sum_i h_i X[i][lane]: (X^T h)_lane, = (X h)_lane for symmetric X
template <int D, typename Real>
__device__ __forceinline__ Real dot_h(const Real* x, const Real* h) {
    Real a0 = Real(0.0), a1 = Real(0.0);
#pragma unroll
    for (int i = 0; i + 1 < D; i += 2) { a0 = __builtin_fma(h[i], x[i], a0); a1 = __builtin_fma(h[i + 1], x[i + 1], a1); }
    if constexpr (D & 1) a0 = __builtin_fma(h[D - 1], x[D - 1], a0);
    return a0 + a1;
}
template <int D, typename Real>
__device__ __forceinline__ void rank1(Real* z, Id<Real> p, Id<Real> q) { Asm<Real, D>::rank1(z, p, q); }

// xt = X^T through the row's LDS patch.  Patch rows >= D are zero (cleared once, never written), so lanes >= D
// read zeros.
constexpr int kTrDimMax = 8;                // fp32: in-register transposes up to this d (Asm<float, D>::tr)
// INREG (fp32, d <= 8): d^2 DPP selects instead of the round trip through LDS.  Worth it only where many waves share a
// SIMD -- measured at d = 6, fp32, 2^20 steps: rc_smooth1 (6 waves) 0.180 -> 0.152 ms, but rc_reduce1 / rc_apply1
// (3-4 waves) 0.234 -> 0.276 / 0.350 -> 0.433 ms -- so only the smoother asks for it.
template <int D, bool INREG = false, typename Real>
__device__ __forceinline__ void transpose(const Real* x, Real* xt, Real* patch, int lane) {
    if constexpr (INREG && sizeof(Real) == 4 && D <= kTrDimMax) {
        zero<D>(xt);
        Asm<Real, D>::tr(xt, x);
        return;
    }
#pragma unroll
    for (int i = 0; i < D; ++i) patch[i * kLdT + lane] = x[i];
    sync();
#pragma unroll
    for (int i = 0; i < D; ++i) xt[i] = patch[lane * kLdT + i];
    sync();
}
template <int D, bool INREG = false, typename Real>
__device__ __forceinline__ void symmetrise(Real* x, Real* patch, int lane) {
    Real xt[D];
    transpose<D, INREG>(x, xt, patch, lane);
#pragma unroll
    for (int i = 0; i < D; ++i) x[i] = Real(0.5) * (x[i] + xt[i]);
}
// zero-initialised wave-private LDS area of `bytes` bytes (the WIDE slots: their last 16 bytes are the zeros that
// lanes >= D read)
__device__ __forceinline__ void lds_clear(char* p, int bytes) {
    for (int e = threadIdx.x * 16; e < bytes; e += 64 * 16) *reinterpret_cast<float4*>(p + e) = make_float4(0.f, 0.f, 0.f, 0.f);
}

template <typename Real>
__device__ __forceinline__ Real* patch_init(Real* tl, int row) {
    for (int e = threadIdx.x; e < 4 * kPatch; e += 64) tl[e] = Real(0.0);
    sync();
    return tl + row * kPatch;
}

template <int K>
__device__ __forceinline__ double bcast(double x) { return __builtin_amdgcn_update_dpp(x, x, 0x150 + K, 0xf, 0xf, true); }
template <int K>
__device__ __forceinline__ float bcast(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, x), __builtin_bit_cast(int, x), 0x150 + K, 0xf,
                                                                 0xf, true));
}

// maximum of a distributed vector over the row (every lane gets it)
template <int D, int K = 0>
__device__ __forceinline__ double mvr_max(double v) {
    const double t = bcast<K>(v);
    if constexpr (K + 1 < D) return fmax(t, mvr_max<D, K + 1>(v));
    else return t;
}

// 1 / x by v_rcp_f64 and two Newton steps (the IEEE division sequence is ~25 instructions, this is 5; pivots of
// a positive definite matrix need neither the denormal nor the overflow path)
__device__ __forceinline__ double rcp_nr(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ float rcp_nr(float x) {      // v_rcp_f32 is good to 1 ulp: one Newton step rounds it off
    float r = __builtin_amdgcn_rcpf(x);
    r = __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
    return r;
}

// Gauss-Jordan without pivoting (M symmetric positive definite): B <- M^-1 B.  Row operations in column layout:
// row_r -= M[r][c] * row_c / M[c][c], the factor M[r][c] being lane c's register r broadcast to the row.
// M is destroyed.
template <int D, int C>
struct GjStep {
    template <typename Real>
    static __device__ __forceinline__ void run(Real* M, Real* B) {
        const Real inv = rcp_nr(bcast<C>(M[C]));
        const Real mc = M[C] * inv, bv = B[C] * inv;
        if constexpr (D <= 8) {
            Gj<Real, D, C>::run(M, B, -mc, -bv);
        } else {
            Gj<Real, 8, C>::run(M, B, -mc, -bv);
            Gj<Real, D - 8, C>::run(M + 8, B + 8, -mc, -bv);
        }
        M[C] = mc;                       // the pivot row itself (whatever the block did to it is discarded)
        B[C] = bv;
        if constexpr (C + 1 < D) GjStep<D, C + 1>::run(M, B);
    }
};

// Per-lane addressing of one chain's records.  The four chains of a wave sit Lw steps apart, so every access is
// (wave-uniform base of row 0's step) + (one loop-invariant 32-bit byte offset per lane and layout) + (a
// compile-time offset per register, which rides in the instruction's immediate field): element (i, lane) for the
// column layout, (lane, i) for the row layout.
//
// FAST = every chain of the wave is inside the series for the whole loop (all waves but the first / last).  Its
// accesses are RAW BUFFER loads / stores through a descriptor that spans the four rows' records of the step: lanes
// >= D carry an out-of-range offset, so the hardware returns zeros to them and drops their stores -- no EXEC-masked
// branch around the memory instructions and no select behind a load.  That matters twice: the loads stay in flight
// until their first use, and -- straight-line code -- the compiler counts the outstanding memory operations exactly,
// so waiting for a load does not also wait for the younger stores of the step (a branch around a store makes the
// count unknown at the join and every later wait a full `s_waitcnt vmcnt(0)`: measured, that drained the stores'
// round trip into every time step).
// The prefetch of the next step's inputs has no dependence on anything in the step, so the machine scheduler would sink
// it to the end of the loop body (shortest live ranges) and expose the whole memory latency: fence it in.
#define PGPS_RC_PIN() __builtin_amdgcn_sched_barrier(0)

// Waves per SIMD rc_reduce1 / rc_apply1 are compiled for (register budget 512 / waves).  At d = 11 fp64 they need
// about 275 registers; compiled for two waves they spill 80..200 B per lane inside the step loop and lose 30 %.
// (Their VALU is busy 53..56 % of the time on one wave per SIMD, so a second wave would pay -- once ~25 registers of
// live state are gone; rc_smooth1 fits two waves and gains nothing, it is HBM-bound: profiles/r02_experiments.txt.)
#ifndef PGPS_RC_WAVES
#define PGPS_RC_WAVES 1
#endif
#ifndef PGPS_RC_WAVES_R
#define PGPS_RC_WAVES_R PGPS_RC_WAVES           // rc_reduce1 alone
#endif

constexpr unsigned kOob = 0x7ffff000u;         // + any immediate offset: beyond every descriptor's range, no 32-bit wrap

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void buf_ld(__amdgpu_buffer_rsrc_t r, unsigned off, double& x) {
    using U2 = __attribute__((ext_vector_type(2))) unsigned int;
    const U2 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0);
    x = __builtin_bit_cast(double, v);
}
__device__ __forceinline__ void buf_ld(__amdgpu_buffer_rsrc_t r, unsigned off, float& x) {
    x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0));
}
__device__ __forceinline__ void buf_st(__amdgpu_buffer_rsrc_t r, unsigned off, double x) {
    using U2 = __attribute__((ext_vector_type(2))) unsigned int;
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(U2, x), r, (int)off, 0, 0);
}
__device__ __forceinline__ void buf_st(__amdgpu_buffer_rsrc_t r, unsigned off, float x) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, x), r, (int)off, 0, 0);
}

// NV 1 KiB pieces straight from global memory into LDS (LDS-DMA: buffer_load_dwordx4 ... lds -- lane t of piece v
// lands at lds + v * 1024 + t * 16; out-of-range lanes land zeros), no staging registers.  The compiler does not count
// these loads: whoever reads the slot waits for them itself (dma_wait_all: s_waitcnt vmcnt(0) -- hipcc's own waits can
// only be stricter than it thinks, never laxer: it counts the younger operations it knows of, and there are at least
// those).  M0 is restored.
#define PGPS_DMA_NEXT "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
#define PGPS_DMA_LD(n) "buffer_load_dwordx4 %[v" #n "], %[rs], 0 offen lds\n\t"
template <int NV>
__device__ __forceinline__ void dma_pieces(__amdgpu_buffer_rsrc_t rs, const unsigned (&vo)[NV], unsigned lds) {
    static_assert(NV >= 1 && NV <= 8, "one to eight pieces per matrix");
    unsigned keep;
    constexpr const char* dummy = "";
    (void)dummy;
#define PGPS_DMA_HEAD "s_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[lds]\n\ts_nop 4\n\t"
#define PGPS_DMA_TAIL "s_mov_b32 m0, %[keep]"
#define PGPS_DMA_OUT [keep] "=&s"(keep)
#define PGPS_DMA_IN [lds] "s"(lds), [rs] "s"(rs)
    if constexpr (NV == 1) {
        asm volatile(PGPS_DMA_HEAD PGPS_DMA_LD(0) PGPS_DMA_TAIL : PGPS_DMA_OUT : PGPS_DMA_IN, [v0] "v"(vo[0]) : "memory");
    } else if constexpr (NV == 2) {
        asm volatile(PGPS_DMA_HEAD PGPS_DMA_LD(0) PGPS_DMA_NEXT PGPS_DMA_LD(1) PGPS_DMA_TAIL
                     : PGPS_DMA_OUT : PGPS_DMA_IN, [v0] "v"(vo[0]), [v1] "v"(vo[1]) : "memory");
    } else if constexpr (NV == 3) {
        asm volatile(PGPS_DMA_HEAD PGPS_DMA_LD(0) PGPS_DMA_NEXT PGPS_DMA_LD(1) PGPS_DMA_NEXT PGPS_DMA_LD(2) PGPS_DMA_TAIL
                     : PGPS_DMA_OUT : PGPS_DMA_IN, [v0] "v"(vo[0]), [v1] "v"(vo[1]), [v2] "v"(vo[2]) : "memory");
    } else if constexpr (NV == 4) {
        asm volatile(PGPS_DMA_HEAD PGPS_DMA_LD(0) PGPS_DMA_NEXT PGPS_DMA_LD(1) PGPS_DMA_NEXT PGPS_DMA_LD(2) PGPS_DMA_NEXT
                     PGPS_DMA_LD(3) PGPS_DMA_TAIL
                     : PGPS_DMA_OUT : PGPS_DMA_IN, [v0] "v"(vo[0]), [v1] "v"(vo[1]), [v2] "v"(vo[2]), [v3] "v"(vo[3]) : "memory");
    } else if constexpr (NV == 5) {
        asm volatile(PGPS_DMA_HEAD PGPS_DMA_LD(0) PGPS_DMA_NEXT PGPS_DMA_LD(1) PGPS_DMA_NEXT PGPS_DMA_LD(2) PGPS_DMA_NEXT
                     PGPS_DMA_LD(3) PGPS_DMA_NEXT PGPS_DMA_LD(4) PGPS_DMA_TAIL
                     : PGPS_DMA_OUT : PGPS_DMA_IN, [v0] "v"(vo[0]), [v1] "v"(vo[1]), [v2] "v"(vo[2]), [v3] "v"(vo[3]), [v4] "v"(vo[4])
                     : "memory");
    } else if constexpr (NV == 6) {
        asm volatile(PGPS_DMA_HEAD PGPS_DMA_LD(0) PGPS_DMA_NEXT PGPS_DMA_LD(1) PGPS_DMA_NEXT PGPS_DMA_LD(2) PGPS_DMA_NEXT
                     PGPS_DMA_LD(3) PGPS_DMA_NEXT PGPS_DMA_LD(4) PGPS_DMA_NEXT PGPS_DMA_LD(5) PGPS_DMA_TAIL
                     : PGPS_DMA_OUT : PGPS_DMA_IN, [v0] "v"(vo[0]), [v1] "v"(vo[1]), [v2] "v"(vo[2]), [v3] "v"(vo[3]), [v4] "v"(vo[4]),
                       [v5] "v"(vo[5]) : "memory");
    } else if constexpr (NV == 7) {
        asm volatile(PGPS_DMA_HEAD PGPS_DMA_LD(0) PGPS_DMA_NEXT PGPS_DMA_LD(1) PGPS_DMA_NEXT PGPS_DMA_LD(2) PGPS_DMA_NEXT
                     PGPS_DMA_LD(3) PGPS_DMA_NEXT PGPS_DMA_LD(4) PGPS_DMA_NEXT PGPS_DMA_LD(5) PGPS_DMA_NEXT PGPS_DMA_LD(6) PGPS_DMA_TAIL
                     : PGPS_DMA_OUT : PGPS_DMA_IN, [v0] "v"(vo[0]), [v1] "v"(vo[1]), [v2] "v"(vo[2]), [v3] "v"(vo[3]), [v4] "v"(vo[4]),
                       [v5] "v"(vo[5]), [v6] "v"(vo[6]) : "memory");
    } else {
        asm volatile(PGPS_DMA_HEAD PGPS_DMA_LD(0) PGPS_DMA_NEXT PGPS_DMA_LD(1) PGPS_DMA_NEXT PGPS_DMA_LD(2) PGPS_DMA_NEXT
                     PGPS_DMA_LD(3) PGPS_DMA_NEXT PGPS_DMA_LD(4) PGPS_DMA_NEXT PGPS_DMA_LD(5) PGPS_DMA_NEXT PGPS_DMA_LD(6) PGPS_DMA_NEXT
                     PGPS_DMA_LD(7) PGPS_DMA_TAIL
                     : PGPS_DMA_OUT : PGPS_DMA_IN, [v0] "v"(vo[0]), [v1] "v"(vo[1]), [v2] "v"(vo[2]), [v3] "v"(vo[3]), [v4] "v"(vo[4]),
                       [v5] "v"(vo[5]), [v6] "v"(vo[6]), [v7] "v"(vo[7]) : "memory");
    }
#undef PGPS_DMA_HEAD
#undef PGPS_DMA_TAIL
#undef PGPS_DMA_OUT
#undef PGPS_DMA_IN
}
#undef PGPS_DMA_NEXT
#undef PGPS_DMA_LD
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// Level-1 kernels fetch their per-step inputs by LDS-DMA (two slots per matrix, alternating) instead of through NVW x 4
// staging registers per matrix: at d = 11 fp64 that is what takes rc_reduce1 / rc_apply1 from 274 / 283 registers to
// under 256, i.e. to two waves per SIMD.
#ifndef PGPS_RC_DMA
#define PGPS_RC_DMA 1
#endif
#ifndef PGPS_RC_EDGE_FAST
#define PGPS_RC_EDGE_FAST 1                  // the first and the last workgroup of a series on the FAST road (rc_reduce1)
#endif

template <int D, typename Real>
struct Io {
    static constexpr unsigned W = sizeof(Real);
    unsigned oc, orw, ov, ro8;          // byte offsets of (0, lane), (lane, 0), vector element `lane`; of the row's step
    unsigned foc, forw, fov;            // the same for the buffer accesses: out of range for lanes >= D
    unsigned span_m, span_v;            // bytes from row 0's record to the end of row 3's: matrices, vectors
    unsigned ros, srl, span_s;          // symmetric records (below): byte offset of the row's step, sym_row(lane), span
    bool lv;
    int lane;
    __device__ __forceinline__ void init(int lane_, int row, int Lw) {
        lane = lane_;
        lv = lane < D;
        ros = (unsigned)row * (unsigned)Lw * (unsigned)RECS;
        srl = (unsigned)(lane * D - lane * (lane - 1) / 2 - lane);
        span_s = (3u * (unsigned)Lw + 1u) * (unsigned)RECS;
        const unsigned ro = (unsigned)row * (unsigned)Lw * (unsigned)(D * D);
        ro8 = ro * W;
        oc = lv ? (ro + (unsigned)lane) * W : 0u;
        orw = lv ? (ro + (unsigned)(lane * D)) * W : 0u;
        ov = lv ? ((unsigned)row * (unsigned)Lw * (unsigned)D + (unsigned)lane) * W : 0u;
        foc = lv ? oc : kOob; forw = lv ? orw : kOob; fov = lv ? ov : kOob;
        span_m = (3u * (unsigned)Lw + 1u) * (unsigned)(D * D) * W;
        span_v = (3u * (unsigned)Lw + 1u) * (unsigned)D * W;
    }
    template <bool ROWL>
    static constexpr int step_bytes() { return ROWL ? (int)W : (int)W * D; }      // from register i to register i + 1
    // fast path: X <- the matrix (zeros in lanes >= D)
    // (EXEC-masked global loads: range-checked buffer loads make every lane of the wave do address work, the
    // out-of-range ones included -- measured on c5's rc_reduce1: 0.69 -> 0.91 ms)
    template <bool ROWL>
    __device__ __forceinline__ void mat_fast(const Real* base, Real* X) const {
        if (lv) {
            const char* p = reinterpret_cast<const char*>(base) + (ROWL ? orw : oc);
#pragma unroll
            for (int i = 0; i < D; ++i) X[i] = *reinterpret_cast<const Real*>(p + i * step_bytes<ROWL>());
        }
    }
    // general path: rows that are not `real` get dg * I.  One divergent if / else around the whole block, no
    // selects behind the loads (a select would wait for them on the spot).
    template <bool ROWL>
    __device__ __forceinline__ void mat_slow(const Real* base, bool real, Real dg, Real* X) const {
        if (lv && real) {
            const char* p = reinterpret_cast<const char*>(base) + (ROWL ? orw : oc);
#pragma unroll
            for (int i = 0; i < D; ++i) X[i] = *reinterpret_cast<const Real*>(p + i * step_bytes<ROWL>());
        } else {
#pragma unroll
            for (int i = 0; i < D; ++i) X[i] = (i == lane && lv) ? dg : Real(0.0);
        }
    }
    // fast path: x <- the vector element (zero in lanes >= D)
    __device__ __forceinline__ void vec_fast(const Real* base, Real& x) const {
        if (lv) x = *reinterpret_cast<const Real*>(reinterpret_cast<const char*>(base) + ov);
    }
    __device__ __forceinline__ Real vec(const Real* base, bool real) const {
        Real v = Real(0.0);
        if (lv && real) v = *reinterpret_cast<const Real*>(reinterpret_cast<const char*>(base) + ov);
        return v;
    }
    // stores: FAST through the descriptor (every lane executes them, lanes >= D are out of range), otherwise predicated
    template <bool FAST>
    __device__ __forceinline__ void st_mat(Real* base, bool pred, const Real* X) const {
        if constexpr (FAST) {
            const __amdgpu_buffer_rsrc_t r = make_rsrc(base, span_m);
#pragma unroll
            for (int i = 0; i < D; ++i) buf_st(r, foc + (unsigned)(i * (int)W * D), X[i]);
        } else if (lv && pred) {
            char* p = reinterpret_cast<char*>(base) + oc;
#pragma unroll
            for (int i = 0; i < D; ++i) *reinterpret_cast<Real*>(p + i * (int)W * D) = X[i];
        }
    }
    template <bool FAST>
    __device__ __forceinline__ void st_vec(Real* base, bool pred, Real x) const {
        if constexpr (FAST) buf_st(make_rsrc(base, span_v), fov, x);
        else if (lv && pred) *reinterpret_cast<Real*>(reinterpret_cast<char*>(base) + ov) = x;
    }

    // ---- symmetric records: the smoother's L waits in scratch between rc_apply1 / rc_selem1 and rc_smooth1, and the
    // smoother pass is HBM-bound on (E, L, g) in, (sm, sP) out -- so L travels as its upper triangle, row-major
    // (element (i, j), j >= i, at sym_row(i) + j), NS = d (d + 1) / 2 scalars in records RECS bytes apart (padded to
    // whole 16-byte pieces).  Lane j writes the part of its column above the diagonal; it reads the part below as
    // row j.  d = 11 fp64: 528 instead of 968 bytes per step and direction (c5: rc_smooth1 0.65 -> 0.57 ms).
    static constexpr int NS = D * (D + 1) / 2;
    static constexpr int RECS = (NS * (int)sizeof(Real) + 15) / 16 * 16;
    static constexpr int sym_row(int i) { return i * D - i * (i - 1) / 2 - i; }
    static __device__ __forceinline__ const Real* sym_at(const Real* base, long k) {
        return reinterpret_cast<const Real*>(reinterpret_cast<const char*>(base) + k * RECS);
    }
    static __device__ __forceinline__ Real* sym_at(Real* base, long k) {
        return reinterpret_cast<Real*>(reinterpret_cast<char*>(base) + k * RECS);
    }
    // scalar offset of element (i, lane) of the lane's column inside a record
    __device__ __forceinline__ unsigned sym_off(int i) const {
        return i <= lane ? (unsigned)(sym_row(i) + lane) : srl + (unsigned)i;
    }
    template <bool FAST>
    __device__ __forceinline__ void st_sym(Real* rec, bool pred, const Real* X) const {
        if constexpr (FAST) {
            const __amdgpu_buffer_rsrc_t r = make_rsrc(rec, span_s);
#pragma unroll
            for (int i = 0; i < D; ++i)
                buf_st(r, (lv && lane >= i) ? ros + (unsigned)(sym_row(i) + lane) * W : kOob, X[i]);
        } else if (lv && pred) {
            char* p = reinterpret_cast<char*>(rec) + ros + (unsigned)lane * W;
#pragma unroll
            for (int i = 0; i < D; ++i)
                if (lane >= i) *reinterpret_cast<Real*>(p + sym_row(i) * (int)W) = X[i];
        }
    }
    // X <- the record's matrix, column layout; rows that are not `real` get zeros
    __device__ __forceinline__ void ld_sym(const Real* rec, bool real, Real* X) const {
        if (lv && real) {
            const char* p = reinterpret_cast<const char*>(rec) + ros;
#pragma unroll
            for (int i = 0; i < D; ++i) X[i] = *reinterpret_cast<const Real*>(p + sym_off(i) * W);
        } else {
#pragma unroll
            for (int i = 0; i < D; ++i) X[i] = Real(0.0);
        }
    }

    // ---- WIDE path of the FAST bodies: whole records as 16-byte pieces, staged through wave-private LDS ------------
    // The per-register accesses above move 4 rows x D scalars (96 .. 512 bytes) per memory instruction, and the level-1
    // kernels issue 45 .. 70 of them per step: measured, their time is the memory pipeline's per-instruction cost
    // (about 12 cycles + 0.11 cycles per byte per CU), not HBM bandwidth and not latency.  So the four rows' records
    // of a step travel as 16-byte pieces -- piece q of the step belongs to row q / NPC and is handled by lane q % 64 of
    // instruction q / 64: a full kilobyte per instruction -- into registers, from there into an LDS slot (row r at
    // r * RECPAD), and the lanes read both layouts they need out of LDS (the second global read of F disappears too).
    // Results take the same road back.  The loads of step s + 1 are issued right after step s's pieces have been
    // committed to LDS, a whole step ahead of their use, at the price of NVW x 4 registers per matrix.
    static constexpr int REC = D * D * (int)W;              // bytes of one matrix record
    static constexpr int RECPAD = (REC + 15) / 16 * 16;
    static constexpr int NPC = RECPAD / 16;                 // pieces per row (the last one may reach 8 bytes beyond)
    static constexpr int NPF = REC / 16;                    // whole pieces per row
    static constexpr int TAILB = REC % 16;                  // bytes of the tail piece of a row: 8 (odd d, fp64), 4 (odd d, fp32)
    static constexpr bool TAIL = TAILB != 0;
    static constexpr int NVW = (4 * NPC + 63) / 64;         // wide instructions per matrix and step
    static constexpr int SLOT = NVW * 1024;                 // bytes of one LDS slot (whole instructions: no masking)
    // rc_apply1: five slots (two inputs, three results) when four such waves plus their transpose patches fit a CU's LDS
#ifdef PGPS_RC_NO_BATCH
    static constexpr bool kBatchOut = false;
#else
    static constexpr bool kBatchOut = 5 * (SLOT + 16) + 4 * 16 * 17 * (int)W <= 40 * 1024;
#endif
    using V4 = __attribute__((ext_vector_type(4))) unsigned int;
    unsigned wg[NVW], wgs[NVW], wtail;  // global byte offsets of this lane's pieces: loads / stores (out of range when
                                        // there is no such piece; stores: also for the tail piece) / the 8-byte tail
    unsigned lc, lr, ltail;             // LDS byte offsets of (0, lane) column layout, (lane, 0) row layout; tail
    // symmetric records on the same road: NPS pieces per row and step, no tails (RECS is whole pieces)
    static constexpr int NPS = RECS / 16;
    static constexpr int NVS = (4 * NPS + 63) / 64;
    unsigned wsym[NVS];                 // global byte offsets of this lane's pieces (out of range when there is none)
    unsigned lsrow, zoff;               // LDS byte offset of the row's record in a slot; of the slot's zeros
    int tid;
    __device__ __forceinline__ void init_wide(int row, int Lw, unsigned zero_off) {
        tid = threadIdx.x & 63;
        const unsigned pitch = (unsigned)Lw * (unsigned)REC;
#pragma unroll
        for (int v = 0; v < NVW; ++v) {
            const int q = v * 64 + tid, r = q / NPC, idx = q % NPC;
            const bool ok = q < 4 * NPC;
            wg[v] = ok ? (unsigned)r * pitch + (unsigned)idx * 16u : kOob;
            wgs[v] = (ok && idx < NPF) ? (unsigned)r * pitch + (unsigned)idx * 16u : kOob;
        }
#pragma unroll
        for (int v = 0; v < NVS; ++v) {
            const int q = v * 64 + tid, r = q / NPS, idx = q % NPS;
            wsym[v] = q < 4 * NPS ? (unsigned)r * (unsigned)Lw * (unsigned)RECS + (unsigned)idx * 16u : kOob;
        }
        lsrow = (unsigned)row * (unsigned)RECS;
        zoff = zero_off;
        wtail = (TAIL && tid < 4) ? (unsigned)tid * pitch + (unsigned)NPF * 16u : kOob;
        ltail = (unsigned)(tid & 3) * (unsigned)RECPAD + (unsigned)NPF * 16u;
        // lanes >= D read the zeros at zero_off (relative to the slot) in both layouts: no select behind the LDS read
        lc = lv ? (unsigned)row * (unsigned)RECPAD + (unsigned)lane * W : zero_off;
        lr = lv ? (unsigned)row * (unsigned)RECPAD + (unsigned)(lane * D) * W : zero_off;
    }
    __device__ __forceinline__ void wide_load(const Real* base, V4* r) const {
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(base, span_m + 16u);    // (+16: the last piece of a TAIL record)
#pragma unroll
        for (int v = 0; v < NVW; ++v) r[v] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)wg[v], 0, 0);
    }
    // (descriptor cut `left` records after base's: see wide_dma_cut)
    __device__ __forceinline__ void wide_load_cut(const Real* base, V4* r, long left) const {
        const long bytes = left * (long)REC;
        const unsigned lim = bytes <= 0 ? 0u : (bytes < (long)(span_m + 16u) ? (unsigned)bytes : span_m + 16u);
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(base, lim);
#pragma unroll
        for (int v = 0; v < NVW; ++v) r[v] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)wg[v], 0, 0);
    }
    __device__ __forceinline__ void wide_dma(const Real* base, const char* slot) const {
        dma_pieces<NVW>(make_rsrc(base, span_m + 16u), wg, (unsigned)(size_t)slot);
    }
    // the same with the descriptor cut `left` records after base's (the end of the array): what lies beyond lands as zeros
    // (the range check is per dword -- tools/micro/buf_range.hip -- so a record that is not whole pieces is still exact)
    __device__ __forceinline__ void wide_dma_cut(const Real* base, const char* slot, long left) const {
        const long bytes = left * (long)REC;
        const unsigned lim = bytes <= 0 ? 0u : (bytes < (long)(span_m + 16u) ? (unsigned)bytes : span_m + 16u);
        dma_pieces<NVW>(make_rsrc(base, lim), wg, (unsigned)(size_t)slot);
    }
    __device__ __forceinline__ void wide_commit(char* slot, const V4* r) const {
#pragma unroll
        for (int v = 0; v < NVW; ++v) *reinterpret_cast<V4*>(slot + (v * 64 + tid) * 16) = r[v];
    }
    // X <- the slot's matrix in column (ROWL = false) or row layout; lanes >= D get zeros.  `stride0` = 0 for them.
    template <bool ROWL>
    __device__ __forceinline__ void lds_get(const char* slot, Real* X) const {
        const char* p = slot + (ROWL ? lr : lc);
        const int st = lv ? step_bytes<ROWL>() : 0;
#pragma unroll
        for (int i = 0; i < D; ++i) X[i] = *reinterpret_cast<const Real*>(p + i * st);
    }
    // results: the slot <- X (column layout) ... (several slots may be filled before one sync() and their drains)
    __device__ __forceinline__ void lds_put(char* slot, const Real* X) const {
        if (lv) {
            char* p = slot + lc;
#pragma unroll
            for (int i = 0; i < D; ++i) *reinterpret_cast<Real*>(p + i * (int)W * D) = X[i];
        }
    }
    // ... and, after a sync(), the slot's four records out as 16-byte pieces (+ the 8-byte tails)
    __device__ __forceinline__ void wide_drain(Real* base, const char* slot) const {
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(base, span_m);
#pragma unroll
        for (int v = 0; v < NVW; ++v) {
            const V4 x = *reinterpret_cast<const V4*>(slot + (v * 64 + tid) * 16);
            __builtin_amdgcn_raw_buffer_store_b128(x, rs, (int)wgs[v], 0, 0);
        }
        static_assert(TAILB == 0 || TAILB == 4 || TAILB == 8, "a record is a whole number of scalars, d^2 of them");
        if constexpr (TAILB == 8) {
            using U2 = __attribute__((ext_vector_type(2))) unsigned int;
            const U2 t = *reinterpret_cast<const U2*>(slot + ltail);
            __builtin_amdgcn_raw_buffer_store_b64(t, rs, (int)wtail, 0, 0);
        } else if constexpr (TAILB == 4) {
            const unsigned int t = *reinterpret_cast<const unsigned int*>(slot + ltail);
            __builtin_amdgcn_raw_buffer_store_b32(t, rs, (int)wtail, 0, 0);
        }
    }
    __device__ __forceinline__ void wide_store(Real* base, char* slot, const Real* X) const {
        sync();                                             // earlier readers of the slot are done
        lds_put(slot, X);
        sync();
        wide_drain(base, slot);
    }
    // the same for symmetric records (slots are sized for full records: these fit)
    __device__ __forceinline__ void wide_load_sym(const Real* rec, V4* r) const {
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(rec, span_s);
#pragma unroll
        for (int v = 0; v < NVS; ++v) r[v] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)wsym[v], 0, 0);
    }
    __device__ __forceinline__ void wide_commit_sym(char* slot, const V4* r) const {
#pragma unroll
        for (int v = 0; v < NVS; ++v) *reinterpret_cast<V4*>(slot + (v * 64 + tid) * 16) = r[v];
    }
    __device__ __forceinline__ void lds_get_sym(const char* slot, Real* X) const {
        const char* p = slot + (lv ? lsrow : zoff);
#pragma unroll
        for (int i = 0; i < D; ++i) X[i] = *reinterpret_cast<const Real*>(p + (lv ? sym_off(i) * W : 0u));
    }
    __device__ __forceinline__ void lds_put_sym(char* slot, const Real* X) const {
        if (lv) {
            char* p = slot + lsrow + (unsigned)lane * W;
#pragma unroll
            for (int i = 0; i < D; ++i)
                if (lane >= i) *reinterpret_cast<Real*>(p + sym_row(i) * (int)W) = X[i];
        }
    }
    __device__ __forceinline__ void wide_drain_sym(Real* rec, const char* slot) const {
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(rec, span_s);
#pragma unroll
        for (int v = 0; v < NVS; ++v) {
            const V4 x = *reinterpret_cast<const V4*>(slot + (v * 64 + tid) * 16);
            __builtin_amdgcn_raw_buffer_store_b128(x, rs, (int)wsym[v], 0, 0);
        }
    }
    __device__ __forceinline__ void wide_store_sym(Real* rec, char* slot, const Real* X) const {
        sync();
        lds_put_sym(slot, X);
        sync();
        wide_drain_sym(rec, slot);
    }
};

// Batched evaluation (pgps_lti_ll_batch_*): blockIdx.y selects one of `batch` models over the same series; each
// model has its own slice of the discretised arrays, of the scan scratch and of the model table
// [F | Pinf | H | R] (stride bs_model).  The kernel bodies below never see the difference.
template <typename Real>
__device__ __forceinline__ RcArgsT<Real> model_view(const RcArgsT<Real>& a) {
    RcArgsT<Real> b = a;
    if (a.Rs) {
        const long mb = blockIdx.y;
        b.Fs += mb * a.bs_F;
        if (b.Qs) b.Qs += mb * a.bs_F;
        b.P0 += mb * a.bs_model; b.H += mb * a.bs_model;
        b.R = a.Rs[mb * a.bs_model];
        b.agg1 += mb * a.bs_agg;
        b.pre += mb * a.bs_agg;
        b.llpart += mb * a.nchunk;
    }
    return b;
}

// The level-1 kernels come as two bodies behind one launch: FAST for waves whose four chains lie entirely inside
// the series and do not contain its first step (no per-row predicates, no selects behind loads -- a select would
// put an s_waitcnt vmcnt(0) right after the loads and expose the whole memory latency every step), and the general
// body for the first and the last wave(s).  Two bodies, two loops: no joins inside the hot loop.

// ====================================================================================================
// level 1: reduce -- filt_extend over the chunk (pgps_math.h filt_extend, parallel.py:46-72,100-118)
// ====================================================================================================
// FAST: the four chains are complete and nothing is predicated; EDGE (with FAST): the workgroup holds the series' first
// step (F = I, Q = 0 there) and / or its last one (the fetch a step ahead runs off the end of the arrays: descriptor cut,
// index clamped) -- the same road, a handful of selects.  The general body is for ragged ends only: it is slower, and in a
// one-round grid its wave is what the kernel waits for (c5: 563 against 511 us with the edge waves switched off).
// IMPQ (the process noise is implicit, below) is a compile-time flavour: with the flag read at run time both Q and Pinf stay
// live through the step loop -- 22 registers at d = 11 fp64, exactly what kept the kernel from two waves per SIMD (268
// registers; either flavour alone: two waves, no scratch).
template <typename Real, int D, bool FAST, bool EDGE, bool IMPQ>
__device__ __forceinline__ void reduce1_body_q(const RcArgsT<Real>& a, Real* patch, char* wslots, int lane, int row) {
    static_assert(FAST || !EDGE, "EDGE is a flavour of the FAST body");
    constexpr int dd = D * D;
    const long kw = (long)blockIdx.x * 4 * a.Lw;        // row 0's first step
    const long c = (long)blockIdx.x * 4 + row;
    const long k0 = c * a.Lw, k1 = min(a.N, k0 + a.Lw);
    Io<D, Real> io;
    io.init(lane, row, a.Lw);
    const bool lv = io.lv;
    Real h[D];
#pragma unroll
    for (int i = 0; i < D; ++i) h[i] = a.H[i];
    Real A[D], C[D], J[D], b = Real(0.0), eta = Real(0.0);
#pragma unroll
    for (int i = 0; i < D; ++i) {
        // chunk 0 starts from (0, 0, P0, 0, 0) and takes step 0 with F = I, Q = 0: that is filt_first
        const bool head = (c == 0 && a.seg_first);
        A[i] = (!head && i == lane) ? Real(1.0) : Real(0.0);
        C[i] = (head && lv) ? Real(0.5) * (a.P0[i * D + lane] + a.P0[lane * D + i]) : Real(0.0);
        J[i] = Real(0.0);
    }
    Real Fc[D], Fr[D], Q[D], y;
    zero<D>(Fc); zero<D>(Fr); zero<D>(Q);
    // Implicit process noise (general-LTI log-likelihood calls): Q_k = Pinf - F_k Pinf F_k^T is never formed --
    // F C F^T + Q = F (C - Pinf) F^T + Pinf -- so the (N, d, d) array Qs does not exist; F = I gives Q = 0 by itself.
    constexpr bool impq = IMPQ;                  // (a.implicit_q: bit 0: Qs absent; bit 1: Qs there, but P0 known stationary -- skip reading it)
    Real Pinf[D];
#pragma unroll
    for (int i = 0; i < D; ++i) Pinf[i] = (impq && lv) ? Real(0.5) * (a.P0[i * D + lane] + a.P0[lane * D + i]) : Real(0.0);
    // inputs of this row's step kw + row Lw + s; steps outside the chunk and step 0 of the series run as F = I, Q = 0
    auto load = [&](int s) {                    // the general body's per-register loads
        const long ku = kw + s, k = k0 + s;
        const long kc = ku < a.N ? ku : a.N - 1;
        const bool real = k < k1 && !(k == 0 && a.seg_first);
        io.template mat_slow<false>(a.Fs + kc * dd, real, Real(1.0), Fc);
        io.template mat_slow<true>(a.Fs + kc * dd, real, Real(1.0), Fr);
        if (!impq) io.template mat_slow<false>(a.Qs + kc * dd, real, Real(0.0), Q);
        y = __builtin_nan("");
        if (k < k1) y = a.ys[k];
    };
    // FAST: whole records as 16-byte pieces through LDS (Io, WIDE path), requested a whole step ahead
    using IOT = Io<D, Real>;
    constexpr bool DMA = PGPS_RC_DMA != 0;
    typename IOT::V4 pF[DMA ? 1 : IOT::NVW], pQ[DMA ? 1 : IOT::NVW];
    Real ny = Real(0.0);
    // slots: [F | Q].  DMA: the next step's pieces are requested into the same slots as soon as this step's have been
    // read out of them (lgkmcnt(0): the reads are in registers) -- the whole step's arithmetic is still ahead of them
    constexpr int SL = IOT::SLOT + 16;
    auto prefetch = [&](int s) {
        if constexpr (DMA && EDGE) {
            io.wide_dma_cut(a.Fs + (kw + s) * dd, wslots, a.N - (kw + s));
            if (!impq) io.wide_dma_cut(a.Qs + (kw + s) * dd, wslots + SL, a.N - (kw + s));
        } else if constexpr (DMA) {
            io.wide_dma(a.Fs + (kw + s) * dd, wslots);
            if (!impq) io.wide_dma(a.Qs + (kw + s) * dd, wslots + SL);
        } else {
            io.wide_load(a.Fs + (kw + s) * dd, pF);
            if (!impq) io.wide_load(a.Qs + (kw + s) * dd, pQ);
        }
        ny = a.ys[EDGE ? min(k0 + s, a.N - 1) : k0 + s];
    };
    auto take = [&](int s) {                    // step s's pieces into LDS, step s + 1's on their way, both layouts out
        char* slotF = wslots;
        char* slotQ = slotF + SL;
        if constexpr (DMA) {
            dma_wait_all();                     // step s's pieces (requested a step ago) have landed
            y = ny;
            sync();
            io.template lds_get<false>(slotF, Fc);
            io.template lds_get<true>(slotF, Fr);
            if (!impq) io.template lds_get<false>(slotQ, Q);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            PGPS_RC_PIN();
            prefetch(s + 1);                    // (one step beyond the chunk at the end: inside the series for a FAST wave)
            PGPS_RC_PIN();
            if constexpr (EDGE) {
                if (s == 0 && k0 == 0 && a.seg_first) {             // the series' first step: F = I, Q = 0 (filt_first)
#pragma unroll
                    for (int i = 0; i < D; ++i) {
                        Fc[i] = (i == lane && lv) ? Real(1.0) : Real(0.0);
                        Fr[i] = Fc[i];
                        Q[i] = Real(0.0);
                    }
                }
            }
        } else {
            sync();
            io.wide_commit(slotF, pF);
            if (!impq) io.wide_commit(slotQ, pQ);
            y = ny;
            sync();
            PGPS_RC_PIN();
            prefetch(s + 1);
            PGPS_RC_PIN();
            io.template lds_get<false>(slotF, Fc);
            io.template lds_get<true>(slotF, Fr);
            if (!impq) io.template lds_get<false>(slotQ, Q);
        }
    };
    if constexpr (FAST) { io.init_wide(row, a.Lw, (unsigned)IOT::SLOT); prefetch(0); }
    else load(0);
    for (int s = 0; s < a.Lw; ++s) {
        if constexpr (FAST) take(s);
        Real Ap[D], FC[D], Cp[D];
        zero<D>(Ap); mm<D>(Ap, Fc, A);
        if (impq) {
            Real Cm[D];
#pragma unroll
            for (int i = 0; i < D; ++i) Cm[i] = C[i] - Pinf[i];
            zero<D>(FC); mm<D>(FC, Fc, Cm);
            copy<D>(Cp, Pinf);
        } else {
            zero<D>(FC); mm<D>(FC, Fc, C);
            copy<D>(Cp, Q);
        }
        mm<D>(Cp, FC, Fr);
        const Real bp = mvr<D>(Fr, b, Real(0.0));
        const Real yk = y;
        if constexpr (!FAST) {
            if (s + 1 < a.Lw) load(s + 1);      // next step's inputs: their registers are free from here on
        }
        symmetrise<D>(Cp, patch, lane);
        const Real u = dot_h<D>(Cp, h), v = dot_h<D>(Ap, h);
        const Real S = mvr<D>(h, u, a.R), hb = mvr<D>(h, bp, Real(0.0));
        const bool obs = !(yk != yk);
        const Real inv = obs ? Real(1.0) / S : Real(0.0);
        const Real res = obs ? yk - hb : Real(0.0);
        copy<D>(A, Ap); rank1<D>(A, u, -v * inv);
        copy<D>(C, Cp); rank1<D>(C, u, -u * inv);
        rank1<D>(J, v, v * inv);
        b = bp + u * (inv * res);
        eta += v * (res * inv);
    }
    if (c < a.nchunk && lv) {
        Real* rec = a.agg1 + c * nfilt(D);
#pragma unroll
        for (int i = 0; i < D; ++i) { rec[i * D + lane] = A[i]; rec[dd + i * D + lane] = C[i]; rec[2 * dd + i * D + lane] = J[i]; }
        rec[3 * dd + lane] = b;
        rec[3 * dd + D + lane] = eta;
    }
}

template <typename Real, int D, bool FAST, bool EDGE = false>
__device__ __forceinline__ void reduce1_body(const RcArgsT<Real>& a, Real* patch, char* wslots, int lane, int row) {
    if (a.implicit_q != 0) reduce1_body_q<Real, D, FAST, EDGE, true>(a, patch, wslots, lane, row);
    else reduce1_body_q<Real, D, FAST, EDGE, false>(a, patch, wslots, lane, row);
}

template <typename Real, int D>
__global__ __launch_bounds__(64, PGPS_RC_WAVES_R) void rc_reduce1(const RcArgsT<Real> a0) {
    __shared__ Real tl[4 * kPatch];
    const int lane = threadIdx.x & 15, row = threadIdx.x >> 4;
    __shared__ __attribute__((aligned(16))) char wslots[2 * (Io<D, Real>::SLOT + 16)];
    lds_clear(wslots, (int)sizeof(wslots));
    Real* patch = patch_init(tl, row);
    const RcArgsT<Real> a = model_view(a0);
#ifdef PGPS_RC_FAST_ONLY              // (diagnostic: the FAST body's registers alone)
    reduce1_body<Real, D, true>(a, patch, wslots, lane, row);
#else
    // (EDGE needs the LDS-DMA fetch -- its descriptor is what gets cut -- and whole chains: ragged ends take the general body)
    const bool full = ((long)blockIdx.x + 1) * 4 * a.Lw <= a.N;
    if (blockIdx.x >= 1 && blockIdx.x < a.wfast) reduce1_body<Real, D, true>(a, patch, wslots, lane, row);
#ifdef PGPS_RC_SKIP_EDGE               // (timing experiment: what the kernel costs without its edge waves; results are wrong)
    else return;
#else
    else if (PGPS_RC_DMA != 0 && PGPS_RC_EDGE_FAST != 0 && full) reduce1_body<Real, D, true, true>(a, patch, wslots, lane, row);
    else reduce1_body<Real, D, false>(a, patch, wslots, lane, row);
#endif
#endif
}

// ====================================================================================================
// level 1: apply -- Kalman pass, log-likelihood, smoothing elements and the chunk's smoothing total
// (kf_step / smth_element / smth_combine of pgps_math.h; parallel.py:135-151, 155-184)
// ====================================================================================================
// (FAST / EDGE as in reduce1_body; EDGE here also runs the series' first update from the prior itself and, in the step
// after the last chain of the series, takes Q = I for the F = 0 the cut descriptor delivers)
// DFORM (whole-series filter + smoother with every moment stored): the chain's smoothing TOTAL is accumulated in innovation form
// (pgps_math.h, smth_extend_u): folding step k-1's element into it takes E u (u = Pp H^T, the direction of step k's rank-one
// update), its image under the total's gain, and a rank-one update of L -- two matrix-vector products instead of the two
// matrix products E_a L and (E_a L) E_a^T and the symmetrisation behind them.  The per-step elements (E, g, L) are stored as
// before (rc_smooth1 applies them unchanged); rc_smooth1 enters a chain with the total's (g, L) PLUS the filtered moments of
// the chain's successor step (RcArgs::dform).  Segments keep the reference's form: their totals travel between ranks.
template <typename Real, int D, bool SMOOTH, bool FAST, bool IMPQS, bool STORE, bool EDGE = false, bool DFORM = false>
__device__ __forceinline__ void apply1_body(const RcArgsT<Real>& a, Real* patch, char* wslots, int lane, int row) {
    static_assert(FAST || !EDGE, "EDGE is a flavour of the FAST body");
    static_assert(!DFORM || (SMOOTH && STORE), "innovation-form totals: filter + smoother with the filtered moments stored");
    constexpr int dd = D * D;
    const long kw = (long)blockIdx.x * 4 * a.Lw;
    const long c = (long)blockIdx.x * 4 + row;
    const long k0 = c * a.Lw, k1 = min(a.N, k0 + a.Lw);
    Io<D, Real> io;
    io.init(lane, row, a.Lw);
    const bool lv = io.lv, cv = c < a.nchunk;
    Real h[D];
#pragma unroll
    for (int i = 0; i < D; ++i) h[i] = a.H[i];
    const Real hl = lv ? a.H[lane] : Real(0.0);     // this lane's entry of H: row sums of hl * v replace H-weighted ones
    // state entering the chunk: (b, C) of the inclusive prefix of the chunk before (A = 0 there); prior for chunk 0
    Real m, P[D];
    {
        // (a later segment of a sharded series enters its first chain with the carry-in of the ranks before it)
        const bool pr = cv && (c > 0 || !a.seg_first);
        const Real* rec = c > 0 ? a.pre + (cv ? c - 1 : 0) * nfilt(D) : (a.seg_first ? a.pre : a.carry);
        m = (pr && lv) ? rec[3 * dd + lane] : Real(0.0);
#pragma unroll
        for (int i = 0; i < D; ++i)
            P[i] = !lv ? Real(0.0) : pr ? rec[dd + i * D + lane] : (c == 0 ? Real(0.5) * (a.P0[i * D + lane] + a.P0[lane * D + i]) : Real(0.0));
    }
    Real Ec[D], L[D], g = Real(0.0);               // smoothing total of the steps seen so far
    if (SMOOTH) {
#pragma unroll
        for (int i = 0; i < D; ++i) { Ec[i] = (i == lane) ? Real(1.0) : Real(0.0); L[i] = Real(0.0); }
    }
    LogLik ll;
    Real Fc[D], Fr[D], Q[D], y;
    zero<D>(Fc); zero<D>(Fr); zero<D>(Q);
    // see rc_reduce1.  Filter only: a runtime flag; with the smoothing elements (which need F P itself: one more
    // product) a separate instantiation, so that the array path's kernel stays exactly as it was
    const bool impq = SMOOTH ? IMPQS : (a.implicit_q & 1);
    Real Pinf[D];
#pragma unroll
    for (int i = 0; i < D; ++i) Pinf[i] = (impq && lv) ? Real(0.5) * (a.P0[i * D + lane] + a.P0[lane * D + i]) : Real(0.0);
    // steps at or beyond N run with F = 0, Q = I: the element built from them is (0, m, P), i.e. the last
    // element of the series (parallel.py:155-156), and a total whose E is 0 absorbs whatever follows unchanged
    auto load = [&](int s) {                    // the general body's per-register loads
        const long ku = kw + s, k = k0 + s;
        {
            const long kc = ku < a.N ? ku : a.N - 1;
            // step N of a segment that is not the last: the first step of the next rank (halo), out of its record
            const bool hal = (k == a.N) && a.halo_F != nullptr;
            const bool real = k < a.N || hal;
            const char* bF = hal ? reinterpret_cast<const char*>(a.halo_F) - io.ro8 : reinterpret_cast<const char*>(a.Fs + kc * dd);
            const char* bQ = hal ? reinterpret_cast<const char*>(a.halo_Q) - io.ro8 : reinterpret_cast<const char*>(a.Qs + kc * dd);
            io.template mat_slow<false>(reinterpret_cast<const Real*>(bF), real, Real(0.0), Fc);
            io.template mat_slow<true>(reinterpret_cast<const Real*>(bF), real, Real(0.0), Fr);
            if (!impq) io.template mat_slow<false>(reinterpret_cast<const Real*>(bQ), real, Real(1.0), Q);
            y = __builtin_nan("");
            if (s < a.Lw && k < k1) y = a.ys[k];
            if (DFORM && s == a.Lw && k < a.N) y = a.ys[k];     // the step after the chunk lends its update to the last element
        }
    };
    // FAST: whole records as 16-byte pieces through LDS (Io, WIDE path), requested a whole step ahead; the results
    // leave the same way through a third slot
    using IOT = Io<D, Real>;
    typename IOT::V4 pF[IOT::NVW], pQ[IOT::NVW];
    Real ny = Real(0.0);
    char* slotF = wslots;
    char* slotQ = wslots + (IOT::SLOT + 16);
    // results: E, L of the step before, P of this step.  BATCH (the LDS of four waves per CU allows five slots): written
    // as they are made, drained together at the end of the step -- one LDS round trip for the three; otherwise one
    // slot, each result out on the spot
    constexpr bool BATCH = IOT::kBatchOut;
    char* slotE = wslots + 2 * (IOT::SLOT + 16);
    char* slotL = wslots + (BATCH ? 3 : 2) * (IOT::SLOT + 16);
    char* slotP = wslots + (BATCH ? 4 : 2) * (IOT::SLOT + 16);
    auto prefetch = [&](int s) {
        if constexpr (EDGE) {
            io.wide_load_cut(a.Fs + (kw + s) * dd, pF, a.N - (kw + s));
            if (!impq) io.wide_load_cut(a.Qs + (kw + s) * dd, pQ, a.N - (kw + s));
            ny = a.ys[min(k0 + s, a.N - 1)];
        } else {
            io.wide_load(a.Fs + (kw + s) * dd, pF);
            if (!impq) io.wide_load(a.Qs + (kw + s) * dd, pQ);
            ny = a.ys[k0 + s];
        }
    };
    auto take = [&](auto more_c, int s) {       // step s's pieces into LDS, step s + 1's on their way, both layouts out
        sync();
        io.wide_commit(slotF, pF);
        if (!impq) io.wide_commit(slotQ, pQ);
        y = ny;
        sync();
        if constexpr (decltype(more_c)::value) { PGPS_RC_PIN(); prefetch(s + 1); PGPS_RC_PIN(); }
        io.template lds_get<false>(slotF, Fc);
        io.template lds_get<true>(slotF, Fr);
        if (!impq) io.template lds_get<false>(slotQ, Q);
        if constexpr (EDGE && !decltype(more_c)::value) {
            // the step after the last chain of the series: F = 0 came out of the cut descriptor, Q = I by hand
            // (parallel.py:155-156: the last element is (0, m, P))
            if (k0 + s >= a.N) {
#pragma unroll
                for (int i = 0; i < D; ++i) Q[i] = (i == lane && lv) ? Real(1.0) : Real(0.0);
            }
        }
    };
    // a matrix result of step `ku`: FAST into its output slot (drained at the end of the step), otherwise predicated
    // per-register stores
    auto put = [&](char* slot, Real* arr, long ku, bool pred, const Real* X) {
        if constexpr (FAST && BATCH) io.lds_put(slot, X);
        else if constexpr (FAST) io.wide_store(arr + ku * dd, slot, X);
        else io.template st_mat<false>(arr + ku * dd, pred, X);
    };
    if constexpr (FAST) { io.init_wide(row, a.Lw, (unsigned)IOT::SLOT); prefetch(0); }
    else load(0);
    // One step, in three flavours so that the loop proper is straight-line code (see Io): FIRST has no smoothing element
    // to build (there is no step before the chunk's first), LAST (the step after the chunk, SMOOTH only) builds the
    // last step's element and does not filter.  Next step's inputs are requested as soon as this step's are dead when the
    // registers allow (EARLY), otherwise -- fp64 from d = 9 with the smoothing total on board -- just before this
    // step's own stores, so that waiting for them never waits for those stores.
    constexpr bool EARLY = !SMOOTH || sizeof(Real) == 4 || D <= 8;
    auto step = [&](auto first_c, auto last_c, const int s) {
        constexpr bool FIRST = decltype(first_c)::value, LAST = decltype(last_c)::value;
        const long ku = kw + s, k = k0 + s;
        if constexpr (FAST) take(std::integral_constant<bool, !LAST>{}, s);
        // predict
        Real FP[D], Pp[D];
        if (impq && !SMOOTH) {
            Real Pm[D];
#pragma unroll
            for (int i = 0; i < D; ++i) Pm[i] = P[i] - Pinf[i];
            zero<D>(FP); mm<D>(FP, Fc, Pm);
            copy<D>(Pp, Pinf);
            mm<D>(Pp, FP, Fr);
        } else if (impq) {
            // with the smoothing elements F P is needed as it is: Pp = Pinf + (F P - F Pinf) F^T
            Real FI[D], Dm[D];
            zero<D>(FP); mm<D>(FP, Fc, P);
            zero<D>(FI); mm<D>(FI, Fc, Pinf);
#pragma unroll
            for (int i = 0; i < D; ++i) Dm[i] = FP[i] - FI[i];
            copy<D>(Pp, Pinf);
            mm<D>(Pp, Dm, Fr);
        } else {
            zero<D>(FP); mm<D>(FP, Fc, P);
            copy<D>(Pp, Q);
            mm<D>(Pp, FP, Fr);
        }
        const Real mp = mvr<D>(Fr, m, Real(0.0));
        const Real yk = y;
        if constexpr (!FAST && !LAST && EARLY) load(s + 1);
        symmetrise<D>(Pp, patch, lane);
        // this step's rank-one update (direction, 1 / S, residual): the filter below and, DFORM, the element of step k-1
        const bool obs = !(yk != yk);
        Real u = Real(0.0), S = Real(1.0), mu = Real(0.0);
        if constexpr (!LAST || DFORM) {
            u = dot_h<D>(Pp, h);
            S = rowsum<D>(hl * u, a.R);
            mu = rowsum<D>(hl * mp, Real(0.0));
        }
        if constexpr (SMOOTH && !FIRST) {
            // element of step k-1: W = Pp^-1 F P = E^T (i.e. E in row layout), g = m - E mp, L = P - E F P
            Real M[D], W[D];
            copy<D>(M, Pp); copy<D>(W, FP);
            GjStep<D, 0>::run(M, W);
            const Real gn = m - mvr<D>(W, mp, Real(0.0));
            Real En[D], Ln[D], T[D];
            transpose<D>(W, En, patch, lane);
            zero<D>(T); mm<D>(T, En, FP);
#pragma unroll
            for (int i = 0; i < D; ++i) Ln[i] = P[i] - T[i];
            {
                const bool st = FAST || (k - 1 < k1);
                put(slotE, a.Es, ku - 1, st, En);
                if constexpr (FAST && BATCH) io.lds_put_sym(slotL, Ln);
                else if constexpr (FAST) io.wide_store_sym(IOT::sym_at(a.Lws, ku - 1), slotL, Ln);
                else io.template st_sym<false>(IOT::sym_at(a.Lws, ku - 1), st, Ln);
                io.template st_vec<FAST>(a.gs + (ku - 1) * D, st, gn);
            }
            // total <- total (x) element:  E = Ea En, g = Ea gn + ga, L = Ea Ln Ea^T + La
            Real E2[D];
            zero<D>(E2); mm<D>(E2, Ec, En);
            if constexpr (!DFORM) { zero<D>(T); mm<D>(T, Ec, Ln); }
            Real Er[D];                       // the total's E in row layout, made where it is used
            transpose<D>(Ec, Er, patch, lane);
            if constexpr (DFORM) {
                const Real dinv = obs ? Real(1.0) / S : Real(0.0);
                const Real dres = obs ? yk - mu : Real(0.0);
                const Real v = mvr<D>(W, u, Real(0.0));         // (E u)_lane: W is E in row layout
                const Real w = mvr<D>(Er, v, Real(0.0));        // its image under the total's gain
                g = g + w * (dres * dinv);
                rank1<D>(L, w, -w * dinv);
                copy<D>(Ec, E2);
            } else if (FAST) {
                g = mvr<D>(Er, gn, g);
                mm<D>(L, T, Er);
                symmetrise<D>(L, patch, lane);
                copy<D>(Ec, E2);
            } else {
                // Beyond the end of a segment that is NOT the last of its series there is nothing to fold: the
                // F = 0 steps would put E = 0 into a total that the ranks after this one still have to extend.
                const bool fold = a.seg_last || (k - 1 < a.N);
                const Real g2 = mvr<D>(Er, gn, g);
                Real L2[D];
                copy<D>(L2, L); mm<D>(L2, T, Er);
                symmetrise<D>(L2, patch, lane);
                g = fold ? g2 : g;
#pragma unroll
                for (int i = 0; i < D; ++i) { L[i] = fold ? L2[i] : L[i]; Ec[i] = fold ? E2[i] : Ec[i]; }
            }
        }
        if constexpr (!LAST) {
            const bool upd = FAST || k < k1;
            if (obs) ll.add((double)yk - (double)mu, (double)S);
            Real mb = mp;
            if ((!FAST || EDGE) && blockIdx.x == 0 && s == 0) {
                // first step of the series (chain 0 only): the update uses the prior itself (parallel.py:24-30),
                // the likelihood term above used F0 P0 F0^T + Q0 (parallel.py:136-141)
                const bool first = (c == 0 && a.seg_first);
                const Real u0 = dot_h<D>(P, h);
                const Real S0 = rowsum<D>(hl * u0, a.R), mu0 = rowsum<D>(hl * m, Real(0.0));
#pragma unroll
                for (int i = 0; i < D; ++i) Pp[i] = first ? P[i] : Pp[i];
                mb = first ? m : mp;
                u = first ? u0 : u;
                S = first ? S0 : S;
                mu = first ? mu0 : mu;
            }
            const Real inv = obs ? Real(1.0) / S : Real(0.0);
            const Real res = obs ? yk - mu : Real(0.0);
            m = mb + u * (inv * res);
            copy<D>(P, Pp); rank1<D>(P, u, -u * inv);
            if constexpr (!FAST && !EARLY) load(s + 1);
            if constexpr (STORE) {              // log-likelihood-only and projected-posterior calls skip these
                put(slotP, a.fPs, ku, upd, P);
                io.template st_vec<FAST>(a.fms + ku * D, upd, m);
            }
        }
        if constexpr (FAST && BATCH) {          // the step's matrix results leave together
            sync();
            if constexpr (SMOOTH && !FIRST) { io.wide_drain(a.Es + (ku - 1) * dd, slotE); io.wide_drain_sym(IOT::sym_at(a.Lws, ku - 1), slotL); }
            if constexpr (!LAST && STORE) io.wide_drain(a.fPs + ku * dd, slotP);
        }
    };
    using Yes = std::integral_constant<bool, true>;
    using No = std::integral_constant<bool, false>;
    if constexpr (SMOOTH) {
        step(Yes{}, No{}, 0);
        for (int s = 1; s < a.Lw; ++s) step(No{}, No{}, s);
        step(No{}, Yes{}, a.Lw);
    } else {
        // (the last step requests one step beyond the chunk: inside the series for a FAST wave, clamped otherwise)
        for (int s = 0; s < a.Lw; ++s) step(Yes{}, No{}, s);
    }
    if (cv) {
        if (SMOOTH && lv) {
            Real* rec = a.sagg1 + c * nsmth(D);
#pragma unroll
            for (int i = 0; i < D; ++i) { rec[i * D + lane] = Ec[i]; rec[dd + i * D + lane] = L[i]; }
            rec[2 * dd + lane] = g;
        }
        if (lane == 0) a.llpart[c] = ll.value();
    }
}

// STORE: the filtered moments are written (pkf / pkfs); not for the log-likelihood-only and projected-posterior calls
template <typename Real, int D, bool SMOOTH, bool IMPQS, bool STORE, bool DFORM = false>
__global__ __launch_bounds__(64, PGPS_RC_WAVES) void rc_apply1(const RcArgsT<Real> a0) {
    static_assert(SMOOTH || !IMPQS, "the implicit-noise instantiation is the smoothing one");
    __shared__ Real tl[4 * kPatch];
    const int lane = threadIdx.x & 15, row = threadIdx.x >> 4;
    __shared__ __attribute__((aligned(16))) char wslots[(Io<D, Real>::kBatchOut ? 5 : 3) * (Io<D, Real>::SLOT + 16)];
    lds_clear(wslots, (int)sizeof(wslots));
    Real* patch = patch_init(tl, row);
    const RcArgsT<Real> a = model_view(a0);
#ifdef PGPS_RC_FAST_ONLY
    apply1_body<Real, D, SMOOTH, true, IMPQS, STORE, false, DFORM>(a, patch, wslots, lane, row);
#else
    // EDGE: whole chains; the workgroup that ends the series only when nothing follows it (a segment that is not the last of
    // its series takes the first step of the next rank from the halo record: general body)
    const long end4 = ((long)blockIdx.x + 1) * 4 * a.Lw;
    // (the filter + smoothing-element kernels only: the filter-only ones fit two waves per SIMD and a third body would cost
    // them that)
    const bool edge_ok = SMOOTH && PGPS_RC_EDGE_FAST != 0 && end4 <= a.N && !(end4 == a.N && (a.halo_F != nullptr || !a.seg_last));
    if (blockIdx.x >= 1 && blockIdx.x < a.wfast) apply1_body<Real, D, SMOOTH, true, IMPQS, STORE, false, DFORM>(a, patch, wslots, lane, row);
#ifdef PGPS_RC_SKIP_EDGE               // (timing experiment: what the kernel costs without its edge waves; results are wrong)
    else return;
#else
    else if (edge_ok) apply1_body<Real, D, SMOOTH, true, IMPQS, STORE, SMOOTH, DFORM>(a, patch, wslots, lane, row);
    else apply1_body<Real, D, SMOOTH, false, IMPQS, STORE, false, DFORM>(a, patch, wslots, lane, row);
#endif
#endif
}

// ====================================================================================================
// level 1 of the stand-alone smoother (pks, parallel.py:187-196): smoothing elements from GIVEN filtered
// moments -- the element part of rc_apply1 with (m, P) of step k read from fms / fPs and the predict taken with
// F, Q of step k + 1 (parallel.py:159-173); stores E, g, L and the chain's smoothing total.
// ====================================================================================================
template <typename Real, int D, bool FAST>
__device__ __forceinline__ void selem1_body(const RcArgsT<Real>& a, Real* patch, int lane, int row) {
    constexpr int dd = D * D;
    const long kw = (long)blockIdx.x * 4 * a.Lw;
    const long c = (long)blockIdx.x * 4 + row;
    const long k0 = c * a.Lw, k1 = min(a.N, k0 + a.Lw);
    Io<D, Real> io;
    io.init(lane, row, a.Lw);
    const bool lv = io.lv, cv = c < a.nchunk;
    Real Ec[D], L[D], g = Real(0.0);               // smoothing total of the steps seen so far
#pragma unroll
    for (int i = 0; i < D; ++i) { Ec[i] = (i == lane) ? Real(1.0) : Real(0.0); L[i] = Real(0.0); }
    Real Fc[D], Fr[D], Q[D], P[D], m = Real(0.0);
    zero<D>(Fc); zero<D>(Fr); zero<D>(Q); zero<D>(P);
    // step k's filtered moments and step k + 1's (F, Q); at or beyond the end of the series F = 0, Q = I, which
    // makes the element (0, m, P): the last element for k = N - 1, a no-op on a total whose E is already 0 after it
    auto load = [&](int s) {
        const long ku = kw + s, k = k0 + s;
        if (FAST) {
            io.template mat_fast<false>(a.Fs + (ku + 1) * dd, Fc);
            io.template mat_fast<true>(a.Fs + (ku + 1) * dd, Fr);
            io.template mat_fast<false>(a.Qs + (ku + 1) * dd, Q);
            io.template mat_fast<false>(a.fPs + ku * dd, P);
            io.vec_fast(a.fms + ku * D, m);
        } else {
            const long kc = ku < a.N ? ku : a.N - 1, kn = ku + 1 < a.N ? ku + 1 : a.N - 1;
            const bool real = k < k1, nreal = k + 1 < a.N;
            io.template mat_slow<false>(a.Fs + kn * dd, nreal, Real(0.0), Fc);
            io.template mat_slow<true>(a.Fs + kn * dd, nreal, Real(0.0), Fr);
            io.template mat_slow<false>(a.Qs + kn * dd, nreal, Real(1.0), Q);
            io.template mat_slow<false>(a.fPs + kc * dd, real, Real(0.0), P);
            m = io.vec(a.fms + kc * D, real);
        }
    };
    load(0);
    for (int s = 0; s < a.Lw; ++s) {
        const long ku = kw + s, k = k0 + s;
        Real Ps[D];
        symmetrise<D>(P, patch, lane);           // the caller's filtered covariances need not be exactly symmetric
        copy<D>(Ps, P);
        const Real ms = m;
        Real FP[D], Pp[D];
        zero<D>(FP); mm<D>(FP, Fc, Ps);
        copy<D>(Pp, Q); mm<D>(Pp, FP, Fr);
        const Real mp = mvr<D>(Fr, ms, Real(0.0));
        if (s + 1 < a.Lw) load(s + 1);
        symmetrise<D>(Pp, patch, lane);
        Real W[D];
        copy<D>(W, FP);
        GjStep<D, 0>::run(Pp, W);                // Pp is not needed afterwards: eliminated in place
        const Real gn = ms - mvr<D>(W, mp, Real(0.0));
        Real En[D], Ln[D], T[D];
        transpose<D>(W, En, patch, lane);
        zero<D>(T); mm<D>(T, En, FP);
#pragma unroll
        for (int i = 0; i < D; ++i) Ln[i] = Ps[i] - T[i];
        {
            const bool st = FAST || k < k1;
            io.template st_mat<FAST>(a.Es + ku * dd, st, En);
            io.template st_sym<FAST>(Io<D, Real>::sym_at(a.Lws, ku), st, Ln);
            io.template st_vec<FAST>(a.gs + ku * D, st, gn);
        }
        Real E2[D], Er[D];
        zero<D>(E2); mm<D>(E2, Ec, En);
        zero<D>(T); mm<D>(T, Ec, Ln);
        transpose<D>(Ec, Er, patch, lane);
        g = mvr<D>(Er, gn, g);
        mm<D>(L, T, Er);
        symmetrise<D>(L, patch, lane);
        copy<D>(Ec, E2);
    }
    if (cv && lv) {
        Real* rec = a.sagg1 + c * nsmth(D);
#pragma unroll
        for (int i = 0; i < D; ++i) { rec[i * D + lane] = Ec[i]; rec[dd + i * D + lane] = L[i]; }
        rec[2 * dd + lane] = g;
    }
}

template <typename Real, int D>
__global__ __launch_bounds__(64) void rc_selem1(const RcArgsT<Real> a) {
    __shared__ Real tl[4 * kPatch];
    const int lane = threadIdx.x & 15, row = threadIdx.x >> 4;
    Real* patch = patch_init(tl, row);
    if (blockIdx.x < a.wfast) selem1_body<Real, D, true>(a, patch, lane, row);    // no first-step special case here
    else selem1_body<Real, D, false>(a, patch, lane, row);
}

// ====================================================================================================
// level 1: smoother -- sm = E sm' + g, sP = E sP' E^T + L from the stored elements (parallel.py:176-184)
// ====================================================================================================
// PROJ (pgps_lti_predict_*): nothing is stored per step; step k writes  H sm_k  and  H sP_k H^T  to slot qslot[k]
// when that is >= 0 (StateSpaceGP.predict_f keeps exactly those, pssgp/model.py:107-111)
template <typename Real, int D, bool FAST, bool PROJ>
__device__ __forceinline__ void smooth1_body(const RcArgsT<Real>& a, Real* patch, char* wslots, int lane, int row) {
    constexpr int dd = D * D;
    const long kw = (long)blockIdx.x * 4 * a.Lw;
    const long c = (long)blockIdx.x * 4 + row;
    const long k0 = c * a.Lw, k1 = min(a.N, k0 + a.Lw);
    Io<D, Real> io;
    io.init(lane, row, a.Lw);
    const bool lv = io.lv;
    // smoothed moments of the first step after the chunk: (g, L) of the inclusive suffix of the next chunk
    // (E = 0 there: every suffix contains the last element); nothing (0, 0) after the last chunk, whose own
    // last element has E = 0
    Real sm, sP[D];
    {
        // (the last chain of a segment that is not the last takes what follows from the ranks after it)
        const bool inner = c + 1 < a.nchunk;
        const bool nx = inner || (c + 1 == a.nchunk && a.carry_back != nullptr);
        const Real* rec = inner ? a.suf + (c + 1) * nsmth(D) : (a.carry_back ? a.carry_back : a.suf);
        sm = (nx && lv) ? rec[2 * dd + lane] : Real(0.0);
#pragma unroll
        for (int i = 0; i < D; ++i) sP[i] = (nx && lv) ? rec[dd + i * D + lane] : Real(0.0);
        if (a.dform && inner && lv) {
            // totals in innovation form (rc_apply1, DFORM): (g, L) of the suffix are sm - m, sP - P at the next chain's first
            // step -- add that step's filtered moments
            const long kn = (c + 1) * a.Lw;
            sm += a.fms[kn * D + lane];
#pragma unroll
            for (int i = 0; i < D; ++i) sP[i] += Real(0.5) * (a.fPs[kn * dd + i * D + lane] + a.fPs[kn * dd + lane * D + i]);
        }
    }
    Real Ec[D], Er[D], L[D], g = Real(0.0);
    zero<D>(Ec); zero<D>(Er); zero<D>(L);
    Real h[D];
    int q = -1;
    if (PROJ) {
#pragma unroll
        for (int i = 0; i < D; ++i) h[i] = a.H[i];
    }
    // stored element of this row's step; steps outside the chunk run as the identity element (I, 0, 0)
    auto load = [&](int s) {                    // the general body's per-register loads
        const long ku = kw + s, k = k0 + s;
        if (PROJ) { q = -1; if (k < k1) q = a.qslot[k]; }
        const long kc = ku < a.N ? ku : a.N - 1;
        const bool real = k < k1;
        io.template mat_slow<false>(a.Es + kc * dd, real, Real(1.0), Ec);
        io.template mat_slow<true>(a.Es + kc * dd, real, Real(1.0), Er);
        io.ld_sym(Io<D, Real>::sym_at(a.Lws, kc), real, L);
        g = io.vec(a.gs + kc * D, real);
    };
    // FAST: the stored elements as 16-byte pieces through LDS (Io, WIDE path), a whole step ahead; E's row layout
    // comes out of the same slot
    using IOT = Io<D, Real>;
    typename IOT::V4 pE[IOT::NVW], pL[IOT::NVS];
    Real ng = Real(0.0);
    int nq = -1;
    char* slotE = wslots;
    char* slotL = wslots + (IOT::SLOT + 16);
    char* slotO = wslots + 2 * (IOT::SLOT + 16);
    auto prefetch = [&](int s) {                // (s = -1 at the end: the step before the chunk; the series' first
        const long kp = kw + s < 0 ? 0 : kw + s;    // workgroup has none and fetches its step 0 again, unused)
        // (whole-series calls keep E in the caller's sPs array until this pass overwrites it: the descriptor ends where
        // that array ends -- the 16 bytes of slack a record's last piece is given elsewhere would be read beyond it)
        io.wide_load_cut(a.Es + kp * dd, pE, a.N - kp);
        io.wide_load_sym(IOT::sym_at(a.Lws, kp), pL);
        io.vec_fast(a.gs + kp * D, ng);
        if (PROJ) nq = a.qslot[k0 + s < 0 ? 0 : k0 + s];
    };
    auto take = [&](int s) {
        sync();
        io.wide_commit(slotE, pE);
        io.wide_commit_sym(slotL, pL);
        g = ng; q = nq;
        sync();
        PGPS_RC_PIN();
        prefetch(s - 1);
        PGPS_RC_PIN();
        io.template lds_get<false>(slotE, Ec);
        io.template lds_get<true>(slotE, Er);
        io.lds_get_sym(slotL, L);
    };
    if constexpr (FAST) { io.init_wide(row, a.Lw, (unsigned)IOT::SLOT); prefetch(a.Lw - 1); }
    else load(a.Lw - 1);
    for (int s = a.Lw - 1; s >= 0; --s) {
        const long ku = kw + s, k = k0 + s;
        if constexpr (FAST) take(s);
        Real T[D], nP[D];
        zero<D>(T); mm<D>(T, Ec, sP);
        copy<D>(nP, L); mm<D>(nP, T, Er);
        sm = mvr<D>(Er, sm, g);
        const int qk = q;
        if constexpr (!FAST) {
            if (s > 0) load(s - 1);
        }
        symmetrise<D, true>(nP, patch, lane);
        copy<D>(sP, nP);
        if (PROJ) {
            const Real mean = mvr<D>(h, sm, Real(0.0));                     // H sm            (every lane of the row)
            const Real var = mvr<D>(h, dot_h<D>(sP, h), Real(0.0));         // H sP H^T
            // branch-free: a descriptor over the K outputs, offset out of range unless this step is a query (lane 0 writes)
            const unsigned qo = (qk >= 0 && lane == 0) ? (unsigned)qk * (unsigned)sizeof(Real) : kOob;
            buf_st(make_rsrc(a.pmean, 0x7fff0000u), qo, mean);
            buf_st(make_rsrc(a.pvar, 0x7fff0000u), qo, var);
        } else {
            const bool st = FAST || k < k1;
            if constexpr (FAST) io.wide_store(a.sPs + ku * dd, slotO, sP);
            else io.template st_mat<false>(a.sPs + ku * dd, st, sP);
            io.template st_vec<FAST>(a.sms + ku * D, st, sm);
        }
    }
}

template <typename Real, int D, bool PROJ>
__global__ __launch_bounds__(64) void rc_smooth1(const RcArgsT<Real> a) {
    __shared__ Real tl[4 * kPatch];
    const int lane = threadIdx.x & 15, row = threadIdx.x >> 4;
    __shared__ __attribute__((aligned(16))) char wslots[3 * (Io<D, Real>::SLOT + 16)];
    lds_clear(wslots, (int)sizeof(wslots));
    Real* patch = patch_init(tl, row);
    // FAST: the four chains are complete (the smoother reads nothing beyond its chains: the first and the last workgroup of
    // a series qualify like any other -- until round 3 they took the general body, whose wave a one-round grid waits for)
    const bool full = ((long)blockIdx.x + 1) * 4 * a.Lw <= a.N;
    if (PGPS_RC_EDGE_FAST != 0 ? full : (blockIdx.x >= 1 && blockIdx.x < a.wfast)) smooth1_body<Real, D, true, PROJ>(a, patch, wslots, lane, row);
#ifdef PGPS_RC_SKIP_EDGE               // (timing experiment: what the kernel costs without its edge waves; results are wrong)
    else return;
#else
    else smooth1_body<Real, D, false, PROJ>(a, patch, wslots, lane, row);
#endif
}

// ====================================================================================================
// upper levels: Kogge-Stone scans over the chain totals, one ROW per record (four records per wave), operands
// in registers in the same column layout as level 1.
// ====================================================================================================
// Gauss-Jordan with partial pivoting, B <- M^-1 B (M general).  Column c lives in lane c, so every lane learns
// the candidates by broadcast and applies the same predicated row swaps to its own registers.
template <int D, int C>
struct GjPivStep {
    template <typename Real>
    static __device__ __forceinline__ void run(Real* M, Real* B) {
        Real pv = bcast<C>(M[C]);
#pragma unroll
        for (int r = C + 1; r < D; ++r) {
            const Real t = bcast<C>(M[r]);
            const bool sw = __builtin_fabs(t) > __builtin_fabs(pv);
            const Real mc = M[C], mr = M[r], bc = B[C], br = B[r];
            M[C] = sw ? mr : mc; M[r] = sw ? mc : mr;
            B[C] = sw ? br : bc; B[r] = sw ? bc : br;
            pv = sw ? t : pv;
        }
        const Real inv = Real(1.0) / pv;
        const Real mc = M[C] * inv, bv = B[C] * inv;
        if constexpr (D <= 8) {
            Gj<Real, D, C>::run(M, B, -mc, -bv);
        } else {
            Gj<Real, 8, C>::run(M, B, -mc, -bv);
            Gj<Real, D - 8, C>::run(M + 8, B + 8, -mc, -bv);
        }
        M[C] = mc;
        B[C] = bv;
        if constexpr (C + 1 < D) GjPivStep<D, C + 1>::run(M, B);
    }
};

template <int D, typename Real>
__device__ __forceinline__ void ld_rec_mat(const Real* g, bool ok, int lane, Real* X) {
    if (ok) {
#pragma unroll
        for (int i = 0; i < D; ++i) X[i] = g[i * D + lane];
    }
}
template <int D, typename Real>
__device__ __forceinline__ void st_rec_mat(Real* g, bool ok, int lane, const Real* X) {
    if (ok) {
#pragma unroll
        for (int i = 0; i < D; ++i) g[i * D + lane] = X[i];
    }
}

// e1 (x) e2, the general filtering operator (filt_combine of pgps_math.h, parallel.py:100-118), rearranged so
// that one elimination with a single right-hand side serves everything:
//   M = I + C1 J2;  Nm = M^-1 C1 (symmetric);  z = eta2 - J2 b1;  W = J2 A1;  G = M^-1 A1 = A1 - Nm W
//   A = A2 G;  b = A2 (b1 + Nm z) + b2;  C = sym(A2 Nm A2^T) + C2;  eta = G^T z + eta1;  J = sym(G^T W) + J1
// Outputs: Ao, bo, eo, and C2 / J1 updated in place to the result's C / J.
template <int D, typename Real>
__device__ __forceinline__ void filt_combine_rc(Real* patch, int lane, const Real* A1, const Real* C1, Real* J1,
                                                Real b1, Real e1, const Real* A2, Real* C2, const Real* J2,
                                                Real b2, Real e2, Real* Ao, Real& bo, Real& eo) {
    Real M[D], Nm[D];
#pragma unroll
    for (int i = 0; i < D; ++i) { M[i] = (i == lane) ? Real(1.0) : Real(0.0); Nm[i] = C1[i]; }
    mm<D>(M, C1, J2);
    GjPivStep<D, 0>::run(M, Nm);
    const Real z = e2 - mvr<D>(J2, b1, Real(0.0));
    Real W[D], T[D], G[D];
    zero<D>(W); mm<D>(W, J2, A1);
    zero<D>(T); mm<D>(T, Nm, W);
#pragma unroll
    for (int i = 0; i < D; ++i) G[i] = A1[i] - T[i];
    Real X[D], A2r[D], Gt[D];
    zero<D>(Ao); mm<D>(Ao, A2, G);
    zero<D>(X); mm<D>(X, A2, Nm);
    transpose<D>(A2, A2r, patch, lane);
    mm<D>(C2, X, A2r);                          // C2 <- C2 + A2 Nm A2^T
    symmetrise<D>(C2, patch, lane);
    const Real w = mvr<D>(Nm, z, b1);         // b1 + Nm z   (Nm symmetric: its column layout is its row layout)
    bo = mvr<D>(A2r, w, b2);
    eo = mvr<D>(G, z, e1);                      // G^T z + eta1
    transpose<D>(G, Gt, patch, lane);
    mm<D>(J1, Gt, W);                           // J1 <- J1 + G^T W
    symmetrise<D>(J1, patch, lane);
}

// a (x) b in time order (smth_combine, parallel.py:176-184):  E = Ea Eb;  g = Ea gb + ga;  L = sym(Ea Lb Ea^T) + La
template <int D, typename Real>
__device__ __forceinline__ void smth_combine_rc(Real* patch, int lane, const Real* Ea, const Real* La, Real ga,
                                                const Real* Eb, const Real* Lb, Real gb, Real* Eo, Real* Lo,
                                                Real& go) {
    Real T[D], Ear[D];
    zero<D>(Eo); mm<D>(Eo, Ea, Eb);
    zero<D>(T); mm<D>(T, Ea, Lb);
    transpose<D>(Ea, Ear, patch, lane);
    go = mvr<D>(Ear, gb, ga);
    copy<D>(Lo, La); mm<D>(Lo, T, Ear);
    symmetrise<D>(Lo, patch, lane);
}

// compact records in global memory: filter [A | C | J | b | eta], smoother [E | L | g]
template <int D, typename Real>
__device__ __forceinline__ void ld_filt(const Real* r, bool ok, int lane, Real* A, Real* C, Real* J, Real& b,
                                        Real& e) {
    constexpr int dd = D * D;
    ld_rec_mat<D>(r, ok, lane, A); ld_rec_mat<D>(r + dd, ok, lane, C); ld_rec_mat<D>(r + 2 * dd, ok, lane, J);
    if (ok) { b = r[3 * dd + lane]; e = r[3 * dd + D + lane]; }
}
template <int D, typename Real>
__device__ __forceinline__ void st_filt(Real* r, bool ok, int lane, const Real* A, const Real* C, const Real* J,
                                        Real b, Real e) {
    constexpr int dd = D * D;
    st_rec_mat<D>(r, ok, lane, A); st_rec_mat<D>(r + dd, ok, lane, C); st_rec_mat<D>(r + 2 * dd, ok, lane, J);
    if (ok) { r[3 * dd + lane] = b; r[3 * dd + D + lane] = e; }
}

// out[c] = in[c - stride] (x) in[c]; with `fixed` given: out[c] = fixed (x) in[c] for every c (a segment's
// carry-in combined into all its prefixes)
// With fixblk > 0 (the last stage of the blocked scan below): record c takes the fixed operand fixed[c / fixblk - 1] --
// the scanned total of the blocks before its own -- and the records of block 0 pass through.
template <typename Real, int D>
__global__ __launch_bounds__(64) void rc_ks_filter(long n, long stride, const Real* in, Real* out, long bstride,
                                                   const Real* fixed, long fixblk = 0) {
    __shared__ Real tl[4 * kPatch];
    const int lane = threadIdx.x & 15, row = threadIdx.x >> 4;
    Real* patch = patch_init(tl, row);
    in += blockIdx.y * bstride;
    out += blockIdx.y * bstride;
    constexpr int dd = D * D, nf = 3 * D * D + 2 * D;
    const long c = (long)blockIdx.x * 4 + row;
    const bool lv = lane < D, cv = c < n, comb = cv && (fixblk > 0 ? c >= fixblk : (fixed != nullptr || c >= stride));
    const bool ok = lv && cv;
    // later operand (also the pass-through value)
    Real A2[D], C2[D], J2[D], b2 = Real(0.0), e2 = Real(0.0);
    zero<D>(A2); zero<D>(C2); zero<D>(J2);
    const Real* r2 = in + (cv ? c : 0) * nf;
    ld_filt<D>(r2, ok, lane, A2, C2, J2, b2, e2);
    // earlier operand; rows that only pass through combine with the identity (A = I): same arithmetic, result unused
    Real A1[D], C1[D], J1[D], b1 = Real(0.0), e1 = Real(0.0);
#pragma unroll
    for (int i = 0; i < D; ++i) { A1[i] = (i == lane) ? Real(1.0) : Real(0.0); C1[i] = Real(0.0); J1[i] = Real(0.0); }
    const Real* r1 = fixblk > 0 ? fixed + (comb ? c / fixblk - 1 : 0) * nf : (fixed ? fixed : in + (comb ? c - stride : 0) * nf);
    ld_filt<D>(r1, lv && comb, lane, A1, C1, J1, b1, e1);
    Real Ao[D], bo, eo;
    filt_combine_rc<D>(patch, lane, A1, C1, J1, b1, e1, A2, C2, J2, b2, e2, Ao, bo, eo);
    Real* ro = out + (cv ? c : 0) * nf;
    if (comb) {
        st_filt<D>(ro, ok, lane, Ao, C2, J1, bo, eo);
    } else if (ok) {
        // pass-through: reload (C2 was overwritten above) and copy
#pragma unroll
        for (int i = 0; i < D; ++i) {
            ro[i * D + lane] = A2[i];
            ro[dd + i * D + lane] = r2[dd + i * D + lane];
            ro[2 * dd + i * D + lane] = J2[i];
        }
        ro[3 * dd + lane] = b2;
        ro[3 * dd + D + lane] = e2;
    }
}

// out[c] = in[c] (x) in[c + stride] in time order; with `fixed` given: out[c] = in[c] (x) fixed for every c
// (fixblk > 0: record c takes fixed[c / fixblk + 1], the scanned total of the blocks after its own; the last block passes)
template <typename Real, int D>
__global__ __launch_bounds__(64) void rc_ks_smoother(long n, long stride, const Real* in, Real* out, const Real* fixed,
                                                     long fixblk = 0) {
    __shared__ Real tl[4 * kPatch];
    const int lane = threadIdx.x & 15, row = threadIdx.x >> 4;
    Real* patch = patch_init(tl, row);
    constexpr int dd = D * D, ns = 2 * D * D + D;
    const long c = (long)blockIdx.x * 4 + row;
    const long nblk = fixblk > 0 ? (n + fixblk - 1) / fixblk : 0;
    const bool lv = lane < D, cv = c < n,
               comb = cv && (fixblk > 0 ? c / fixblk + 1 < nblk : (fixed != nullptr || c + stride < n));
    const bool ok = lv && cv, okb = lv && comb;
    Real Ea[D], La[D], ga = Real(0.0);
    zero<D>(Ea); zero<D>(La);
    const Real* ra = in + (cv ? c : 0) * ns;
    ld_rec_mat<D>(ra, ok, lane, Ea); ld_rec_mat<D>(ra + dd, ok, lane, La);
    if (ok) ga = ra[2 * dd + lane];
    Real Eb[D], Lb[D], gb = Real(0.0);
#pragma unroll
    for (int i = 0; i < D; ++i) { Eb[i] = (i == lane) ? Real(1.0) : Real(0.0); Lb[i] = Real(0.0); }      // identity when passing through
    const Real* rb = fixblk > 0 ? fixed + (comb ? c / fixblk + 1 : 0) * ns : (fixed ? fixed : in + (comb ? c + stride : 0) * ns);
    ld_rec_mat<D>(rb, okb, lane, Eb); ld_rec_mat<D>(rb + dd, okb, lane, Lb);
    if (okb) gb = rb[2 * dd + lane];
    Real Eo[D], Lo[D], go;
    smth_combine_rc<D>(patch, lane, Ea, La, ga, Eb, Lb, gb, Eo, Lo, go);
    Real* ro = out + (cv ? c : 0) * ns;
    if (comb) {
        st_rec_mat<D>(ro, ok, lane, Eo); st_rec_mat<D>(ro + dd, ok, lane, Lo);
        if (ok) ro[2 * dd + lane] = go;
    } else {
        st_rec_mat<D>(ro, ok, lane, Ea); st_rec_mat<D>(ro + dd, ok, lane, La);
        if (ok) ro[2 * dd + lane] = ga;
    }
}

// ====================================================================================================
// Blocked scan of the chain totals (round 3).  The Kogge-Stone scan above is one LAUNCH per level -- 12 to 14 launches
// per scan, each a 10 - 15 us round trip of every record through global memory (c5: 24 launches = 0.3 ms of a 2.7 ms
// pass).  Here a workgroup of B rows (one 16-lane row per record, B <= 64 by the LDS budget) runs log2(B) levels inside
// ONE launch, exchanging records through LDS, and writes its block's inclusive local scan plus the block total; the
// totals are scanned the same way (recursively: n / B, n / B^2, ...), and one more launch per level combines the scanned
// total of the neighbouring blocks into every record (rc_ks_* with fixblk).  n = 4096 at B = 16: 5 launches instead of
// 12; n = 16384 at B = 64: 5 instead of 14.  The scan is in place.
// ====================================================================================================
template <typename Real, int D, int B>
__global__ __launch_bounds__(B * 16) void rc_scan_blk_f(long n, Real* data, Real* tot) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int dd = D * D, nf = 3 * D * D + 2 * D;
    Real* recs = reinterpret_cast<Real*>(smem_raw);                 // B records
    Real* tl = recs + (size_t)B * nf;                               // B transpose patches
    const int lane = threadIdx.x & 15, row = threadIdx.x >> 4;
    for (int e = threadIdx.x; e < B * kPatch; e += blockDim.x) tl[e] = Real(0.0);
    __syncthreads();
    Real* patch = tl + row * kPatch;
    const long c = (long)blockIdx.x * B + row;
    const bool lv = lane < D, cv = c < n, ok = lv && cv;
    // this row's record; rows beyond the last record hold the identity (A = I)
    Real A2[D], C2[D], J2[D], b2 = Real(0.0), e2 = Real(0.0);
#pragma unroll
    for (int i = 0; i < D; ++i) { A2[i] = (i == lane && lv) ? Real(1.0) : Real(0.0); C2[i] = Real(0.0); J2[i] = Real(0.0); }
    Real* rg = data + (cv ? c : 0) * nf;
    ld_filt<D>(rg, ok, lane, A2, C2, J2, b2, e2);
    Real* mine = recs + (size_t)row * nf;
    // records in this block: a level whose stride reaches past them changes nothing -- a block of four totals (the second
    // pass over a short series' chains) runs two levels, not log2(B) (a level is the latency of one combine: 4.7 us at d = 6)
    const long left = n - (long)blockIdx.x * B;
    const int nvalid = left < B ? (int)left : B;
    for (int s = 1; s < nvalid; s <<= 1) {
        if (lv) st_filt<D>(mine, true, lane, A2, C2, J2, b2, e2);
        __syncthreads();
        const bool have = row >= s;
        // earlier operand: the record s rows before (rows without one combine with the identity and keep their own)
        Real A1[D], C1[D], J1[D], b1 = Real(0.0), e1 = Real(0.0);
#pragma unroll
        for (int i = 0; i < D; ++i) { A1[i] = (i == lane && lv) ? Real(1.0) : Real(0.0); C1[i] = Real(0.0); J1[i] = Real(0.0); }
        ld_filt<D>(recs + (size_t)(have ? row - s : 0) * nf, lv && have, lane, A1, C1, J1, b1, e1);
        Real Ao[D], Co[D], bo, eo;
        copy<D>(Co, C2);
        filt_combine_rc<D>(patch, lane, A1, C1, J1, b1, e1, A2, Co, J2, b2, e2, Ao, bo, eo);
#pragma unroll
        for (int i = 0; i < D; ++i) { A2[i] = have ? Ao[i] : A2[i]; C2[i] = have ? Co[i] : C2[i]; J2[i] = have ? J1[i] : J2[i]; }
        b2 = have ? bo : b2;
        e2 = have ? eo : e2;
        __syncthreads();
    }
    st_filt<D>(rg, ok, lane, A2, C2, J2, b2, e2);
    if (row == nvalid - 1 && lv) st_filt<D>(tot + (long)blockIdx.x * nf, true, lane, A2, C2, J2, b2, e2);
}

template <typename Real, int D, int B>
__global__ __launch_bounds__(B * 16) void rc_scan_blk_s(long n, Real* data, Real* tot) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int dd = D * D, ns = 2 * D * D + D;
    Real* recs = reinterpret_cast<Real*>(smem_raw);
    Real* tl = recs + (size_t)B * ns;
    const int lane = threadIdx.x & 15, row = threadIdx.x >> 4;
    for (int e = threadIdx.x; e < B * kPatch; e += blockDim.x) tl[e] = Real(0.0);
    __syncthreads();
    Real* patch = tl + row * kPatch;
    const long c = (long)blockIdx.x * B + row;
    const bool lv = lane < D, cv = c < n, ok = lv && cv;
    // this row's record (the EARLIER operand of a suffix step); rows beyond the last record hold the identity (E = I)
    Real Ea[D], La[D], ga = Real(0.0);
#pragma unroll
    for (int i = 0; i < D; ++i) { Ea[i] = (i == lane && lv) ? Real(1.0) : Real(0.0); La[i] = Real(0.0); }
    Real* rg = data + (cv ? c : 0) * ns;
    ld_rec_mat<D>(rg, ok, lane, Ea); ld_rec_mat<D>(rg + dd, ok, lane, La);
    if (ok) ga = rg[2 * dd + lane];
    Real* mine = recs + (size_t)row * ns;
    const long left = n - (long)blockIdx.x * B;
    const int nvalid = left < B ? (int)left : B;           // (see rc_scan_blk_f)
    for (int s = 1; s < nvalid; s <<= 1) {
        if (lv) { st_rec_mat<D>(mine, true, lane, Ea); st_rec_mat<D>(mine + dd, true, lane, La); mine[2 * dd + lane] = ga; }
        __syncthreads();
        const bool have = row + s < B;
        Real Eb[D], Lb[D], gb = Real(0.0);
#pragma unroll
        for (int i = 0; i < D; ++i) { Eb[i] = (i == lane && lv) ? Real(1.0) : Real(0.0); Lb[i] = Real(0.0); }
        const Real* rb = recs + (size_t)(have ? row + s : 0) * ns;
        ld_rec_mat<D>(rb, lv && have, lane, Eb); ld_rec_mat<D>(rb + dd, lv && have, lane, Lb);
        if (lv && have) gb = rb[2 * dd + lane];
        Real Eo[D], Lo[D], go;
        smth_combine_rc<D>(patch, lane, Ea, La, ga, Eb, Lb, gb, Eo, Lo, go);
#pragma unroll
        for (int i = 0; i < D; ++i) { Ea[i] = have ? Eo[i] : Ea[i]; La[i] = have ? Lo[i] : La[i]; }
        ga = have ? go : ga;
        __syncthreads();
    }
    st_rec_mat<D>(rg, ok, lane, Ea); st_rec_mat<D>(rg + dd, ok, lane, La);
    if (ok) rg[2 * dd + lane] = ga;
    if (row == 0 && lv) {
        Real* rt = tot + (long)blockIdx.x * ns;
        st_rec_mat<D>(rt, true, lane, Ea); st_rec_mat<D>(rt + dd, true, lane, La);
        rt[2 * dd + lane] = ga;
    }
}

// ---- segment stitching (multi-GPU, pssgp/distributed.py): the exchanged records are the lane-chunk family's
// packed ones, filter [A | b | C sym | J sym | eta | F_0 | Q_0], smoother [E | g | L sym | pad | ll] -------------
template <int D>
__device__ __forceinline__ int symidx(int i, int j) { return i <= j ? (i * D - (i * (i - 1)) / 2 + (j - i)) : (j * D - (j * (j - 1)) / 2 + (i - j)); }

// carry-in of segment `rank` (> 0): total_0 (x) ... (x) total_{rank-1}, written as a compact filter record.  One
// wave; the four rows do the same work, row 0 stores.
template <typename Real, int D>
__global__ __launch_bounds__(64) void rc_seg_carry_f(const Real* gathered, int rank, int reclen, Real* out) {
    __shared__ Real tl[4 * kPatch];
    const int lane = threadIdx.x & 15, row = threadIdx.x >> 4;
    Real* patch = patch_init(tl, row);
    constexpr int dd = D * D, SYM = D * (D + 1) / 2;
    const bool lv = lane < D;
    auto load = [&](int j, Real* A, Real* C, Real* J, Real& b, Real& e) {
        const Real* r = gathered + (long)j * reclen;
        zero<D>(A); zero<D>(C); zero<D>(J); b = Real(0.0); e = Real(0.0);
        if (lv) {
#pragma unroll
            for (int i = 0; i < D; ++i) {
                A[i] = r[i * D + lane];
                C[i] = r[dd + D + symidx<D>(i, lane)];
                J[i] = r[dd + D + SYM + symidx<D>(i, lane)];
            }
            b = r[dd + lane];
            e = r[dd + D + 2 * SYM + lane];
        }
    };
    Real A1[D], C1[D], J1[D], b1, e1;
    load(0, A1, C1, J1, b1, e1);
    for (int j = 1; j < rank; ++j) {
        Real A2[D], C2[D], J2[D], b2, e2, Ao[D], bo, eo;
        load(j, A2, C2, J2, b2, e2);
        filt_combine_rc<D>(patch, lane, A1, C1, J1, b1, e1, A2, C2, J2, b2, e2, Ao, bo, eo);
        copy<D>(A1, Ao); copy<D>(C1, C2);       // J1 already holds the result's J
        b1 = bo; e1 = eo;
    }
    st_filt<D>(out, lv && row == 0, lane, A1, C1, J1, b1, e1);
}

// what follows segment `rank` (< nranks - 1): total_{rank+1} (x) ... (x) total_{nranks-1} as a compact smoother record
template <typename Real, int D>
__global__ __launch_bounds__(64) void rc_seg_carry_s(const Real* gathered, int rank, int nranks, int reclen, Real* out) {
    __shared__ Real tl[4 * kPatch];
    const int lane = threadIdx.x & 15, row = threadIdx.x >> 4;
    Real* patch = patch_init(tl, row);
    constexpr int dd = D * D;
    const bool lv = lane < D;
    auto load = [&](int j, Real* E, Real* L, Real& g) {
        const Real* r = gathered + (long)j * reclen;
        zero<D>(E); zero<D>(L); g = Real(0.0);
        if (lv) {
#pragma unroll
            for (int i = 0; i < D; ++i) { E[i] = r[i * D + lane]; L[i] = r[dd + D + symidx<D>(i, lane)]; }
            g = r[dd + lane];
        }
    };
    Real Ea[D], La[D], ga;
    load(rank + 1, Ea, La, ga);
    for (int j = rank + 2; j < nranks; ++j) {
        Real Eb[D], Lb[D], gb, Eo[D], Lo[D], go;
        load(j, Eb, Lb, gb);
        smth_combine_rc<D>(patch, lane, Ea, La, ga, Eb, Lb, gb, Eo, Lo, go);
        copy<D>(Ea, Eo); copy<D>(La, Lo); ga = go;
    }
    const bool ok = lv && row == 0;
    st_rec_mat<D>(out, ok, lane, Ea); st_rec_mat<D>(out + dd, ok, lane, La);
    if (ok) out[2 * dd + lane] = ga;
}

// ====================================================================================================
// LTI discretisation (kernels/base.py:29-47): Fs[k] = expm(dt_k F) by Pade-13 scaling and squaring (what
// tf.linalg.expm implements), Qs[k] = Pinf - Fs[k] Pinf Fs[k]^T (equal to the reference's matrix-fraction
// expression because Pinf solves the Lyapunov equation; tests/test_oracle.py).  One ROW per time step, every
// matrix in registers; a wave takes `per` steps per row one after the other.
// ====================================================================================================
template <int D>
__global__ __launch_bounds__(64) void rc_discretise(long N, int per, const double* Fg, const double* Pg, const double* ts,
                                                    double t_prev, double* Fs, double* Qs, long bs_model) {
    __shared__ double tl[4 * kPatch];
    const int lane = threadIdx.x & 15, row = threadIdx.x >> 4;
    double* patch = patch_init(tl, row);
    constexpr int dd = D * D;
    Fg += blockIdx.y * bs_model; Pg += blockIdx.y * bs_model;         // batched: one model per blockIdx.y
    Fs += blockIdx.y * N * dd;
    if (Qs) Qs += blockIdx.y * N * dd;
    const bool lv = lane < D;
    const double b[14] = {64764752532480000., 32382376266240000., 7771770303897600., 1187353796428800.,
                          129060195264000., 10559470521600., 670442572800., 33522128640.,
                          1323241920., 40840800., 960960., 16380., 182., 1.};
    double F[D], Pm[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        F[i] = lv ? Fg[i * D + lane] : 0.0;
        Pm[i] = lv ? 0.5 * (Pg[i * D + lane] + Pg[lane * D + i]) : 0.0;
    }
    // 1-norm of F: column sums are lane-local, the maximum goes round the row
    double colsum = 0.0;
#pragma unroll
    for (int i = 0; i < D; ++i) colsum += __builtin_fabs(F[i]);
    double normF = mvr_max<D>(colsum);
    // A regular grid (the reference's benchmarks sample np.linspace; monthly or weekly records) has ONE transition: when the
    // step of every row of the wave equals the one it discretised last -- to a few ulp: differences of equally spaced
    // times are not bitwise equal -- the matrices of the previous iteration are stored again instead of recomputed
    // (expm(dt F) moves by |F dt| times that relative difference: 1e-15).  d = 11, 2^20 equal steps: 1.2 -> 0.3 ms.
    double dt_last = __builtin_nan("");
    double R[D], Qk[D];
    zero<D>(R); zero<D>(Qk);
    for (int q = 0; q < per; ++q) {
        const long k = ((long)blockIdx.x * per + q) * 4 + row;
        const bool kv = k < N;
        const long kc = kv ? k : N - 1;
        const double dt = ts[kc] - (kc > 0 ? ts[kc - 1] : t_prev);
        const bool same = __builtin_fabs(dt - dt_last) <= 8.0 * 2.220446049250313e-16 * __builtin_fabs(dt);
        dt_last = dt;
        if (__all(same)) {
            if (lv && kv) {
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    Fs[k * dd + i * D + lane] = R[i];
                    if (Qs) Qs[k * dd + i * D + lane] = Qk[i];
                }
            }
            continue;
        }
        int sq = 0;
        const double nrm = __builtin_fabs(dt) * normF;
        if (nrm > 5.371920351148152) {
            sq = (int)ceil(log2(nrm / 5.371920351148152));
            sq = sq < 0 ? 0 : (sq > 60 ? 60 : sq);
        }
        const double sc = ldexp(dt, -sq);
        double A[D], A2[D], A4[D], A6[D], W[D], Z[D];
#pragma unroll
        for (int i = 0; i < D; ++i) A[i] = sc * F[i];
        zero<D>(A2); mm<D>(A2, A, A);
        zero<D>(A4); mm<D>(A4, A2, A2);
        zero<D>(A6); mm<D>(A6, A4, A2);
        // Degree of the approximant, as tf.linalg.expm chooses it from the 1-norm (theta_7 = 0.95, else 13); one
        // choice per wave -- the largest norm of its four rows decides.
        double nmax = nrm;
        nmax = fmax(nmax, __shfl_xor(nmax, 16, 64));
        nmax = fmax(nmax, __shfl_xor(nmax, 32, 64));
        double U[D], V[D];
        if (nmax <= 0.9504178996162932) {
            // Pade-7: U = A (c7 A6 + c5 A4 + c3 A2 + c1 I), V = c6 A6 + c4 A4 + c2 A2 + c0 I
            const double c[8] = {17297280., 8648640., 1995840., 277200., 25200., 1512., 56., 1.};
#pragma unroll
            for (int i = 0; i < D; ++i) {
                Z[i] = c[7] * A6[i] + c[5] * A4[i] + c[3] * A2[i] + ((i == lane) ? c[1] : 0.0);
                V[i] = c[6] * A6[i] + c[4] * A4[i] + c[2] * A2[i] + ((i == lane) ? c[0] : 0.0);
            }
            zero<D>(U); mm<D>(U, A, Z);
            // |A|_1 <= theta_7 makes (V - U) / c0 = I - E with |E|_1 <= e^{0.48} - 1 < 1: strictly column diagonally
            // dominant, elimination without pivoting is stable (lanes >= D: give their zero columns a unit diagonal)
            double M[D];
#pragma unroll
            for (int i = 0; i < D; ++i) { M[i] = V[i] - U[i] + ((i == lane && !lv) ? 1.0 : 0.0); R[i] = V[i] + U[i]; }
            GjStep<D, 0>::run(M, R);
        } else {
            // Pade-13: U = A (A6 (b13 A6 + b11 A4 + b9 A2) + b7 A6 + b5 A4 + b3 A2 + b1 I)
#pragma unroll
            for (int i = 0; i < D; ++i) {
                W[i] = b[13] * A6[i] + b[11] * A4[i] + b[9] * A2[i];
                Z[i] = b[7] * A6[i] + b[5] * A4[i] + b[3] * A2[i] + ((i == lane) ? b[1] : 0.0);
            }
            mm<D>(Z, A6, W);
            zero<D>(U); mm<D>(U, A, Z);
            // V = A6 (b12 A6 + b10 A4 + b8 A2) + b6 A6 + b4 A4 + b2 A2 + b0 I
#pragma unroll
            for (int i = 0; i < D; ++i) {
                W[i] = b[12] * A6[i] + b[10] * A4[i] + b[8] * A2[i];
                V[i] = b[6] * A6[i] + b[4] * A4[i] + b[2] * A2[i] + ((i == lane) ? b[0] : 0.0);
            }
            mm<D>(V, A6, W);
            // (V - U) R = V + U, partial pivoting (lanes >= D: zero columns, they never pivot)
            double M[D];
#pragma unroll
            for (int i = 0; i < D; ++i) { M[i] = V[i] - U[i]; R[i] = V[i] + U[i]; }
            GjPivStep<D, 0>::run(M, R);
        }
        // squarings: the count differs between the rows of a wave
        int smax = sq;
        smax = max(smax, __shfl_xor(smax, 16, 64));
        smax = max(smax, __shfl_xor(smax, 32, 64));
        for (int t = 0; t < smax; ++t) {
            double R2[D];
            zero<D>(R2); mm<D>(R2, R, R);
            const bool on = t < sq;
#pragma unroll
            for (int i = 0; i < D; ++i) R[i] = on ? R2[i] : R[i];
        }
        if (Qs == nullptr) {                    // implicit process noise: only F_k is wanted
            if (lv && kv) {
#pragma unroll
                for (int i = 0; i < D; ++i) Fs[k * dd + i * D + lane] = R[i];
            }
            continue;
        }
        // Q = Pinf - sym(R Pinf R^T)
        double T[D], Rr[D], X[D];
        zero<D>(T); mm<D>(T, R, Pm);
        transpose<D>(R, Rr, patch, lane);
        zero<D>(X); mm<D>(X, T, Rr);
        symmetrise<D>(X, patch, lane);
#pragma unroll
        for (int i = 0; i < D; ++i) Qk[i] = Pm[i] - X[i];
        if (lv && kv) {
#pragma unroll
            for (int i = 0; i < D; ++i) { Fs[k * dd + i * D + lane] = R[i]; Qs[k * dd + i * D + lane] = Qk[i]; }
        }
    }
}

// ---- host side: the level-1 launches of one instantiation ---------------------------------------------
// phase 0: reduce, 1: apply + smoothing elements, 2: apply only, 3: smoother, 4: smoother writing projections,
// 5: smoothing elements from given filtered moments (stand-alone pks)
template <typename Real, int D>
int launch_rc_level1(pgps_ctx* ctx, const RcArgsT<Real>& a, int phase) {
    const dim3 blk(64), g1((unsigned)((a.nchunk + 3) / 4), (unsigned)(a.batch > 1 ? a.batch : 1));
    const bool st = a.store_f != 0;
    switch (phase) {
        case 0: timed_launch(ctx, PGPS_K_FILTER_REDUCE, rc_reduce1<Real, D>, g1, blk, 0u, a); break;
        case 1:
            if (a.implicit_q & 1) {
                if (st) timed_launch(ctx, PGPS_K_FILTER_APPLY, rc_apply1<Real, D, true, true, true>, g1, blk, 0u, a);
                else timed_launch(ctx, PGPS_K_FILTER_APPLY, rc_apply1<Real, D, true, true, false>, g1, blk, 0u, a);
            } else {
                if (st && a.dform) timed_launch(ctx, PGPS_K_FILTER_APPLY, rc_apply1<Real, D, true, false, true, true>, g1, blk, 0u, a);
                else if (st) timed_launch(ctx, PGPS_K_FILTER_APPLY, rc_apply1<Real, D, true, false, true>, g1, blk, 0u, a);
                else timed_launch(ctx, PGPS_K_FILTER_APPLY, rc_apply1<Real, D, true, false, false>, g1, blk, 0u, a);
            }
            break;
        case 2:
            if (st) timed_launch(ctx, PGPS_K_FILTER_APPLY, rc_apply1<Real, D, false, false, true>, g1, blk, 0u, a);
            else timed_launch(ctx, PGPS_K_FILTER_APPLY, rc_apply1<Real, D, false, false, false>, g1, blk, 0u, a);
            break;
        case 3: timed_launch(ctx, PGPS_K_SMOOTHER_APPLY, rc_smooth1<Real, D, false>, g1, blk, 0u, a); break;
        case 5: timed_launch(ctx, PGPS_K_SMOOTHER_REDUCE, rc_selem1<Real, D>, g1, blk, 0u, a); break;
        default: timed_launch(ctx, PGPS_K_SMOOTHER_APPLY, rc_smooth1<Real, D, true>, g1, blk, 0u, a); break;
    }
    HIPCHK(ctx, hipGetLastError());
    return PGPS_OK;
}

// one Kogge-Stone step over n records: which = 0 filter totals (prefix), 1 smoothing totals (suffix)
template <typename Real, int D>
int launch_rc_ks(pgps_ctx* ctx, int which, long n, long stride, const Real* in, Real* out, int batch, long bstride,
                 const Real* fixed) {
    const dim3 blk(64), g((unsigned)((n + 3) / 4), (unsigned)batch);
    if (which == 0) timed_launch(ctx, PGPS_K_FILTER_REDUCE, rc_ks_filter<Real, D>, g, blk, 0u, n, stride, in, out, bstride, fixed, 0L);
    else timed_launch(ctx, PGPS_K_SMOOTHER_REDUCE, rc_ks_smoother<Real, D>, g, blk, 0u, n, stride, in, out, fixed, 0L);
    HIPCHK(ctx, hipGetLastError());
    return PGPS_OK;
}

// rows (records) per workgroup of the blocked scan: the largest power of two <= 64 whose records + transpose patches
// fit 144 KiB of LDS
#ifndef PGPS_RC_SCAN_ROWS
#define PGPS_RC_SCAN_ROWS 32
#endif
template <typename Real, int D>
constexpr int scan_block_rows(int which) {
    const int rec = which == 0 ? 3 * D * D + 2 * D : 2 * D * D + D;
    // (a level of a block costs one combine per wave, the block's waves sharing four SIMDs: at 64 rows -- sixteen waves --
    // a launch of six levels took 36 us at d = 6 fp32, whatever the number of blocks; eight waves do five in about half)
    int b = PGPS_RC_SCAN_ROWS;
    while (b > 2 && (size_t)b * (rec + kPatch) * sizeof(Real) > 144 * 1024) b /= 2;
    return b;
}
// inclusive scan of n records in place (which = 0: filter totals, prefix; 1: smoothing totals, suffix); `scratch` holds the
// block totals of every level (n / (B - 1) + 8 records are enough)
template <typename Real, int D>
int launch_rc_scan_blocked(pgps_ctx* ctx, int which, long n, Real* data, Real* scratch) {
    if (n <= 1) return PGPS_OK;
    // d = 16: the blocked kernels spill (528 / 272 B per lane in fp64, 272 / 144 B in fp32) -- not built, the driver scans
    // those totals one Kogge-Stone level per launch (kScanBlockedDimMax in pgps_internal.h; tools/scratch_gate.py)
    if constexpr (D > kScanBlockedDimMax) return PGPS_E_UNSUPPORTED_DIM;
    else {
    constexpr int BF = scan_block_rows<Real, D>(0), BS = scan_block_rows<Real, D>(1);
    const int B = which == 0 ? BF : BS;
    const int rec = which == 0 ? 3 * D * D + 2 * D : 2 * D * D + D;
    const long nblk = (n + B - 1) / B;
    const size_t shmem = (size_t)B * (rec + kPatch) * sizeof(Real);
    const dim3 grid((unsigned)nblk), blk((unsigned)B * 16);
    if (which == 0) {
        static bool attr_f = false;
        if (!attr_f) { HIPCHK(ctx, hipFuncSetAttribute((const void*)rc_scan_blk_f<Real, D, BF>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr_f = true; }
        timed_launch(ctx, PGPS_K_FILTER_REDUCE, rc_scan_blk_f<Real, D, BF>, grid, blk, (unsigned)shmem, n, data, scratch);
    } else {
        static bool attr_s = false;
        if (!attr_s) { HIPCHK(ctx, hipFuncSetAttribute((const void*)rc_scan_blk_s<Real, D, BS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr_s = true; }
        timed_launch(ctx, PGPS_K_SMOOTHER_REDUCE, rc_scan_blk_s<Real, D, BS>, grid, blk, (unsigned)shmem, n, data, scratch);
    }
    HIPCHK(ctx, hipGetLastError());
    if (nblk > 1) {
        int rcode = launch_rc_scan_blocked<Real, D>(ctx, which, nblk, scratch, scratch + nblk * rec);
        if (rcode) return rcode;
        const dim3 g((unsigned)((n + 3) / 4));
        if (which == 0) timed_launch(ctx, PGPS_K_FILTER_REDUCE, rc_ks_filter<Real, D>, g, dim3(64), 0u, n, 0L, (const Real*)data, data, 0L, (const Real*)scratch, (long)B);
        else timed_launch(ctx, PGPS_K_SMOOTHER_REDUCE, rc_ks_smoother<Real, D>, g, dim3(64), 0u, n, 0L, (const Real*)data, data, (const Real*)scratch, (long)B);
        HIPCHK(ctx, hipGetLastError());
    }
    return PGPS_OK;
    }
}

template <int D>
int launch_rc_disc(pgps_ctx* ctx, long N, const double* F, const double* Pinf, const double* ts, double t0, double* Fs,
                   double* Qs, int batch, long bs_model) {
    // steps per row: amortises the model load and the norm -- and, on a regular grid, the one transition matrix a row
    // computes (rc_discretise stores it again for every further step with the same dt)
    // -- as long as the launch still has a wave for every SIMD: at the reference's series lengths a row that takes 8 steps
    // one after the other is 8 Pade evaluations of latency (19 us at d = 6, N = 1000) on 32 of the chip's 1024 SIMDs
    const long work = N * (long)(batch > 0 ? batch : 1);
    const int per = N >= (1L << 16) ? 32 : work > (1L << 15) ? 8 : work > (1L << 14) ? 4 : work > (1L << 13) ? 2 : 1;
    const long grid = (N + 4L * per - 1) / (4L * per);
    timed_launch(ctx, PGPS_K_DISCRETISE, rc_discretise<D>, dim3((unsigned)grid, (unsigned)batch), dim3(64), 0u, N, per, F, Pinf,
                 ts, t0, Fs, Qs, bs_model);
    HIPCHK(ctx, hipGetLastError());
    return PGPS_OK;
}

// carry records of a segment from the gathered packed totals: which = 0 carry-in (filter), 1 carry-back (smoother)
template <typename Real, int D>
int launch_rc_seg_carry(pgps_ctx* ctx, int which, const Real* gathered, int rank, int nranks, int reclen, Real* out) {
    if (which == 0) hipLaunchKernelGGL((rc_seg_carry_f<Real, D>), dim3(1), dim3(64), 0, ctx->stream, gathered, rank, reclen, out);
    else hipLaunchKernelGGL((rc_seg_carry_s<Real, D>), dim3(1), dim3(64), 0, ctx->stream, gathered, rank, nranks, reclen, out);
    HIPCHK(ctx, hipGetLastError());
    return PGPS_OK;
}

}  // namespace rc
}  // namespace pgps
