// pgps_kernels.hip.h -- the parallel Kalman filter / RTS smoother scan as CDNA4 kernels.
//
// "lane-chunk" family: one lane owns a contiguous chunk of Lc time steps and keeps whole
// d x d operands in registers (d <= 6); a wavefront owns 64*Lc contiguous steps, a workgroup
// (4 wavefronts) 256*Lc.  Three launches cover filter + log-likelihood + smoother:
//
//   k_filter_reduce   lane-serial chunk aggregates (filt_extend), wavefront Kogge-Stone scan by
//                     64-wide __shfl_up, cross-wave fold through LDS  ->  per-lane workgroup-local
//                     exclusive prefixes (lpre) + one 5-tuple per workgroup (spine)
//   k_filter_apply    every workgroup folds the spine entries to its left (tree over lanes), pushes
//                     the carried (m, P) through each lane's local prefix (filt_apply), then runs
//                     the cheap lane-serial Kalman recursion: writes fms / fPs, accumulates the
//                     log-likelihood, and -- fused -- builds the smoothing elements and their
//                     per-lane aggregates, suffix-scans them and writes lsuf + sspine
//   k_smoother_apply  folds the sspine entries to its right, applies each lane's local suffix,
//                     runs the lane-serial RTS recursion backwards: writes sms / sPs; workgroup 0
//                     also reduces the per-workgroup log-likelihood partials
//
// Every prefix that contains the first element of the series has A = 0 (parallel.py:28), so the
// carried state is just (m, P): only the reductions handle full 5-tuples.
//
// Reference: pssgp/kalman/parallel.py (pkf 121-152, pks 187-196, pkfs 199-201).
#pragma once

#include <hip/hip_runtime.h>

#include "pgps_internal.h"
#include "pgps_math.h"

namespace pgps {

// In-kernel phase stamps: DIAGNOSTIC BUILD ONLY (make stamps).  Lane 0 of every workgroup stores
// s_memtime at phase boundaries into a buffer nothing else reads; the shipped library compiles
// PGPS_STAMP to nothing.
#ifdef PGPS_STAMPS
#define PGPS_STAMP(KERNEL, IDX)                                                                          \
    do {                                                                                                 \
        if (a.stamps && threadIdx.x == 0)                                                                \
            a.stamps[((long)(KERNEL) * a.nblocks + blockIdx.x) * 8 + (IDX)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define PGPS_STAMP(KERNEL, IDX) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------
// vector loads / stores of small contiguous records (16-byte accesses when the record allows)
// ---------------------------------------------------------------------------------------------
template <int BYTES>
struct VecBytes {
    static constexpr int W = (BYTES % 16 == 0) ? 16 : (BYTES % 8 == 0) ? 8 : 4;
};

template <typename T, int N>
__device__ __forceinline__ void load_rec(const T* __restrict__ p, T* out) {
    constexpr int W = VecBytes<N * sizeof(T)>::W;
    constexpr int PER = W / sizeof(T);
    if constexpr (W == 16) {
        using V = __attribute__((ext_vector_type(4))) unsigned int;
#pragma unroll
        for (int i = 0; i < N / PER; ++i) {
            V v = reinterpret_cast<const V*>(p)[i];
            T tmp[PER];
            __builtin_memcpy(tmp, &v, 16);
#pragma unroll
            for (int j = 0; j < PER; ++j) out[i * PER + j] = tmp[j];
        }
    } else if constexpr (W == 8 && sizeof(T) == 4) {
        using V = __attribute__((ext_vector_type(2))) unsigned int;
#pragma unroll
        for (int i = 0; i < N / PER; ++i) {
            V v = reinterpret_cast<const V*>(p)[i];
            T tmp[PER];
            __builtin_memcpy(tmp, &v, 8);
#pragma unroll
            for (int j = 0; j < PER; ++j) out[i * PER + j] = tmp[j];
        }
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) out[i] = p[i];
    }
}

template <typename T, int N>
__device__ __forceinline__ void store_rec(T* __restrict__ p, const T* in) {
    constexpr int W = VecBytes<N * sizeof(T)>::W;
    constexpr int PER = W / sizeof(T);
    if constexpr (W == 16) {
        using V = __attribute__((ext_vector_type(4))) unsigned int;
#pragma unroll
        for (int i = 0; i < N / PER; ++i) {
            T tmp[PER];
#pragma unroll
            for (int j = 0; j < PER; ++j) tmp[j] = in[i * PER + j];
            V v;
            __builtin_memcpy(&v, tmp, 16);
            reinterpret_cast<V*>(p)[i] = v;
        }
    } else if constexpr (W == 8 && sizeof(T) == 4) {
        using V = __attribute__((ext_vector_type(2))) unsigned int;
#pragma unroll
        for (int i = 0; i < N / PER; ++i) {
            T tmp[PER];
#pragma unroll
            for (int j = 0; j < PER; ++j) tmp[j] = in[i * PER + j];
            V v;
            __builtin_memcpy(&v, tmp, 8);
            reinterpret_cast<V*>(p)[i] = v;
        }
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) p[i] = in[i];
    }
}

template <typename T, int D>
__device__ __forceinline__ void sym_from_full(const T* full, T* sym) {
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = i; j < D; ++j)
            sym[symi<D>(i, j)] = (i == j) ? full[i * D + i] : T(0.5) * (full[i * D + j] + full[j * D + i]);
}

template <typename T, int D>
__device__ __forceinline__ void full_from_sym(const T* sym, T* full) {
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) full[i * D + j] = sym[symi<D>(i, j)];
}

// ---------------------------------------------------------------------------------------------
// flat views of the element structs (for shuffles, LDS and workspace I/O)
// ---------------------------------------------------------------------------------------------
template <typename T, int D>
__device__ __forceinline__ void pack(const FiltElem<T, D>& e, T* v) {
    int o = 0;
#pragma unroll
    for (int i = 0; i < D * D; ++i) v[o++] = e.A[i];
#pragma unroll
    for (int i = 0; i < D; ++i) v[o++] = e.b[i];
#pragma unroll
    for (int i = 0; i < Dim<D>::SYM; ++i) v[o++] = e.C[i];
#pragma unroll
    for (int i = 0; i < Dim<D>::SYM; ++i) v[o++] = e.J[i];
#pragma unroll
    for (int i = 0; i < D; ++i) v[o++] = e.eta[i];
}
template <typename T, int D>
__device__ __forceinline__ void unpack(const T* v, FiltElem<T, D>& e) {
    int o = 0;
#pragma unroll
    for (int i = 0; i < D * D; ++i) e.A[i] = v[o++];
#pragma unroll
    for (int i = 0; i < D; ++i) e.b[i] = v[o++];
#pragma unroll
    for (int i = 0; i < Dim<D>::SYM; ++i) e.C[i] = v[o++];
#pragma unroll
    for (int i = 0; i < Dim<D>::SYM; ++i) e.J[i] = v[o++];
#pragma unroll
    for (int i = 0; i < D; ++i) e.eta[i] = v[o++];
}
template <typename T, int D>
__device__ __forceinline__ void pack(const SmthElem<T, D>& e, T* v) {
    int o = 0;
#pragma unroll
    for (int i = 0; i < D * D; ++i) v[o++] = e.E[i];
#pragma unroll
    for (int i = 0; i < D; ++i) v[o++] = e.g[i];
#pragma unroll
    for (int i = 0; i < Dim<D>::SYM; ++i) v[o++] = e.L[i];
}
template <typename T, int D>
__device__ __forceinline__ void unpack(const T* v, SmthElem<T, D>& e) {
    int o = 0;
#pragma unroll
    for (int i = 0; i < D * D; ++i) e.E[i] = v[o++];
#pragma unroll
    for (int i = 0; i < D; ++i) e.g[i] = v[o++];
#pragma unroll
    for (int i = 0; i < Dim<D>::SYM; ++i) e.L[i] = v[o++];
}

template <typename E> struct ElemTraits;
template <typename T, int D>
struct ElemTraits<FiltElem<T, D>> {
    static constexpr int N = Dim<D>::NFILT;
    using Scalar = T;
    __device__ static __forceinline__ void identity(FiltElem<T, D>& e) { filt_identity(e); }
    // time-ordered combine: `a` covers the earlier steps
    __device__ static __forceinline__ void combine(const FiltElem<T, D>& a, const FiltElem<T, D>& b, FiltElem<T, D>& o) {
        filt_combine(a, b, o);
    }
};
template <typename T, int D>
struct ElemTraits<SmthElem<T, D>> {
    static constexpr int N = Dim<D>::NSMTH;
    using Scalar = T;
    __device__ static __forceinline__ void identity(SmthElem<T, D>& e) { smth_identity(e); }
    __device__ static __forceinline__ void combine(const SmthElem<T, D>& a, const SmthElem<T, D>& b, SmthElem<T, D>& o) {
        smth_combine(a, b, o);
    }
};

// 64-wide shuffles of one scalar; overloaded for dual numbers in pgps_grad.hip.h
__device__ __forceinline__ float wshfl_up(float x, int s) { return __shfl_up(x, s, kWave); }
__device__ __forceinline__ double wshfl_up(double x, int s) { return __shfl_up(x, s, kWave); }
__device__ __forceinline__ float wshfl_down(float x, int s) { return __shfl_down(x, s, kWave); }
__device__ __forceinline__ double wshfl_down(double x, int s) { return __shfl_down(x, s, kWave); }
__device__ __forceinline__ float wshfl_idx(float x, int l) { return __shfl(x, l, kWave); }
__device__ __forceinline__ double wshfl_idx(double x, int l) { return __shfl(x, l, kWave); }

template <typename E>
__device__ __forceinline__ E shfl_up_elem(const E& e, int s) {
    using TR = ElemTraits<E>;
    typename TR::Scalar v[TR::N];
    pack(e, v);
#pragma unroll
    for (int i = 0; i < TR::N; ++i) v[i] = wshfl_up(v[i], s);
    E r;
    unpack(v, r);
    return r;
}
template <typename E>
__device__ __forceinline__ E shfl_down_elem(const E& e, int s) {
    using TR = ElemTraits<E>;
    typename TR::Scalar v[TR::N];
    pack(e, v);
#pragma unroll
    for (int i = 0; i < TR::N; ++i) v[i] = wshfl_down(v[i], s);
    E r;
    unpack(v, r);
    return r;
}

// ---------------------------------------------------------------------------------------------
// Wave scans on DPP.  The shuffles above are ds_bpermute_b32 -- an LDS-pipeline instruction per dword and level;
// the cross-lane moves of a scan are regular, so they go out as v_mov_b32_dpp instead (a VALU move with a lane
// pattern, no LDS traffic): row_shr / row_shl : 1, 2, 4, 8 scan the four 16-lane rows, row_bcast:15 (rows 1, 3 take
// lane 15 / 47) and row_bcast:31 (rows 2, 3 take lane 31) carry the row totals forwards; wave_shr:1 / wave_shl:1 turn
// the inclusive result into the exclusive one.  There is no backward counterpart of row_bcast, so the suffix scan
// takes the two cross-row levels through two rounds of indexed shuffles (first lane of the next row / of the row after).
// Operands stay contiguous ranges of lanes at every step, so associativity is all the operator needs.
// Plain float / double elements only: dual-number elements (pgps_grad.hip.h) keep the shuffle tree.
// -DPGPS_DPP_SCAN=0 builds the shuffle trees everywhere (A/B: profiles/r03_experiments.txt).
// ---------------------------------------------------------------------------------------------
#ifndef PGPS_DPP_SCAN
#define PGPS_DPP_SCAN 1
#endif
template <typename S> struct DppScalar { static constexpr bool ok = false; };
template <> struct DppScalar<float> { static constexpr bool ok = PGPS_DPP_SCAN != 0; };
template <> struct DppScalar<double> { static constexpr bool ok = PGPS_DPP_SCAN != 0; };

constexpr int kDppRowShr = 0x110, kDppRowShl = 0x100, kDppWaveShl1 = 0x130, kDppWaveShr1 = 0x138,
              kDppBcast15 = 0x142, kDppBcast31 = 0x143;

template <int CTRL, int ROWMASK>
__device__ __forceinline__ float dpp_mov(float x) {
    const int v = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(v, v, CTRL, ROWMASK, 0xf, false));
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_mov(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROWMASK, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROWMASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL, int ROWMASK, typename E>
__device__ __forceinline__ E dpp_elem(const E& e) {
    using TR = ElemTraits<E>;
    typename TR::Scalar v[TR::N];
    pack(e, v);
#pragma unroll
    for (int i = 0; i < TR::N; ++i) v[i] = dpp_mov<CTRL, ROWMASK>(v[i]);
    E r;
    unpack(v, r);
    return r;
}
template <typename E>
__device__ __forceinline__ E shfl_idx_elem(const E& e, int l) {
    using TR = ElemTraits<E>;
    typename TR::Scalar v[TR::N];
    pack(e, v);
#pragma unroll
    for (int i = 0; i < TR::N; ++i) v[i] = wshfl_idx(v[i], l);
    E r;
    unpack(v, r);
    return r;
}

// inclusive scan over the 64 lanes of a wave (lane order = time order), in place
template <typename E, bool FORWARD>
__device__ __forceinline__ void wave_scan_inclusive(E& incl, int lane) {
    using TR = ElemTraits<E>;
    if constexpr (DppScalar<typename TR::Scalar>::ok) {
        const int r = lane & 15;
        auto step = [&](const E& other, bool act) {
            if (act) {
                E t;
                if (FORWARD) TR::combine(other, incl, t); else TR::combine(incl, other, t);
                incl = t;
            }
        };
        if constexpr (FORWARD) {
            step(dpp_elem<kDppRowShr + 1, 0xf>(incl), r >= 1);
            step(dpp_elem<kDppRowShr + 2, 0xf>(incl), r >= 2);
            step(dpp_elem<kDppRowShr + 4, 0xf>(incl), r >= 4);
            step(dpp_elem<kDppRowShr + 8, 0xf>(incl), r >= 8);
            step(dpp_elem<kDppBcast15, 0xa>(incl), (lane & 16) != 0);
            step(dpp_elem<kDppBcast31, 0xc>(incl), lane >= 32);
        } else {
            step(dpp_elem<kDppRowShl + 1, 0xf>(incl), r + 1 < 16);
            step(dpp_elem<kDppRowShl + 2, 0xf>(incl), r + 2 < 16);
            step(dpp_elem<kDppRowShl + 4, 0xf>(incl), r + 4 < 16);
            step(dpp_elem<kDppRowShl + 8, 0xf>(incl), r + 8 < 16);
            const int row = lane >> 4;
            step(shfl_idx_elem(incl, ((row + 1) & 3) * 16), row < 3);      // total of the next row
            step(shfl_idx_elem(incl, ((row + 2) & 3) * 16), row < 2);      // rows 2.. (row 0) / row 3 (row 1)
        }
    } else {
#pragma unroll
        for (int s = 1; s < kWave; s <<= 1) {
            E other = FORWARD ? shfl_up_elem(incl, s) : shfl_down_elem(incl, s);
            const bool act = FORWARD ? (lane >= s) : (lane + s < kWave);
            if (act) {
                E t;
                if (FORWARD) TR::combine(other, incl, t); else TR::combine(incl, other, t);
                incl = t;
            }
        }
    }
}
// the neighbour's value: lane - 1 (FORWARD) or lane + 1; the first / last lane gets something it must not use
template <typename E, bool FORWARD>
__device__ __forceinline__ E wave_shift1(const E& e) {
    using TR = ElemTraits<E>;
    if constexpr (DppScalar<typename TR::Scalar>::ok) {
        if constexpr (FORWARD) return dpp_elem<kDppWaveShr1, 0xf>(e);
        else return dpp_elem<kDppWaveShl1, 0xf>(e);
    } else {
        return FORWARD ? shfl_up_elem(e, 1) : shfl_down_elem(e, 1);
    }
}

// strided (field-major) workspace I/O: value f of lane-slot t lives at ws[f * stride + t]
template <typename E>
__device__ __forceinline__ void ws_store(typename ElemTraits<E>::Scalar* ws, long stride, long t, const E& e) {
    using TR = ElemTraits<E>;
    typename TR::Scalar v[TR::N];
    pack(e, v);
#pragma unroll
    for (int i = 0; i < TR::N; ++i) ws[i * stride + t] = v[i];
}
template <typename E>
__device__ __forceinline__ void ws_load(const typename ElemTraits<E>::Scalar* ws, long stride, long t, E& e) {
    using TR = ElemTraits<E>;
    typename TR::Scalar v[TR::N];
#pragma unroll
    for (int i = 0; i < TR::N; ++i) v[i] = ws[i * stride + t];
    unpack(v, e);
}
// record-major (AoS) I/O for spine entries
template <typename E>
__device__ __forceinline__ void rec_store(typename ElemTraits<E>::Scalar* p, const E& e) {
    using TR = ElemTraits<E>;
    typename TR::Scalar v[TR::N];
    pack(e, v);
#pragma unroll
    for (int i = 0; i < TR::N; ++i) p[i] = v[i];
}
template <typename E>
__device__ __forceinline__ void rec_load(const typename ElemTraits<E>::Scalar* p, E& e) {
    using TR = ElemTraits<E>;
    typename TR::Scalar v[TR::N];
#pragma unroll
    for (int i = 0; i < TR::N; ++i) v[i] = p[i];
    unpack(v, e);
}

// ---------------------------------------------------------------------------------------------
// workgroup scans.  FORWARD: exclusive prefix over lanes (time order = lane order);
// BACKWARD: exclusive suffix.  `lds` holds kWaves * N scalars.  Returns the exclusive value for
// this lane in `excl` and the workgroup total in `total` (valid in every lane).
// ---------------------------------------------------------------------------------------------
template <typename E, bool FORWARD>
__device__ __forceinline__ void block_scan_exclusive(const E& mine, E& excl, E& total,
                                                     typename ElemTraits<E>::Scalar* lds) {
    using TR = ElemTraits<E>;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    E incl = mine;
    wave_scan_inclusive<E, FORWARD>(incl, lane);
    E wex = wave_shift1<E, FORWARD>(incl);
    if (FORWARD ? (lane == 0) : (lane == kWave - 1)) TR::identity(wex);
    // wavefront totals to LDS
    if (FORWARD ? (lane == kWave - 1) : (lane == 0)) {
        typename TR::Scalar v[TR::N];
        pack(incl, v);
#pragma unroll
        for (int i = 0; i < TR::N; ++i) lds[wave * TR::N + i] = v[i];
    }
    __syncthreads();
    E acc;
    TR::identity(acc);
    excl = wex;
    bool have = false;
    if (FORWARD) {
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
            E wt;
            rec_load(lds + w * TR::N, wt);
            if (w == wave && have) { E r; TR::combine(acc, wex, r); excl = r; }
            if (have) { E r; TR::combine(acc, wt, r); acc = r; } else { acc = wt; have = true; }
        }
    } else {
#pragma unroll
        for (int w = kWaves - 1; w >= 0; --w) {
            E wt;
            rec_load(lds + w * TR::N, wt);
            if (w == wave && have) { E r; TR::combine(wex, acc, r); excl = r; }
            if (have) { E r; TR::combine(wt, acc, r); acc = r; } else { acc = wt; have = true; }
        }
    }
    total = acc;
    __syncthreads();
}

// Ordered reduction of one element per lane over the whole workgroup (lane order = time order).
// The result is valid in every lane.
template <typename E>
__device__ __forceinline__ void block_reduce_ordered(const E& mine, E& total, typename ElemTraits<E>::Scalar* lds) {
    using TR = ElemTraits<E>;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    E acc = mine;
    int src_lane = 0;
    if constexpr (DppScalar<typename TR::Scalar>::ok) {
        // the forward DPP scan leaves the wave's total in its last lane
        wave_scan_inclusive<E, true>(acc, lane);
        src_lane = kWave - 1;
    } else {
#pragma unroll
        for (int s = 1; s < kWave; s <<= 1) {
            E other = shfl_down_elem(acc, s);
            if ((lane & (2 * s - 1)) == 0) { E r; TR::combine(acc, other, r); acc = r; }
        }
    }
    if (lane == src_lane) {
        typename TR::Scalar v[TR::N];
        pack(acc, v);
#pragma unroll
        for (int i = 0; i < TR::N; ++i) lds[wave * TR::N + i] = v[i];
    }
    __syncthreads();
    rec_load(lds, total);
#pragma unroll
    for (int w = 1; w < kWaves; ++w) {
        E wt, r;
        rec_load(lds + w * TR::N, wt);
        TR::combine(total, wt, r);
        total = r;
    }
    __syncthreads();
}

__device__ __forceinline__ double block_sum_double(double x, double* lds) {
#pragma unroll
    for (int s = kWave / 2; s > 0; s >>= 1) x += __shfl_down(x, s, kWave);
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    if (lane == 0) lds[wave] = x;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) t += lds[w];
    __syncthreads();
    return t;
}

// The workgroup that draws the last ticket sums every workgroup's log-likelihood partial -- in a fixed order, so the result
// does not depend on which workgroup that is -- and writes the total: the filter-only calls need no finalize launch of
// their own (5 us of kernel and a launch boundary on a 37 us call at c1's length).  `ticket` is a device word that is zero
// between calls (word 16 of the context's status buffer); the summing workgroup resets it.  One call in flight per context.
constexpr int kLlTicketWord = 16;
__device__ __forceinline__ void ll_finish(const double* llpart, int nblocks, double* ll, int* ticket, double* lds) {
    __shared__ int s_last;
    if (threadIdx.x == 0) {
        __threadfence();                            // this workgroup's partial before its ticket
        s_last = (atomicAdd(ticket, 1) == (int)gridDim.x - 1) ? 1 : 0;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    double v = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += (int)blockDim.x)
        v += __hip_atomic_load(&llpart[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double t = block_sum_double(v, lds);
    if (threadIdx.x == 0) {
        *ll = t;
        __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---------------------------------------------------------------------------------------------
// Wavefront-private LDS staging: coalesced global <-> LDS, lane-owned records out of LDS.
//
// A lane owns Lc consecutive steps, so its records sit Lc*RECB bytes apart from its neighbour's:
// reading them straight from global memory touches 64 different 128-byte lines per wave
// instruction and uses 16 bytes of each (measured: the smoother pass ran at 1.7 TB/s of
// algorithmic bytes that way, L2-request bound).  Instead the wave streams through its span in
// sub-tiles of G steps per lane: piece q (16 bytes) of the sub-tile belongs to lane q / NV and is
// loaded by lane q % 64 of instruction q / 64, so consecutive lanes read consecutive 16-byte
// pieces of one lane-segment (SEG = G*RECB contiguous bytes: whole 128-byte lines for G = 4,
// d = 2, fp64), the pieces are parked in LDS at owner*STRIDE (+16 bytes of padding per owner so
// the owners' ds_read_b128 of their own records are bank-conflict free) and each lane then reads
// its own G records from LDS.  Outputs take the same road back.  The next sub-tile's global loads
// are issued into registers before the current one is computed (one sub-tile of prefetch).
// Everything is wave-private: no workgroup barrier inside the streaming loops.
// ---------------------------------------------------------------------------------------------
using V4 = __attribute__((ext_vector_type(4))) unsigned int;

template <int RECB, int G>
struct StageGeom {
    static constexpr int SEG = RECB * G;
    static_assert(SEG % 16 == 0, "lane segment must be a whole number of 16-byte pieces");
    static constexpr int NV = SEG / 16;
    static constexpr int STRIDE = SEG + 16;
    static constexpr int BYTES = kWave * STRIDE;
};

// Ordering of a wave's OWN LDS accesses (the staging buffers are wave-private: one lane writes what another lane of the same
// wave reads next).  The LDS executes a wave's instructions in issue order, so only the compiler has to be held to program
// order: fences at WAVEFRONT scope.  Until round 5 these were workgroup-scope fences, which on gfx950 also drain every
// vector-memory operation of the wave (s_waitcnt vmcnt(0) in front of each): every sub-tile waited for its own prefetch and
// for the previous sub-tile's output stores.  -DPGPS_LDS_SYNC_WORKGROUP restores that (A/B: profiles/r05_experiments.txt).
__device__ __forceinline__ void wave_lds_sync() {
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

template <typename GEO>
__device__ __forceinline__ void stage_issue(const char* __restrict__ g, long lane_pitch, V4* r) {
    const int lane = threadIdx.x & (kWave - 1);
#pragma unroll
    for (int v = 0; v < GEO::NV; ++v) {
        const int q = v * kWave + lane;
        r[v] = *reinterpret_cast<const V4*>(g + (long)(q / GEO::NV) * lane_pitch + (q % GEO::NV) * 16);
    }
}
template <typename GEO>
__device__ __forceinline__ void stage_commit(char* lds, const V4* r) {
    const int lane = threadIdx.x & (kWave - 1);
#pragma unroll
    for (int v = 0; v < GEO::NV; ++v) {
        const int q = v * kWave + lane;
        *reinterpret_cast<V4*>(lds + (q / GEO::NV) * GEO::STRIDE + (q % GEO::NV) * 16) = r[v];
    }
}
// nt = streaming (non-temporal) stores: worth 12 % on the smoother pass once a pass no longer fits
// the 256 MiB Infinity Cache (N = 2^24), slightly harmful when it does (N = 2^20) -- chosen per call
// LSTRIDE: distance between the owners' segments in LDS (another geometry's when the records live inside its buffer)
template <typename GEO, bool NT, int LSTRIDE = GEO::STRIDE>
__device__ __forceinline__ void stage_drain(char* __restrict__ g, long lane_pitch, const char* lds) {
    const int lane = threadIdx.x & (kWave - 1);
#pragma unroll
    for (int v = 0; v < GEO::NV; ++v) {
        const int q = v * kWave + lane;
        const V4 x = *reinterpret_cast<const V4*>(lds + (q / GEO::NV) * LSTRIDE + (q % GEO::NV) * 16);
        V4* dst = reinterpret_cast<V4*>(g + (long)(q / GEO::NV) * lane_pitch + (q % GEO::NV) * 16);
        if constexpr (NT) __builtin_nontemporal_store(x, dst);
        else *dst = x;
    }
}
// this lane's record i of the current sub-tile
template <typename GEO, typename T, int N>
__device__ __forceinline__ void stage_get(const char* lds, int i, T* out) {
    const int lane = threadIdx.x & (kWave - 1);
    load_rec<T, N>(reinterpret_cast<const T*>(lds + lane * GEO::STRIDE) + i * N, out);
}
template <typename GEO, typename T, int N, int LSTRIDE = GEO::STRIDE>
__device__ __forceinline__ void stage_put(char* lds, int i, const T* in) {
    const int lane = threadIdx.x & (kWave - 1);
    store_rec<T, N>(reinterpret_cast<T*>(lds + lane * LSTRIDE) + i * N, in);
}
// The G observations of a lane's sub-tile come straight from global memory (G * sizeof(T) contiguous bytes per lane,
// one or two 16-byte loads): they are 8 of the 72 bytes a step moves, and without a transposition buffer for them
// (and with the filtered means leaving through Q's buffer, below) two workgroups fit the LDS of a CU.
template <typename T, int G>
__device__ __forceinline__ void y_issue(const T* __restrict__ p, T* out) { load_rec<T, G>(p, out); }

// staging is used when a lane's sub-tile fits the LDS budget: d <= 2, d = 3 with 144-byte lane segments (fp64 G = 2,
// fp32 G = 4), and -- one step per sub-tile -- d = 4 (64- / 128-byte records) and d = 6 in fp32 (144-byte records:
// config c3, where the direct accesses cost 40 % of k_filter_apply: 0.237 ms, 0.184 without its loads, 0.187 without
// its stores); G = steps per lane per sub-tile
template <typename T, int D, int G>
struct StageCfg {
    static constexpr int W = (int)sizeof(T);
    static constexpr bool on = (G > 0) && (D <= 2 || (D == 3 && G * W == 16) || (D == 4 && G == 1) || (D == 6 && W == 4 && G == 1));
    static constexpr int GG = on ? G : 4;
    // the means travel through LDS too when a lane's G of them are whole 16-byte pieces; otherwise (d = 6 fp32: 24
    // bytes) they are read and written directly -- 24 of the 456 bytes a step moves
    static constexpr bool stage_m = (D * W * GG) % 16 == 0;
    using GF = StageGeom<D * D * W, GG>;                                        // F, Q, P records
    using GM = StageGeom<stage_m ? D * W : 16, stage_m ? GG : 1>;               // m records (placeholder when direct)
    using GY = StageGeom<W, GG>;                                                // y records
    // filter kernels: F (in; filtered P out) and Q (in; filtered m out, written compactly into the lane's own
    // segment once Q_i has been read: m_i ends before Q_{i+1} begins) -- 2 x 9 KiB per wave at d = 2 fp64
    static constexpr int F1_BYTES = on ? 2 * GF::BYTES : 16;
    static constexpr int F3_BYTES = on ? 2 * GF::BYTES : 16;
    static constexpr int S3_BYTES = on ? 3 * GF::BYTES + (stage_m ? GM::BYTES : 0) : 16;
};

// Fold spine entries [lo, hi) in time order over the whole workgroup; result in every lane.
// Split in two so that a caller can put other global loads between the spine loads and the
// arithmetic (loads retire in issue order: whatever is issued first is waited for first).
template <typename E>
__device__ __forceinline__ void fold_spine_partial(const typename ElemTraits<E>::Scalar* spine, int lo, int hi, E& acc) {
    using TR = ElemTraits<E>;
    const int n = hi - lo;
    const int per = (n + kBlock - 1) / kBlock;
    TR::identity(acc);
    const int b0 = lo + (int)threadIdx.x * per;
    const int b1 = min(hi, b0 + per);
    if (per <= 2) {
        // the common geometries (<= 2 entries per lane): both records are requested before either is used, so the
        // fold pays one round trip to memory instead of two
        E e0, e1;
        const bool h0 = b0 < b1, h1 = b0 + 1 < b1;
        if (h0) rec_load(spine + (long)b0 * TR::N, e0);
        if (h1) rec_load(spine + (long)(b0 + 1) * TR::N, e1);
        if (h1) TR::combine(e0, e1, acc);
        else if (h0) acc = e0;
        return;
    }
    bool have = false;
    for (int b = b0; b < b1; ++b) {
        E e;
        rec_load(spine + (long)b * TR::N, e);
        if (have) { E r; TR::combine(acc, e, r); acc = r; } else { acc = e; have = true; }
    }
}

template <typename E>
__device__ __forceinline__ void fold_spine(const typename ElemTraits<E>::Scalar* spine, int lo, int hi, E& total,
                                           typename ElemTraits<E>::Scalar* lds) {
    E acc;
    fold_spine_partial<E>(spine, lo, hi, acc);
    block_reduce_ordered(acc, total, lds);
}

// ---------------------------------------------------------------------------------------------
// The forgetting shortcut for the carry across workgroups.
//
// A workgroup's carry-in is the prefix T_0 (x) ... (x) T_{b-1} of the workgroup totals applied to the prior.  In ANY bracketing
// the prefix's (b, C) are those of the LAST total whenever that total's A vanishes: out.b = A_2 w + b_2, out.C = A_2 N A_2^T
// + C_2 (filtering operator, parallel.py:100-118).  A total over a few thousand steps of a filter that forgets has |A| far
// below anything a sum with b or C can see (config c2: 0.917^4096 = 1e-154 from F alone), so the left neighbour's record IS
// the carry and the fold of every total to the left -- ten combine levels: 10 % of a c2 pass, two thirds of the instructions of
// a c3 Kalman-pass launch -- is skipped.  The smoother has the mirror image: a total whose E vanishes (a product of
// thousands of smoother gains) hands the workgroup before it its own (g, L) (parallel.py:176-184).
// Decided per workgroup FROM THE DATA: max |A| (|E|) <= 2^-120 in fp64, 2^-60 in fp32 -- thirty orders of magnitude below the
// unit round-off, NaN fails the test -- the same in every lane (they read the same words); otherwise the general fold runs.
// Where the shortcut applies both roads give the same bits.  Tried only where a workgroup spans >= 2048 steps
// (ScanArgs::shortcut, set by the launch code; pgps_set_shortcut(ctx, 0) turns it off: the tests run both roads).
// ---------------------------------------------------------------------------------------------
template <typename T> struct Forget;
template <> struct Forget<double> { static constexpr double kA = 0x1p-120; };
template <> struct Forget<float> { static constexpr float kA = 0x1p-60f; };

template <typename T, int N>
__device__ __forceinline__ bool forget_test(const T* __restrict__ rec) {
    T x[N];
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = rec[i];
    T amax = T(0);
#pragma unroll
    for (int i = 0; i < N; ++i) amax = fmax(amax, fabs(x[i]));
    return __builtin_amdgcn_readfirstlane((int)(amax <= Forget<T>::kA)) != 0;
}
// filter: true and s = (b, C) of workgroup `nb`'s total when that total has forgotten what came before it
template <typename T, int D>
__device__ __forceinline__ bool carry_shortcut_filter(const T* __restrict__ spine, int nb, MeanCov<T, D>& s) {
    constexpr int MAT = D * D, SYM = Dim<D>::SYM, NF = Dim<D>::NFILT;
    const T* rec = spine + (long)nb * NF;
    if (!forget_test<T, MAT>(rec)) return false;
#pragma unroll
    for (int i = 0; i < D; ++i) s.m[i] = rec[MAT + i];
#pragma unroll
    for (int i = 0; i < SYM; ++i) s.P[i] = rec[MAT + D + i];
    return true;
}
// smoother: true and s = (g, L) of workgroup `nb`'s total
template <typename T, int D>
__device__ __forceinline__ bool carry_shortcut_smoother(const T* __restrict__ sspine, int nb, MeanCov<T, D>& s) {
    constexpr int MAT = D * D, SYM = Dim<D>::SYM, NS = Dim<D>::NSMTH;
    const T* rec = sspine + (long)nb * NS;
    if (!forget_test<T, MAT>(rec)) return false;
#pragma unroll
    for (int i = 0; i < D; ++i) s.m[i] = rec[MAT + i];
#pragma unroll
    for (int i = 0; i < SYM; ++i) s.P[i] = rec[MAT + D + i];
    return true;
}

// ---------------------------------------------------------------------------------------------
// K-F1: filter reduce
// ---------------------------------------------------------------------------------------------
template <typename T, int D>
__device__ __forceinline__ void filter_reduce_step(const ScanArgs<T>& a, long k, const T* F, const T* Qf, T y,
                                                   const T* h, FiltElem<T, D>& agg) {
    constexpr int MAT = D * D, SYM = Dim<D>::SYM;
    if (k == 0 && a.seg_first) {
        T P0f[MAT], P0[SYM];
#pragma unroll
        for (int i = 0; i < MAT; ++i) P0f[i] = a.P0[i];
        sym_from_full<T, D>(P0f, P0);
        filt_first(agg, P0, y, h, a.R);
    } else {
        T Q[SYM];
        sym_from_full<T, D>(Qf, Q);
        filt_extend(agg, F, Q, y, h, a.R);
    }
}

template <typename T, int D>
__device__ __forceinline__ void lane_filter_reduce_direct(const ScanArgs<T>& a, long k0, long k1, const T* h,
                                                          FiltElem<T, D>& agg) {
    constexpr int MAT = D * D;
    if (k0 >= k1) return;
    T Fn[MAT], Qn[MAT];
    T yn;
    load_rec<T, MAT>(a.Fs + k0 * MAT, Fn);
    load_rec<T, MAT>(a.Qs + k0 * MAT, Qn);
    yn = a.ys[k0];
    for (long k = k0; k < k1; ++k) {
        T F[MAT], Qf[MAT];
#pragma unroll
        for (int i = 0; i < MAT; ++i) { F[i] = Fn[i]; Qf[i] = Qn[i]; }
        const T y = yn;
        if (k + 1 < k1) {
            load_rec<T, MAT>(a.Fs + (k + 1) * MAT, Fn);
            load_rec<T, MAT>(a.Qs + (k + 1) * MAT, Qn);
            yn = a.ys[k + 1];
        }
        filter_reduce_step<T, D>(a, k, F, Qf, y, h, agg);
    }
}

template <typename T, int D, int G>
__device__ __forceinline__ void lane_filter_reduce_staged(const ScanArgs<T>& a, long wbase, char* lds, const T* h,
                                                          FiltElem<T, D>& agg) {
    using CFG = StageCfg<T, D, G>;
    using GF = typename CFG::GF;
    constexpr int MAT = D * D;
    const int lane = threadIdx.x & (kWave - 1);
    char* lF = lds;
    char* lQ = lF + GF::BYTES;
    const long pitchF = (long)a.Lc * MAT * sizeof(T);
    const char* gF = reinterpret_cast<const char*>(a.Fs + wbase * MAT);
    const char* gQ = reinterpret_cast<const char*>(a.Qs + wbase * MAT);
    const T* gY = a.ys + wbase + (long)lane * a.Lc;
    const int S = a.Lc / G;
    V4 rF[GF::NV], rQ[GF::NV];
    T yn[G];
    stage_issue<GF>(gF, pitchF, rF);
    stage_issue<GF>(gQ, pitchF, rQ);
    y_issue<T, G>(gY, yn);
    for (int s = 0; s < S; ++s) {
        wave_lds_sync();
        stage_commit<GF>(lF, rF);
        stage_commit<GF>(lQ, rQ);
        T yv[G];
#pragma unroll
        for (int i = 0; i < G; ++i) yv[i] = yn[i];
        if (s + 1 < S) {
            stage_issue<GF>(gF + (long)(s + 1) * GF::SEG, pitchF, rF);
            stage_issue<GF>(gQ + (long)(s + 1) * GF::SEG, pitchF, rQ);
            y_issue<T, G>(gY + (s + 1) * G, yn);
        }
        wave_lds_sync();
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const long k = wbase + (long)lane * a.Lc + s * G + i;
            T F[MAT], Qf[MAT];
            stage_get<GF, T, MAT>(lF, i, F);
            stage_get<GF, T, MAT>(lQ, i, Qf);
            filter_reduce_step<T, D>(a, k, F, Qf, yv[i], h, agg);
        }
    }
}

template <typename T, int D, int G>
__global__ __launch_bounds__(kBlock) void k_filter_reduce(const ScanArgs<T> a) {
    using FE = FiltElem<T, D>;
    using CFG = StageCfg<T, D, G>;
    __shared__ T lds[kWaves * Dim<D>::NFILT];
    __shared__ __attribute__((aligned(16))) char stage[kWaves][CFG::F1_BYTES];

    T h[D];
#pragma unroll
    for (int i = 0; i < D; ++i) h[i] = a.H[i];

    const long gt = (long)blockIdx.x * kBlock + threadIdx.x;
    const long k0 = gt * a.Lc;
    const long k1 = min(a.N, k0 + a.Lc);
    const int wave = threadIdx.x / kWave;
    const long wbase = ((long)blockIdx.x * kBlock + wave * kWave) * a.Lc;

    FE agg;
    filt_identity(agg);
    PGPS_STAMP(0, 0);
    bool staged = false;
    if constexpr (CFG::on) {
        staged = (wbase + (long)kWave * a.Lc <= a.N) && (a.Lc % G == 0);
        if (staged) lane_filter_reduce_staged<T, D, G>(a, wbase, stage[wave], h, agg);
    }
    if (!staged) lane_filter_reduce_direct<T, D>(a, k0, k1, h, agg);
    PGPS_STAMP(0, 1);

    FE excl, total;
    block_scan_exclusive<FE, true>(agg, excl, total, lds);
    PGPS_STAMP(0, 2);
    ws_store(a.lpre, a.nlanes, gt, excl);
    if (threadIdx.x == 0) rec_store(a.spine + (long)blockIdx.x * Dim<D>::NFILT, total);
    PGPS_STAMP(0, 3);
}

// ---------------------------------------------------------------------------------------------
// K-F3: filter apply (+ log-likelihood, + fused smoothing-aggregate build when SMOOTH)
// ---------------------------------------------------------------------------------------------
// one step of the lane-serial Kalman pass; `prev` = filtered state of step k-1 (or the carry-in)
// DFORM: the smoothing total is kept in innovation form (pgps_math.h smth_extend_u): no L = P - E (F P) product and a rank-one
// fold instead of two matrix products; the smoother then adds the filtered moments of the step a total is applied at
// (ScanArgs::dform).  Whole-series pkfs only: the segment protocol exchanges totals in the reference's (E, g, L) form.
template <typename T, int D, bool SMOOTH, typename LL = LogLik, bool DFORM = false>
__device__ __forceinline__ void filter_apply_step(const ScanArgs<T>& a, long k, long k0, const T* F, const T* Qf, T y,
                                                  const T* h, MeanCov<T, D>& s, LL& ll, SmthElem<T, D>& sagg) {
    constexpr int MAT = D * D, SYM = Dim<D>::SYM;
    T Q[SYM];
    sym_from_full<T, D>(Qf, Q);
    T mp[D], Pp[SYM], FP[MAT];
    if constexpr (DFORM && SMOOTH) {
        T u[D], inv, res;
        kf_step_u(s, F, Q, y, h, a.R, (k == 0) && a.seg_first, ll, mp, Pp, FP, u, inv, res);
        if (k > k0) {                       // element of step k-1: its gain, and THIS step's update
            T E[MAT];
            smth_gain<T, D>(FP, Pp, E);
            smth_extend_u(sagg, E, u, inv, res);
        }
    } else {
        MeanCov<T, D> prev = s;
        kf_step(s, F, Q, y, h, a.R, (k == 0) && a.seg_first, ll, mp, Pp, FP);
        if (SMOOTH && k > k0) {                 // element of step k-1 from this step's predict
            SmthElem<T, D> e, r;
            smth_element(prev, mp, Pp, FP, e);
            smth_combine(sagg, e, r);
            sagg = r;
        }
    }
}

// (F, Q) of the step after a chunk's last step: the next chunk's first step, the next segment's
// first step (halo), or nothing at the end of the series.  Returns false in the last case.
template <typename T, int D>
__device__ __forceinline__ bool filter_tail_load(const ScanArgs<T>& a, long k1, T* F, T* Qf) {
    constexpr int MAT = D * D;
    if (k1 < a.N) {
        load_rec<T, MAT>(a.Fs + k1 * MAT, F);
        load_rec<T, MAT>(a.Qs + k1 * MAT, Qf);
        return true;
    }
    if (!a.seg_last) {
#pragma unroll
        for (int i = 0; i < MAT; ++i) { F[i] = a.halo_FQ[i]; Qf[i] = a.halo_FQ[MAT + i]; }
        return true;
    }
#pragma unroll
    for (int i = 0; i < MAT; ++i) { F[i] = T(0); Qf[i] = T(0); }
    return false;
}

// smoothing element of a chunk's last step from the predict of the step after the chunk
template <typename T, int D>
__device__ __forceinline__ void filter_tail_apply(bool have_next, const T* F, const T* Qf, const MeanCov<T, D>& s,
                                                  SmthElem<T, D>& sagg) {
    constexpr int MAT = D * D, SYM = Dim<D>::SYM;
    SmthElem<T, D> e, r;
    if (have_next) {
        T Q[SYM];
        sym_from_full<T, D>(Qf, Q);
        T mp[D], Pp[SYM], FP[MAT];
        mat_vec<T, D>(F, s.m, mp);
        predict_cov<T, D>(F, s.P, Q, FP, Pp);
        smth_element(s, mp, Pp, FP, e);
    } else {
        smth_last(s, e);
    }
    smth_combine(sagg, e, r);
    sagg = r;
}

// the same in innovation form: the step after the chunk contributes its predict AND its update (y of that step)
template <typename T, int D>
__device__ __forceinline__ void filter_tail_apply_u(bool have_next, const T* F, const T* Qf, T yn, const T* h, T R,
                                                    const MeanCov<T, D>& s, SmthElem<T, D>& sagg) {
    constexpr int MAT = D * D, SYM = Dim<D>::SYM;
    if (!have_next) { smth_extend_last_u(sagg); return; }
    T Q[SYM];
    sym_from_full<T, D>(Qf, Q);
    T mp[D], Pp[SYM], FP[MAT], u[D], E[MAT];
    mat_vec<T, D>(F, s.m, mp);
    predict_cov<T, D>(F, s.P, Q, FP, Pp);
    sym_vec<T, D>(Pp, h, u);
    T S = R, mu = T(0);
#pragma unroll
    for (int i = 0; i < D; ++i) { S += h[i] * u[i]; mu += h[i] * mp[i]; }
    const bool obs = !is_nan(yn);
    const T inv = obs ? recip(S) : T(0);
    const T res = obs ? yn - mu : T(0);
    smth_gain<T, D>(FP, Pp, E);
    smth_extend_u(sagg, E, u, inv, res);
}

template <typename T, int D>
__device__ __forceinline__ void filter_apply_tail(const ScanArgs<T>& a, long k1, const MeanCov<T, D>& s,
                                                  SmthElem<T, D>& sagg) {
    T F[D * D], Qf[D * D];
    const bool have_next = filter_tail_load<T, D>(a, k1, F, Qf);
    filter_tail_apply<T, D>(have_next, F, Qf, s, sagg);
}

template <typename T, int D, bool SMOOTH, bool DFORM = false>
__device__ __forceinline__ void lane_filter_apply_direct(const ScanArgs<T>& a, long k0, long k1, const T* h,
                                                         MeanCov<T, D>& s, LogLik& ll, SmthElem<T, D>& sagg) {
    constexpr int MAT = D * D;
    if (k0 >= k1) return;
    T Fn[MAT], Qn[MAT];
    T yn;
    load_rec<T, MAT>(a.Fs + k0 * MAT, Fn);
    load_rec<T, MAT>(a.Qs + k0 * MAT, Qn);
    yn = a.ys[k0];
    for (long k = k0; k < k1; ++k) {
        T F[MAT], Qf[MAT];
#pragma unroll
        for (int i = 0; i < MAT; ++i) { F[i] = Fn[i]; Qf[i] = Qn[i]; }
        const T y = yn;
        if (k + 1 < k1) {
            load_rec<T, MAT>(a.Fs + (k + 1) * MAT, Fn);
            load_rec<T, MAT>(a.Qs + (k + 1) * MAT, Qn);
            yn = a.ys[k + 1];
        }
        filter_apply_step<T, D, SMOOTH, LogLik, DFORM>(a, k, k0, F, Qf, y, h, s, ll, sagg);
        store_rec<T, D>(a.fms + k * D, s.m);
        T Pf[MAT];
        full_from_sym<T, D>(s.P, Pf);
        store_rec<T, MAT>(a.fPs + k * MAT, Pf);
    }
    if constexpr (SMOOTH && DFORM) {
        T Fh[MAT], Qh[MAT];
        const bool have_next = filter_tail_load<T, D>(a, k1, Fh, Qh);
        const T yh = (k1 < a.N) ? a.ys[k1] : T(0);
        filter_tail_apply_u<T, D>(have_next, Fh, Qh, yh, h, a.R, s, sagg);
    } else if (SMOOTH) {
        filter_apply_tail<T, D>(a, k1, s, sagg);
    }
}

// Staged lane-serial Kalman pass.  prefetch() issues the first sub-tile's global loads (and the
// halo step's) into registers; it is called BEFORE the workgroup folds the spine so that the
// memory latency hides behind the fold's arithmetic.  run() streams the wave's span.
template <typename T, int D, bool SMOOTH, int G, bool NT, bool DFORM = false>
struct FilterApplyStaged {
    using CFG = StageCfg<T, D, G>;
    using GF = typename CFG::GF;
    using GM = typename CFG::GM;
    static constexpr int MAT = D * D;
    V4 rF[GF::NV], rQ[GF::NV];
    T yn[G];
    T Fh[MAT], Qh[MAT];
    T yh;                               // DFORM: the observation of the step after the chunk
    bool have_next;

    __device__ __forceinline__ void prefetch(const ScanArgs<T>& a, long wbase) {
        const int lane = threadIdx.x & (kWave - 1);
        const long pitchF = (long)a.Lc * MAT * sizeof(T);
        // No register prefetch of the first sub-tile here (nor of the next one in run()): 72 registers less put the kernel
        // at 206 VGPRs, two waves per SIMD, and the second wave covers the latency instead -- equal at 2^20 steps (one
        // wave per SIMD's worth of work), 5-10 % faster from 2^21 on (2^24: 0.52 -> 0.47 ms)
        // the halo step of lane l is the first step of lane l+1: after the first sub-tile is in LDS
        // it is fetched from there (run()); only the wave's last lane reads global memory
        have_next = false;
        yh = T(0);
        if (SMOOTH && lane == kWave - 1) {
            const long kn = wbase + (long)kWave * a.Lc;
            have_next = filter_tail_load<T, D>(a, kn, Fh, Qh);
            if (DFORM && kn < a.N) yh = a.ys[kn];
        }
    }

    __device__ __forceinline__ void run(const ScanArgs<T>& a, long wbase, char* lds, const T* h, MeanCov<T, D>& s,
                                        LogLik& ll, SmthElem<T, D>& sagg) {
        const int lane = threadIdx.x & (kWave - 1);
        char* lF = lds;                     // F in, P out
        char* lQ = lF + GF::BYTES;          // Q in, m out (compact, inside the lane's own segment)
        const long pitchF = (long)a.Lc * MAT * sizeof(T), pitchM = (long)a.Lc * D * sizeof(T);
        const char* gF = reinterpret_cast<const char*>(a.Fs + wbase * MAT);
        const char* gQ = reinterpret_cast<const char*>(a.Qs + wbase * MAT);
        const T* gY = a.ys + wbase + (long)lane * a.Lc;
        char* gP = reinterpret_cast<char*>(a.fPs + wbase * MAT);
        char* gM = reinterpret_cast<char*>(a.fms + wbase * D);
        const int S = a.Lc / G;
        const long k0 = wbase + (long)lane * a.Lc;
        // The 128-lane build runs one workgroup of two waves per CU at 2^20 steps: no second wave covers the loads, and
        // registers do not decide its occupancy -- so there the next sub-tile is requested as soon as this one's pieces
        // have been committed (kPrefetch).  The 256-lane build (long series, two waves per SIMD) keeps the lean loop.
#ifdef PGPS_NARROW
        constexpr bool kPrefetch = true;
#else
        constexpr bool kPrefetch = false;
#endif
        auto issue = [&](int sb) {
            stage_issue<GF>(gF + (long)sb * GF::SEG, pitchF, rF);
            stage_issue<GF>(gQ + (long)sb * GF::SEG, pitchF, rQ);
            y_issue<T, G>(gY + sb * G, yn);
        };
        if (kPrefetch) issue(0);
        for (int sb = 0; sb < S; ++sb) {
            wave_lds_sync();
            if (!kPrefetch) issue(sb);
            stage_commit<GF>(lF, rF);
            stage_commit<GF>(lQ, rQ);
            T yv[G];
#pragma unroll
            for (int i = 0; i < G; ++i) yv[i] = yn[i];
            if (kPrefetch && sb + 1 < S) issue(sb + 1);
            wave_lds_sync();
            if (SMOOTH && DFORM && sb == 0) {       // (every lane takes part in the shuffle; the last lane keeps what it loaded)
                const T yq = wshfl_down(yv[0], 1);
                if (lane < kWave - 1) yh = yq;
            }
            if (SMOOTH && sb == 0 && lane < kWave - 1) {
                // record 0 of the next lane = this lane's halo step
                load_rec<T, MAT>(reinterpret_cast<const T*>(lF + (lane + 1) * GF::STRIDE), Fh);
                load_rec<T, MAT>(reinterpret_cast<const T*>(lQ + (lane + 1) * GF::STRIDE), Qh);
                have_next = true;
            }
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const long k = k0 + sb * G + i;
                T F[MAT], Qf[MAT];
                stage_get<GF, T, MAT>(lF, i, F);
                stage_get<GF, T, MAT>(lQ, i, Qf);
                filter_apply_step<T, D, SMOOTH, LogLik, DFORM>(a, k, k0, F, Qf, yv[i], h, s, ll, sagg);
                T Pf[MAT];
                full_from_sym<T, D>(s.P, Pf);
                if constexpr (CFG::stage_m) stage_put<GM, T, D, GF::STRIDE>(lQ, i, s.m);    // Q_k is dead too: m_k goes where Q_0..Q_k were
                else store_rec<T, D>(a.fms + k * D, s.m);
                stage_put<GF, T, MAT>(lF, i, Pf);       // F_k is dead after its predict
            }
            wave_lds_sync();
            if constexpr (CFG::stage_m) stage_drain<GM, NT, GF::STRIDE>(gM + (long)sb * GM::SEG, pitchM, lQ);
            stage_drain<GF, NT>(gP + (long)sb * GF::SEG, pitchF, lF);
        }
        if constexpr (SMOOTH && DFORM) filter_tail_apply_u<T, D>(have_next, Fh, Qh, yh, h, a.R, s, sagg);
        else if (SMOOTH) filter_tail_apply<T, D>(have_next, Fh, Qh, s, sagg);
    }
};

// ---------------------------------------------------------------------------------------------
// LDS-DMA ring for the Kalman pass at d = 2, fp64 (configs c2 / c4; 128-lane build only).
//
// Phase stamps of the register-staged kernels at 2^20 steps (profiles/r03_stamps_before.txt): the streaming loops move
// their bytes at 5.7 - 7.3 TB/s -- the memory system's ceiling -- but 17 us of the 81 us pass are prologues and epilogues
// in which every CU folds a spine or scans its lanes while the memory pipes idle.  This variant puts input traffic into
// those phases: F and Q of a wave's sub-tiles travel by `buffer_load_dwordx4 ... lds` (no registers, no ds_write pass)
// into a ring of K slots, requested BEFORE the workgroup folds the spine, so the ring fills while the fold computes and
// the first K sub-tiles of the Kalman pass run at arithmetic speed; y arrives the same way, once per chunk.
//
// A DMA piece is lane-linear on its LDS side (lane p of the wave-instruction writes bytes [16p, 16p+16) of the KiB at
// M0), so the bank-conflict-free image is made on the SOURCE side: piece (owner o, physical position pp) is fetched from
// the owner's LOGICAL piece pp ^ f(o), f(o) = (o & 7) ^ ((o >> 3) & 1).  With that f both the owners' ds_read_b128 of
// their records (four 16-lane groups, 64 banks) and their ds_write_b128 of the results (eight 8-lane groups, 32 banks)
// are conflict-free (checked exhaustively; tools/micro/dma_ring.hip verifies the addressing on the device).
//
// The DMAs are inline asm: hipcc neither counts them in its s_waitcnt bookkeeping nor knows that they write LDS, so
//   * every wait for a slot is a hand-counted s_waitcnt vmcnt(n): n = the vector-memory operations of this wave that are
//     younger than the slot's last piece -- the later slots' pieces (16 per sub-tile) and the drains' stores (12 per
//     sub-tile) -- loads, stores and DMAs retire in issue order;
//   * no compiler-counted LOAD may be in flight while DMAs are: hipcc would wait for it with a count that ignores the
//     DMAs, i.e. for the whole ring.  The kernel therefore pins its small prologue loads (spine, local prefix, halo)
//     before the first DMA is issued, and everything it loads afterwards comes out of LDS.
// ---------------------------------------------------------------------------------------------
// make hipcc finish its own outstanding loads of these values here (it inserts the s_waitcnt in front of the statement)
__device__ __forceinline__ void pin_values(double* v, int n) {
    for (int i = 0; i < n; ++i) asm volatile("" : "+v"(v[i]));
}
template <typename E>
__device__ __forceinline__ void pin_elem(E& e) {
    using TR = ElemTraits<E>;
    typename TR::Scalar v[TR::N];
    pack(e, v);
#pragma unroll
    for (int i = 0; i < TR::N; ++i) asm volatile("" : "+v"(v[i]));
    unpack(v, e);
}

#if defined(PGPS_NARROW)
__device__ __forceinline__ void wait_vmcnt(int n) {
    switch (n) {
#define PGPS_VMC(i) case i: asm volatile("s_waitcnt vmcnt(" #i ")" ::: "memory"); break;
        PGPS_VMC(0) PGPS_VMC(1) PGPS_VMC(2) PGPS_VMC(3) PGPS_VMC(4) PGPS_VMC(5) PGPS_VMC(6) PGPS_VMC(7) PGPS_VMC(8)
        PGPS_VMC(9) PGPS_VMC(10) PGPS_VMC(11) PGPS_VMC(12) PGPS_VMC(13) PGPS_VMC(14) PGPS_VMC(15) PGPS_VMC(16)
        PGPS_VMC(17) PGPS_VMC(18) PGPS_VMC(19) PGPS_VMC(20) PGPS_VMC(21) PGPS_VMC(22) PGPS_VMC(23) PGPS_VMC(24)
        PGPS_VMC(25) PGPS_VMC(26) PGPS_VMC(27) PGPS_VMC(28) PGPS_VMC(29) PGPS_VMC(30) PGPS_VMC(31) PGPS_VMC(32)
        PGPS_VMC(33) PGPS_VMC(34) PGPS_VMC(35) PGPS_VMC(36) PGPS_VMC(37) PGPS_VMC(38) PGPS_VMC(39) PGPS_VMC(40)
        PGPS_VMC(41) PGPS_VMC(42) PGPS_VMC(43) PGPS_VMC(44) PGPS_VMC(45) PGPS_VMC(46) PGPS_VMC(47) PGPS_VMC(48)
        PGPS_VMC(49) PGPS_VMC(50) PGPS_VMC(51) PGPS_VMC(52) PGPS_VMC(53) PGPS_VMC(54) PGPS_VMC(55) PGPS_VMC(56)
        PGPS_VMC(57) PGPS_VMC(58) PGPS_VMC(59) PGPS_VMC(60) PGPS_VMC(61) PGPS_VMC(62)
#undef PGPS_VMC
        default: asm volatile("s_waitcnt vmcnt(63)" ::: "memory"); break;
    }
}

// eight 1 KiB DMA pieces: piece v (owners 8v .. 8v+7) from rs + soff + v * stride + voff{v & 1} to LDS lds + v * 1024
__device__ __forceinline__ void dma_issue8(__amdgpu_buffer_rsrc_t rs, unsigned voff0, unsigned voff1, unsigned soff,
                                           unsigned stride, unsigned lds) {
    unsigned keep, so;
    asm volatile(
        "s_mov_b32 %[keep], m0\n\t"
        "s_mov_b32 m0, %[lds]\n\t"
        "s_mov_b32 %[so], %[soff]\n\t"
        "s_nop 4\n\t"
        "buffer_load_dwordx4 %[v0], %[rs], %[so] offen lds\n\t"
        "s_add_u32 %[so], %[so], %[stride]\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
        "buffer_load_dwordx4 %[v1], %[rs], %[so] offen lds\n\t"
        "s_add_u32 %[so], %[so], %[stride]\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
        "buffer_load_dwordx4 %[v0], %[rs], %[so] offen lds\n\t"
        "s_add_u32 %[so], %[so], %[stride]\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
        "buffer_load_dwordx4 %[v1], %[rs], %[so] offen lds\n\t"
        "s_add_u32 %[so], %[so], %[stride]\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
        "buffer_load_dwordx4 %[v0], %[rs], %[so] offen lds\n\t"
        "s_add_u32 %[so], %[so], %[stride]\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
        "buffer_load_dwordx4 %[v1], %[rs], %[so] offen lds\n\t"
        "s_add_u32 %[so], %[so], %[stride]\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
        "buffer_load_dwordx4 %[v0], %[rs], %[so] offen lds\n\t"
        "s_add_u32 %[so], %[so], %[stride]\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
        "buffer_load_dwordx4 %[v1], %[rs], %[so] offen lds\n\t"
        "s_mov_b32 m0, %[keep]"
        : [keep] "=&s"(keep), [so] "=&s"(so)
        : [lds] "s"(lds), [soff] "s"(soff), [stride] "s"(stride), [v0] "v"(voff0), [v1] "v"(voff1), [rs] "s"(rs)
        : "memory");
}
__device__ __forceinline__ void dma_issue1(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, unsigned lds) {
    unsigned keep;
    asm volatile("s_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[lds]\n\ts_nop 4\n\t"
                 "buffer_load_dwordx4 %[v0], %[rs], %[so] offen lds\n\ts_mov_b32 m0, %[keep]"
                 : [keep] "=&s"(keep) : [lds] "s"(lds), [so] "s"(soff), [v0] "v"(voff), [rs] "s"(rs) : "memory");
}
template <bool SMOOTH, bool NT>
struct FilterApplyDma {
    using T = double;
    static constexpr int D = 2, MAT = 4, G = 4, K = 3;
    static constexpr int SEG = 128, ARR = kWave * SEG, SLOT = 2 * ARR;     // F | Q of one sub-tile
    static constexpr int YB = 32 * 8 * kWave;                               // y of a whole chunk (<= 32 steps per lane)
    static constexpr int BYTES = K * SLOT + YB;                             // per wave
    static constexpr int nD = 16, nS = 12;                                  // DMA pieces / drain stores per sub-tile
    __amdgpu_buffer_rsrc_t rF, rQ, rP, rM;
    unsigned voff0, voff1, fo, lbase, pitch;
    T Fh[MAT], Qh[MAT];
    bool have_next;

    static __device__ __forceinline__ bool usable(const ScanArgs<T>& a) { return a.Lc == 32 || a.Lc == 16; }

    __device__ __forceinline__ void issue(int sb) const {
        const unsigned slot = lbase + (unsigned)(sb % K) * SLOT;
        dma_issue8(rF, voff0, voff1, (unsigned)sb * SEG, 8u * pitch, slot);
        dma_issue8(rQ, voff0, voff1, (unsigned)sb * SEG, 8u * pitch, slot + ARR);
    }

    // issues the first K sub-tiles and the chunk's observations; the caller has pinned every load of its own before
    __device__ __forceinline__ void prefetch(const ScanArgs<T>& a, long wbase, char* lds) {
        const int lane = threadIdx.x & (kWave - 1);
        const int S = a.Lc / G;
        pitch = (unsigned)a.Lc * (MAT * 8u);
        const unsigned span = kWave * pitch;
        rF = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(a.Fs + wbase * MAT), 0, (int)span, 0x00020000);
        rQ = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(a.Qs + wbase * MAT), 0, (int)span, 0x00020000);
        rP = __builtin_amdgcn_make_buffer_rsrc(a.fPs + wbase * MAT, 0, (int)span, 0x00020000);
        rM = __builtin_amdgcn_make_buffer_rsrc(a.fms + wbase * D, 0, (int)(span / 2), 0x00020000);
        const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(a.ys + wbase), 0, kWave * a.Lc * 8, 0x00020000);
        lbase = (unsigned)(size_t)lds;
        const unsigned po = lane >> 3, pp = lane & 7;
        voff0 = po * pitch + ((pp ^ po) << 4);
        voff1 = po * pitch + ((pp ^ po ^ 1u) << 4);
        fo = (lane & 7) ^ ((lane >> 3) & 1);
        issue(0);
        // y: Lc / 2 granules of 16 bytes per owner, owner-major, granule jj of owner o at physical jj ^ h(o)
        const unsigned NG = (unsigned)a.Lc / 2;
        for (int t = 0; t < a.Lc / 2; ++t) {
            const unsigned q = (unsigned)t * kWave + lane;
            const unsigned o = q / NG, pj = q % NG;
            const unsigned ho = (NG == 16) ? (o & 15u) : ((o & 7u) ^ ((o >> 3) & 1u));
            dma_issue1(rY, (o * NG + (pj ^ ho)) * 16u, 0u, lbase + K * SLOT + (unsigned)t * 1024u);
        }
        for (int sb = 1; sb < K && sb < S; ++sb) issue(sb);
    }

    // this lane's record i of array `seg` (its 128-byte segment of a slot)
    static __device__ __forceinline__ void get_rec(const char* seg, unsigned f, int i, T* out) {
        const V4 x0 = *reinterpret_cast<const V4*>(seg + ((((unsigned)(2 * i)) ^ f) << 4));
        const V4 x1 = *reinterpret_cast<const V4*>(seg + ((((unsigned)(2 * i + 1)) ^ f) << 4));
        __builtin_memcpy(&out[0], &x0, 16);
        __builtin_memcpy(&out[2], &x1, 16);
    }

    __device__ __forceinline__ void run(const ScanArgs<T>& a, long wbase, char* lds, const T* h, MeanCov<T, D>& s,
                                        LogLik& ll, SmthElem<T, D>& sagg) {
        const int lane = threadIdx.x & (kWave - 1);
        const int S = a.Lc / G;
        const long k0 = wbase + (long)lane * a.Lc;
        const unsigned NG = (unsigned)a.Lc / 2, hmask = NG - 1;
        const unsigned hself = (NG == 16) ? ((unsigned)lane & 15u) : fo;
        const char* sY = lds + K * SLOT + lane * NG * 16;
        for (int sb = 0; sb < S; ++sb) {
            int later = S - 1 - sb; if (later > K - 1) later = K - 1;
            int prev = sb; if (prev > K - 1) prev = K - 1;
            wait_vmcnt(nD * later + nS * prev);         // 56 at most
            wave_lds_sync();
            char* slot = lds + (sb % K) * SLOT;
            char* sF = slot + lane * SEG;
            char* sQ = sF + ARR;
            if (SMOOTH && sb == 0 && lane < kWave - 1) {
                // record 0 of the next lane = this lane's halo step
                const unsigned fn = ((unsigned)(lane + 1) & 7) ^ (((unsigned)(lane + 1) >> 3) & 1);
                get_rec(sF + SEG, fn, 0, Fh);
                get_rec(sQ + SEG, fn, 0, Qh);
                have_next = true;
            }
            T yv[G];
            {
                const V4 y0 = *reinterpret_cast<const V4*>(sY + ((((unsigned)(2 * sb)) ^ hself) & hmask) * 16);
                const V4 y1 = *reinterpret_cast<const V4*>(sY + ((((unsigned)(2 * sb + 1)) ^ hself) & hmask) * 16);
                __builtin_memcpy(&yv[0], &y0, 16);
                __builtin_memcpy(&yv[2], &y1, 16);
            }
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const long k = k0 + sb * G + i;
                T F[MAT], Qf[MAT];
                get_rec(sF, fo, i, F);
                get_rec(sQ, fo, i, Qf);
                filter_apply_step<T, D, SMOOTH>(a, k, k0, F, Qf, yv[i], h, s, ll, sagg);
                T Pf[MAT];
                full_from_sym<T, D>(s.P, Pf);
                V4 p0, p1, m0;
                __builtin_memcpy(&p0, &Pf[0], 16);
                __builtin_memcpy(&p1, &Pf[2], 16);
                __builtin_memcpy(&m0, &s.m[0], 16);
                *reinterpret_cast<V4*>(sF + ((((unsigned)(2 * i)) ^ fo) << 4)) = p0;         // F_i is dead: P_i takes its place
                *reinterpret_cast<V4*>(sF + ((((unsigned)(2 * i + 1)) ^ fo) << 4)) = p1;
                *reinterpret_cast<V4*>(sQ + ((((unsigned)i) ^ fo) << 4)) = m0;               // logical piece i of Q's segment: Q_0 .. Q_i are dead
            }
            wave_lds_sync();
            // drain: P along the road F came by (same permutation), m as 64 contiguous bytes per owner
#pragma unroll
            for (int v = 0; v < 8; ++v) {
                const V4 x = *reinterpret_cast<const V4*>(slot + v * 1024 + lane * 16);
                __builtin_amdgcn_raw_buffer_store_b128(x, rP, (v & 1) ? voff1 : voff0, (unsigned)sb * SEG + (unsigned)v * 8u * pitch, NT ? 2 : 0);
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const unsigned o = 16u * v + ((unsigned)lane >> 2), i = (unsigned)lane & 3;
                const unsigned f = (o & 7) ^ ((o >> 3) & 1);
                const V4 x = *reinterpret_cast<const V4*>(slot + ARR + o * SEG + ((i ^ f) << 4));
                __builtin_amdgcn_raw_buffer_store_b128(x, rM, o * (pitch / 2) + i * 16u, (unsigned)sb * 64u, NT ? 2 : 0);
            }
            if (sb + K < S) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the slot has been read out before the DMA refills it
                issue(sb + K);
            }
        }
        if (SMOOTH) filter_tail_apply<T, D>(have_next, Fh, Qh, s, sagg);
    }
};
#endif      // PGPS_NARROW

template <typename T, int D, bool SMOOTH, int G, bool NT, bool DMA = false, bool DFORM = false>
__global__ __launch_bounds__(kBlock) void k_filter_apply(const ScanArgs<T> a) {
    static_assert(!(DMA && DFORM), "the LDS-DMA variant keeps the reference's element form");
    constexpr int MAT = D * D, NF = Dim<D>::NFILT;
    using FE = FiltElem<T, D>;
    using SE = SmthElem<T, D>;
    using MC = MeanCov<T, D>;
    using CFG = StageCfg<T, D, G>;
#if defined(PGPS_NARROW)
    using DM = FilterApplyDma<SMOOTH, NT>;
    constexpr int kStageBytes = DMA ? DM::BYTES : CFG::F3_BYTES;
#else
    static_assert(!DMA, "the LDS-DMA ring exists in the 128-lane build only");
    constexpr int kStageBytes = CFG::F3_BYTES;
#endif
    __shared__ T lds[kWaves * NF];
    __shared__ double lds_ll[kWaves];
    __shared__ __attribute__((aligned(16))) char stage[kWaves][kStageBytes];

    T h[D];
#pragma unroll
    for (int i = 0; i < D; ++i) h[i] = a.H[i];

    const long gt = (long)blockIdx.x * kBlock + threadIdx.x;
    const long k0 = gt * a.Lc;
    const long k1 = min(a.N, k0 + a.Lc);
    const int wave = DMA ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave)) : (int)(threadIdx.x / kWave);
    const long wbase = ((long)blockIdx.x * kBlock + wave * kWave) * a.Lc;

    PGPS_STAMP(1, 0);
    // small loads first (spine entries, this lane's local prefix), then the first sub-tile's
    // prefetch: loads retire in order, so the fold below only waits for what it needs
    FE left_part, lp;
    MC s_short;
    // (the shortcut's test waits for one record: tried only where the launch code expects it to hold, never with the LDS-DMA ring)
    const bool shortcut = !DMA && a.shortcut != 0 && blockIdx.x > 0 && carry_shortcut_filter<T, D>(a.spine, (int)blockIdx.x - 1, s_short);
    if (blockIdx.x > 0 && !shortcut) fold_spine_partial<FE>(a.spine, 0, (int)blockIdx.x, left_part);
    ws_load(a.lpre, a.nlanes, gt, lp);
    bool staged = false;
    FilterApplyStaged<T, D, SMOOTH, CFG::GG, NT, DFORM> st;
#if defined(PGPS_NARROW)
    DM dm;
#endif
    if constexpr (CFG::on && !DMA) {
        staged = (wbase + (long)kWave * a.Lc <= a.N) && (a.Lc % G == 0);
        if (staged) st.prefetch(a, wbase);
    }
    // state entering this segment
    MC s;
    if (a.seg_first) {
        T P0f[MAT];
#pragma unroll
        for (int i = 0; i < MAT; ++i) P0f[i] = a.P0[i];
#pragma unroll
        for (int i = 0; i < D; ++i) s.m[i] = T(0);
        sym_from_full<T, D>(P0f, s.P);
    } else {
        // a later segment of a series sharded over GPUs: the prior pushed through the totals of the
        // ranks to the left (every lane folds the <= nranks-1 gathered records itself: no extra launch)
        T P0f[MAT];
#pragma unroll
        for (int i = 0; i < MAT; ++i) P0f[i] = a.P0[i];
#pragma unroll
        for (int i = 0; i < D; ++i) s.m[i] = T(0);
        sym_from_full<T, D>(P0f, s.P);
        const int REC = NF + 2 * MAT;
        for (int r = 0; r < a.rank; ++r) {
            FE e;
            rec_load(a.gathered_f + (long)r * REC, e);
            filt_apply(s, e);
        }
    }
    if (!a.seg_last && blockIdx.x == 0 && threadIdx.x < 2 * MAT) {
        // keep the next segment's first (F, Q) for the smoother launch of this pass
        a.seg_ws[(D + MAT) + threadIdx.x] = a.halo_FQ[threadIdx.x];
    }
#if defined(PGPS_NARROW)
    if constexpr (DMA) {
        // every load hipcc counts is finished HERE, before the first DMA goes out (see the note above FilterApplyDma);
        // then the ring fills while the workgroup folds the spine
        staged = (wbase + (long)kWave * a.Lc <= a.N) && DM::usable(a);
        dm.have_next = false;
        if (staged) {
            const int lane = threadIdx.x & (kWave - 1);
            if (SMOOTH && lane == kWave - 1) dm.have_next = filter_tail_load<T, D>(a, wbase + (long)kWave * a.Lc, dm.Fh, dm.Qh);
            pin_values(dm.Fh, MAT);
            pin_values(dm.Qh, MAT);
        }
        if (blockIdx.x > 0) pin_elem(left_part);
        pin_elem(lp);
        pin_values(h, D);
        pin_values(s.m, D);
        pin_values(s.P, Dim<D>::SYM);
        if (staged) dm.prefetch(a, wbase, stage[wave]);
    }
#endif
#ifdef PGPS_STAMPS
    if (blockIdx.x > 0) pin_elem(left_part);            // diagnostic build: when have the spine records arrived?
    PGPS_STAMP(1, 6);
#endif
    // ... pushed through the workgroups to the left, then through this lane's local prefix
    if (shortcut) {
        s = s_short;
    } else if (blockIdx.x > 0) {
        FE left;
        block_reduce_ordered(left_part, left, lds);
        filt_apply(s, left);
    }
    PGPS_STAMP(1, 1);
    filt_apply(s, lp);
    PGPS_STAMP(1, 2);

    LogLik ll;
    SE sagg;
    smth_identity(sagg);
#if defined(PGPS_NARROW)
    if constexpr (DMA) {
        if (staged) dm.run(a, wbase, stage[wave], h, s, ll, sagg);
    }
#endif
    if constexpr (CFG::on && !DMA) {
        if (staged) st.run(a, wbase, stage[wave], h, s, ll, sagg);
    }
    if (!staged) lane_filter_apply_direct<T, D, SMOOTH, DFORM>(a, k0, k1, h, s, ll, sagg);
    PGPS_STAMP(1, 3);

    // log-likelihood partial of this workgroup
    {
        const double v = ll.value();
        const double t = block_sum_double(v, lds_ll);
        if (threadIdx.x == 0) a.llpart[blockIdx.x] = t;
        if constexpr (!SMOOTH) {                    // filter only: the total as well (with the smoother, its kernel sums)
            if (a.ll != nullptr && a.ll_in_apply) ll_finish(a.llpart, a.nblocks, a.ll, a.status + kLlTicketWord, lds_ll);
        }
    }
    PGPS_STAMP(1, 4);

    if (SMOOTH) {
        SE excl, total;
        block_scan_exclusive<SE, false>(sagg, excl, total, lds);
        ws_store(a.lsuf, a.nlanes, gt, excl);
        if (threadIdx.x == 0) rec_store(a.sspine + (long)blockIdx.x * Dim<D>::NSMTH, total);
    }
    PGPS_STAMP(1, 5);
}

// ---------------------------------------------------------------------------------------------
// K-FS: single-pass filter.  k_filter_reduce and k_filter_apply in ONE launch for d <= 2:
//   phase A  the lane streams its LC steps through LDS into REGISTERS and reduces them (filt_extend);
//            workgroup scan -> per-lane exclusive prefix (stays in registers) + workgroup total
//   hand-off the total is published (agent-scope atomic stores, drained), one arrival is added to a sharded counter,
//            and the workgroup waits until every workgroup has arrived: a grid-wide barrier (round 3; round 2 had every
//            lane poll a flag of its own per predecessor inside look-back windows -- 12 us of wait).  Then it folds the
//            totals to its left exactly as k_filter_apply folds the spine.
//   phase B  lane-serial Kalman pass over the SAME registers -> fms, fPs, ll, smoothing aggregates
// so Fs, Qs, ys are read once: k_filter_reduce's 72 B/step pass -- 295 KB through a CU's memory path, which is what
// bounds the streaming loops (profiles/r03_experiments.txt) -- the lpre round trip and a kernel boundary disappear.
// The barrier needs every workgroup resident: the launch function takes this path only when the grid fits the chip
// (workgroups x waves <= 4 x CUs: one wave per SIMD, the registers of the chunk allow no more), and every spin is
// bounded (status word) so that a grid that is not resident -- another stream holding CUs -- ends instead of hanging.
// Inter-workgroup visibility follows the CDNA4 hand-off rules (cdna_hip_programming.md G16): every handed-off byte is
// written by ONE lane with agent-scope atomic (sc1, write-through) stores, that lane drains its stores
// (s_waitcnt vmcnt(0)) before its arrival; consumers poll relaxed at agent scope and read the payload with agent-scope
// atomic (sc1) loads only.  The fold order is fixed (lane-contiguous runs of totals, then the ordered tree), so the
// result does not depend on timing.
// ---------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void pub_store(T* p, T v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T>
__device__ __forceinline__ T pub_load(const T* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool wait_flag(const int* f, int want, int* status) {
    int spins = 0;
    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        __builtin_amdgcn_s_sleep(4);
        if (++spins > (1 << 22)) {          // ~ a second: give up loudly instead of hanging the GPU
            atomicOr(status, 2);
            return false;
        }
    }
    return true;
}

template <typename T, int D, bool SMOOTH, int LC, bool NT>
__global__ __launch_bounds__(kBlock) void k_filter_single(const ScanArgs<T> a) {
    constexpr int MAT = D * D, SYM = Dim<D>::SYM, NF = Dim<D>::NFILT, G = 4, S = LC / G;
    using FE = FiltElem<T, D>;
    using SE = SmthElem<T, D>;
    using MC = MeanCov<T, D>;
    using CFG = StageCfg<T, D, G>;
    using GF = typename CFG::GF;
    using GM = typename CFG::GM;
    static_assert(CFG::on && LC % G == 0, "single-pass filter: staged dims only, whole sub-tiles");
    __shared__ T lds[kWaves * NF];
    __shared__ double lds_ll[kWaves];
    __shared__ __attribute__((aligned(16))) char stage[kWaves][CFG::F3_BYTES];

    PGPS_STAMP(1, 0);
    const int tile = blockIdx.x;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;

    T h[D];
#pragma unroll
    for (int i = 0; i < D; ++i) h[i] = a.H[i];

    const long gt = (long)tile * kBlock + threadIdx.x;
    const long k0 = gt * LC;
    const long k1 = min(a.N, k0 + LC);
    const long wbase = ((long)tile * kBlock + wave * kWave) * LC;
    const bool staged = (wbase + (long)kWave * LC <= a.N);
    char* lF = stage[wave];
    char* lQ = lF + GF::BYTES;
    const long pitchF = (long)LC * MAT * sizeof(T), pitchM = (long)LC * D * sizeof(T);

    // ---- phase A: stream the chunk into registers and reduce it --------------------------------
    T Freg[LC][MAT], Qreg[LC][SYM];       // Q kept as its symmetric part: 3 instead of 4 values at d = 2
    T yreg[LC];
    T Fh[MAT], Qh[MAT];
    bool have_next = false;
    FE agg;
    filt_identity(agg);
    if (staged) {
        const char* gF = reinterpret_cast<const char*>(a.Fs + wbase * MAT);
        const char* gQ = reinterpret_cast<const char*>(a.Qs + wbase * MAT);
        V4 rF[GF::NV], rQ[GF::NV];
        stage_issue<GF>(gF, pitchF, rF);
        stage_issue<GF>(gQ, pitchF, rQ);
#pragma unroll
        for (int sb = 0; sb < S; ++sb) y_issue<T, G>(a.ys + k0 + sb * G, yreg + sb * G);
        if (SMOOTH && lane == kWave - 1) have_next = filter_tail_load<T, D>(a, wbase + (long)kWave * LC, Fh, Qh);
#pragma unroll
        for (int sb = 0; sb < S; ++sb) {
            wave_lds_sync();
            stage_commit<GF>(lF, rF);
            stage_commit<GF>(lQ, rQ);
            if (sb + 1 < S) {
                stage_issue<GF>(gF + (long)(sb + 1) * GF::SEG, pitchF, rF);
                stage_issue<GF>(gQ + (long)(sb + 1) * GF::SEG, pitchF, rQ);
            }
            wave_lds_sync();
#pragma unroll
            for (int i = 0; i < G; ++i) {
                T Qf[MAT];
                stage_get<GF, T, MAT>(lF, i, Freg[sb * G + i]);
                stage_get<GF, T, MAT>(lQ, i, Qf);
                sym_from_full<T, D>(Qf, Qreg[sb * G + i]);
                filter_reduce_step<T, D>(a, k0 + sb * G + i, Freg[sb * G + i], Qf, yreg[sb * G + i], h, agg);
            }
        }
    } else {
        lane_filter_reduce_direct<T, D>(a, k0, k1, h, agg);
    }
    PGPS_STAMP(1, 1);
    FE excl, total;
    block_scan_exclusive<FE, true>(agg, excl, total, lds);
    PGPS_STAMP(1, 2);

    // ---- hand-off: publish this tile's total, grid barrier, fold the totals to the left ----------------------
    // Every workgroup of the launch is resident (the launch function checks the grid against the chip), so the hand-off
    // is ONE grid-wide barrier: lane 0 stores the total write-through (agent-scope atomic stores = sc1), drains them, and
    // adds one arrival to the counter shard of its tile (8 shards on cache lines of their own: a single word saturates
    // at ~90 atomics per us); lanes 0..7 of the workgroup poll one shard each -- relaxed agent-scope loads with s_sleep,
    // never an acquire per poll -- until every tile has arrived.  The totals are then read with agent-scope atomic
    // (sc1) loads only, which bypass this CU's L1: no acquire fence is needed (MI355X_MICROARCH.md, hand-offs measured
    // with sc1 loads in place of the acquire).  Round 2's version polled one flag per predecessor from every lane of
    // every workgroup (up to 32k pollers hammering L2 while the stragglers were still streaming): its wait took 12 us.
    // Every spin is bounded: a grid that is not resident ends with bit 1 of the status word set instead of hanging.
    if (threadIdx.x == 0) {
        T v[NF];
        pack(total, v);
#pragma unroll
        for (int i = 0; i < NF; ++i) pub_store(a.spine + (long)tile * NF + i, v[i]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(a.flags + (tile & 7) * 32, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x < 8) {
        const int want = (a.nblocks - (int)threadIdx.x + 7) / 8;       // tiles whose index is threadIdx.x mod 8
        wait_flag(a.flags + threadIdx.x * 32, want, a.status);
    }
    __syncthreads();
    PGPS_STAMP(1, 3);
    FE mine;
    filt_identity(mine);
    {
        const int per = (tile + kBlock - 1) / kBlock;
        const int b0 = (int)threadIdx.x * per, b1 = min(tile, b0 + per);
        bool have = false;
        for (int b = b0; b < b1; ++b) {
            T v[NF];
#pragma unroll
            for (int i = 0; i < NF; ++i) v[i] = pub_load(a.spine + (long)b * NF + i);
            FE e;
            unpack(v, e);
            if (have) { FE r; filt_combine(mine, e, r); mine = r; } else { mine = e; have = true; }
        }
    }
    MC s;
    {
        T P0f[MAT];
#pragma unroll
        for (int i = 0; i < MAT; ++i) P0f[i] = a.P0[i];
#pragma unroll
        for (int i = 0; i < D; ++i) s.m[i] = T(0);
        sym_from_full<T, D>(P0f, s.P);
    }
    if (tile > 0) {
        FE left;
        block_reduce_ordered(mine, left, lds);
        filt_apply(s, left);                        // filtered state entering this tile
    }
    filt_apply(s, excl);                            // ... and this lane's chunk

    PGPS_STAMP(1, 4);
    // ---- phase B: lane-serial Kalman pass over the registers ---------------------------------------
    LogLik ll;
    SE sagg;
    smth_identity(sagg);
    if (staged) {
        if (SMOOTH && lane < kWave - 1) have_next = true;
        if (SMOOTH) {
            // halo step of lane l = first step of lane l+1 (its registers); the last lane loaded its own
            T Q0f[MAT];
            full_from_sym<T, D>(Qreg[0], Q0f);
#pragma unroll
            for (int i = 0; i < MAT; ++i) {
                const T f = __shfl_down(Freg[0][i], 1, kWave), q = __shfl_down(Q0f[i], 1, kWave);
                if (lane < kWave - 1) { Fh[i] = f; Qh[i] = q; }
            }
        }
        char* gP = reinterpret_cast<char*>(a.fPs + wbase * MAT);
        char* gM = reinterpret_cast<char*>(a.fms + wbase * D);
#pragma unroll
        for (int sb = 0; sb < S; ++sb) {
            wave_lds_sync();
#pragma unroll
            for (int i = 0; i < G; ++i) {
                T Qf[MAT];
                full_from_sym<T, D>(Qreg[sb * G + i], Qf);
                filter_apply_step<T, D, SMOOTH>(a, k0 + sb * G + i, k0, Freg[sb * G + i], Qf, yreg[sb * G + i], h, s, ll,
                                                sagg);
                T Pf[MAT];
                full_from_sym<T, D>(s.P, Pf);
                stage_put<GM, T, D, GF::STRIDE>(lQ, i, s.m);
                stage_put<GF, T, MAT>(lF, i, Pf);
            }
            wave_lds_sync();
            stage_drain<GM, NT, GF::STRIDE>(gM + (long)sb * GM::SEG, pitchM, lQ);
            stage_drain<GF, NT>(gP + (long)sb * GF::SEG, pitchF, lF);
        }
        if (SMOOTH) filter_tail_apply<T, D>(have_next, Fh, Qh, s, sagg);
    } else {
        lane_filter_apply_direct<T, D, SMOOTH>(a, k0, k1, h, s, ll, sagg);
    }

    PGPS_STAMP(1, 5);
    {
        const double v = ll.value();
        const double t = block_sum_double(v, lds_ll);
        if (threadIdx.x == 0) a.llpart[tile] = t;
    }
    if (SMOOTH) {
        SE sexcl, stotal;
        block_scan_exclusive<SE, false>(sagg, sexcl, stotal, lds);
        ws_store(a.lsuf, a.nlanes, gt, sexcl);
        if (threadIdx.x == 0) rec_store(a.sspine + (long)tile * Dim<D>::NSMTH, stotal);
    }
    PGPS_STAMP(1, 6);
}

// ---------------------------------------------------------------------------------------------
// K-S1: smoother reduce (stand-alone pks only; pkfs gets the aggregates from k_filter_apply)
// ---------------------------------------------------------------------------------------------
template <typename T, int D>
__global__ __launch_bounds__(kBlock) void k_smoother_reduce(const ScanArgs<T> a) {
    constexpr int MAT = D * D, SYM = Dim<D>::SYM;
    using SE = SmthElem<T, D>;
    using MC = MeanCov<T, D>;
    __shared__ T lds[kWaves * Dim<D>::NSMTH];

    const long gt = (long)blockIdx.x * kBlock + threadIdx.x;
    const long k0 = gt * a.Lc;
    const long k1 = min(a.N, k0 + a.Lc);
    SE sagg;
    smth_identity(sagg);
    for (long k = k0; k < k1; ++k) {
        MC f;
        T Pf[MAT];
        load_rec<T, D>(a.fms + k * D, f.m);
        load_rec<T, MAT>(a.fPs + k * MAT, Pf);
        sym_from_full<T, D>(Pf, f.P);
        SE e, r;
        if (k + 1 < a.N || !a.seg_last) {
            T F[MAT], Qf[MAT], Q[SYM];
            if (k + 1 < a.N) {
                load_rec<T, MAT>(a.Fs + (k + 1) * MAT, F);
                load_rec<T, MAT>(a.Qs + (k + 1) * MAT, Qf);
            } else {
#pragma unroll
                for (int i = 0; i < MAT; ++i) { F[i] = a.halo_FQ[i]; Qf[i] = a.halo_FQ[MAT + i]; }
            }
            sym_from_full<T, D>(Qf, Q);
            T mp[D], Pp[SYM], FP[MAT];
            mat_vec<T, D>(F, f.m, mp);
            predict_cov<T, D>(F, f.P, Q, FP, Pp);
            smth_element(f, mp, Pp, FP, e);
        } else {
            smth_last(f, e);
        }
        smth_combine(sagg, e, r);
        sagg = r;
    }
    SE excl, total;
    block_scan_exclusive<SE, false>(sagg, excl, total, lds);
    ws_store(a.lsuf, a.nlanes, gt, excl);
    if (threadIdx.x == 0) rec_store(a.sspine + (long)blockIdx.x * Dim<D>::NSMTH, total);
}

// ---------------------------------------------------------------------------------------------
// K-S3: smoother apply
// ---------------------------------------------------------------------------------------------
// one RTS step: f = filtered (m, P) of step k; (F, Qf) = transition INTO step k+1; s = smoothed k+1 -> k
template <typename T, int D>
__device__ __forceinline__ void smoother_apply_step(const T* F, const T* Qf, const T* mk, const T* Pkf, bool last,
                                                    MeanCov<T, D>& s) {
    constexpr int MAT = D * D, SYM = Dim<D>::SYM;
    MeanCov<T, D> f;
#pragma unroll
    for (int i = 0; i < D; ++i) f.m[i] = mk[i];
    sym_from_full<T, D>(Pkf, f.P);
    if (last) {
        s = f;                              // last element of the series: (0, m_N, P_N)
    } else {
        T Q[SYM], mp[D], Pp[SYM], FP[MAT];
        sym_from_full<T, D>(Qf, Q);
        mat_vec<T, D>(F, f.m, mp);
        predict_cov<T, D>(F, f.P, Q, FP, Pp);
        rts_step(f, mp, Pp, FP, s);
    }
}

// (F, Q) of the step after a chunk: next chunk's first step, the next segment's (halo), or none
template <typename T, int D>
__device__ __forceinline__ void smoother_halo(const ScanArgs<T>& a, long k1, T* Fn, T* Qn) {
    constexpr int MAT = D * D;
    if (k1 < a.N) {
        load_rec<T, MAT>(a.Fs + k1 * MAT, Fn);
        load_rec<T, MAT>(a.Qs + k1 * MAT, Qn);
    } else if (!a.seg_last) {
#pragma unroll
        for (int i = 0; i < MAT; ++i) { Fn[i] = a.halo_FQ[i]; Qn[i] = a.halo_FQ[MAT + i]; }
    } else {
#pragma unroll
        for (int i = 0; i < MAT; ++i) { Fn[i] = T(0); Qn[i] = T(0); }
    }
}

template <typename T, int D>
__device__ __forceinline__ void lane_smoother_apply_direct(const ScanArgs<T>& a, long k0, long k1, MeanCov<T, D>& s) {
    constexpr int MAT = D * D;
    if (k0 >= k1) return;
    T Fn[MAT], Qn[MAT], mn[D], Pn[MAT];
    const bool end_of_series = (k1 == a.N) && a.seg_last;
    smoother_halo<T, D>(a, k1, Fn, Qn);
    load_rec<T, D>(a.fms + (k1 - 1) * D, mn);
    load_rec<T, MAT>(a.fPs + (k1 - 1) * MAT, Pn);
    for (long k = k1 - 1; k >= k0; --k) {
        T F[MAT], Qf[MAT], mk[D], Pk[MAT];
#pragma unroll
        for (int i = 0; i < MAT; ++i) { F[i] = Fn[i]; Qf[i] = Qn[i]; Pk[i] = Pn[i]; }
#pragma unroll
        for (int i = 0; i < D; ++i) mk[i] = mn[i];
        if (k > k0) {
            load_rec<T, MAT>(a.Fs + k * MAT, Fn);
            load_rec<T, MAT>(a.Qs + k * MAT, Qn);
            load_rec<T, D>(a.fms + (k - 1) * D, mn);
            load_rec<T, MAT>(a.fPs + (k - 1) * MAT, Pn);
        }
        smoother_apply_step<T, D>(F, Qf, mk, Pk, end_of_series && k == k1 - 1, s);
        store_rec<T, D>(a.sms + k * D, s.m);
        T Pf[MAT];
        full_from_sym<T, D>(s.P, Pf);
        store_rec<T, MAT>(a.sPs + k * MAT, Pf);
    }
}

// Staged lane-serial RTS pass (sub-tiles in reverse); prefetch() as in FilterApplyStaged.
template <typename T, int D, int G, bool NT>
struct SmootherApplyStaged {
    using CFG = StageCfg<T, D, G>;
    using GF = typename CFG::GF;
    using GM = typename CFG::GM;
    static constexpr int MAT = D * D;
    // The 128-lane build keeps TWO sub-tiles in flight per wave (one workgroup of two waves per CU at 2^20 steps: the
    // bytes in flight, not the registers, bound its streaming rate); the 256-lane build one.
    // (from d = 4 a second set of piece registers would spill: one)
#ifdef PGPS_NARROW
    static constexpr int kDepth = D <= 3 ? 2 : 1;
#else
    static constexpr int kDepth = 1;
#endif
    struct Regs { V4 rF[GF::NV], rQ[GF::NV], rP[GF::NV], rM[GM::NV]; T mv[G][D]; };
    Regs r0, r1;
    T Fc[MAT], Qc[MAT];                 // transition into the step after the one being processed

    __device__ __forceinline__ void issue(const ScanArgs<T>& a, long wbase, int sb, Regs& r) {
        const long pitchF = (long)a.Lc * MAT * sizeof(T), pitchM = (long)a.Lc * D * sizeof(T);
        stage_issue<GF>(reinterpret_cast<const char*>(a.Fs + wbase * MAT) + (long)sb * GF::SEG, pitchF, r.rF);
        stage_issue<GF>(reinterpret_cast<const char*>(a.Qs + wbase * MAT) + (long)sb * GF::SEG, pitchF, r.rQ);
        stage_issue<GF>(reinterpret_cast<const char*>(a.fPs + wbase * MAT) + (long)sb * GF::SEG, pitchF, r.rP);
        if constexpr (CFG::stage_m) {
            stage_issue<GM>(reinterpret_cast<const char*>(a.fms + wbase * D) + (long)sb * GM::SEG, pitchM, r.rM);
        } else {
            const long kb = wbase + (long)(threadIdx.x & (kWave - 1)) * a.Lc + (long)sb * G;
#pragma unroll
            for (int i = 0; i < G; ++i) load_rec<T, D>(a.fms + (kb + i) * D, r.mv[i]);
        }
    }

    __device__ __forceinline__ void prefetch(const ScanArgs<T>& a, long wbase) {
        const int lane = threadIdx.x & (kWave - 1);
        const int S = a.Lc / G;
        issue(a, wbase, S - 1, r0);
        if (kDepth == 2 && S >= 2) issue(a, wbase, S - 2, r1);
        smoother_halo<T, D>(a, wbase + (long)(lane + 1) * a.Lc, Fc, Qc);
    }

    __device__ __forceinline__ void run(const ScanArgs<T>& a, long wbase, char* lds, MeanCov<T, D>& s) {
        const int lane = threadIdx.x & (kWave - 1);
        char* lF = lds;
        char* lQ = lF + GF::BYTES;
        char* lP = lQ + GF::BYTES;          // P in, sP out
        char* lM = lP + GF::BYTES;          // m in, sm out
        const long pitchF = (long)a.Lc * MAT * sizeof(T), pitchM = (long)a.Lc * D * sizeof(T);
        const char* gF = reinterpret_cast<const char*>(a.Fs + wbase * MAT);
        const char* gQ = reinterpret_cast<const char*>(a.Qs + wbase * MAT);
        const char* gP = reinterpret_cast<const char*>(a.fPs + wbase * MAT);
        const char* gM = reinterpret_cast<const char*>(a.fms + wbase * D);
        char* oP = reinterpret_cast<char*>(a.sPs + wbase * MAT);
        char* oM = reinterpret_cast<char*>(a.sms + wbase * D);
        const int S = a.Lc / G;
        const long k1 = wbase + (long)(lane + 1) * a.Lc;
        const bool end_of_series = (k1 == a.N) && a.seg_last;
        (void)gF; (void)gQ; (void)gP; (void)gM;
        auto step = [&](int sb, Regs& r) {
            wave_lds_sync();
            stage_commit<GF>(lF, r.rF);
            stage_commit<GF>(lQ, r.rQ);
            stage_commit<GF>(lP, r.rP);
            T mvv[G][D];
            if constexpr (CFG::stage_m) {
                stage_commit<GM>(lM, r.rM);
            } else {
#pragma unroll
                for (int i = 0; i < G; ++i)
#pragma unroll
                    for (int j = 0; j < D; ++j) mvv[i][j] = r.mv[i][j];
            }
            if (sb - kDepth >= 0) issue(a, wbase, sb - kDepth, r);
            wave_lds_sync();
#pragma unroll
            for (int i = G - 1; i >= 0; --i) {
                T mk[D], Pk[MAT];
                if constexpr (CFG::stage_m) {
                    stage_get<GM, T, D>(lM, i, mk);
                } else {
#pragma unroll
                    for (int j = 0; j < D; ++j) mk[j] = mvv[i][j];
                }
                stage_get<GF, T, MAT>(lP, i, Pk);
                smoother_apply_step<T, D>(Fc, Qc, mk, Pk, end_of_series && sb == S - 1 && i == G - 1, s);
                stage_get<GF, T, MAT>(lF, i, Fc);       // transition into this step: used by step k-1
                stage_get<GF, T, MAT>(lQ, i, Qc);
                T Pf[MAT];
                full_from_sym<T, D>(s.P, Pf);
                if constexpr (CFG::stage_m) stage_put<GM, T, D>(lM, i, s.m);
                else store_rec<T, D>(a.sms + (k1 - a.Lc + (long)sb * G + i) * D, s.m);
                stage_put<GF, T, MAT>(lP, i, Pf);
            }
            wave_lds_sync();
            if constexpr (CFG::stage_m) stage_drain<GM, NT>(oM + (long)sb * GM::SEG, pitchM, lM);
            stage_drain<GF, NT>(oP + (long)sb * GF::SEG, pitchF, lP);
        };
        if constexpr (kDepth == 2) {
            int sb = S - 1;
            for (; sb >= 1; sb -= 2) { step(sb, r0); step(sb - 1, r1); }
            if (sb == 0) step(0, r0);
        } else {
            for (int sb = S - 1; sb >= 0; --sb) step(sb, r0);
        }
    }
};

template <typename T, int D, int G, bool NT>
__global__ __launch_bounds__(kBlock) void k_smoother_apply(const ScanArgs<T> a) {
    constexpr int MAT = D * D, SYM = Dim<D>::SYM, NS = Dim<D>::NSMTH;
    using SE = SmthElem<T, D>;
    using MC = MeanCov<T, D>;
    using CFG = StageCfg<T, D, G>;
    __shared__ T lds[kWaves * NS];
    __shared__ double lds_ll[kWaves];
    __shared__ __attribute__((aligned(16))) char stage[kWaves][CFG::S3_BYTES];

    const long gt = (long)blockIdx.x * kBlock + threadIdx.x;
    const long k0 = gt * a.Lc;
    const long k1 = min(a.N, k0 + a.Lc);
    const int wave = threadIdx.x / kWave;
    const long wbase = ((long)blockIdx.x * kBlock + wave * kWave) * a.Lc;

    PGPS_STAMP(2, 0);
    SE right_part, ls;
    MC s_short;
    const bool has_right = (int)blockIdx.x + 1 < a.nblocks;
    const bool shortcut = a.shortcut != 0 && has_right && carry_shortcut_smoother<T, D>(a.sspine, (int)blockIdx.x + 1, s_short);
    if (has_right && !shortcut) fold_spine_partial<SE>(a.sspine, (int)blockIdx.x + 1, a.nblocks, right_part);
    ws_load(a.lsuf, a.nlanes, gt, ls);
    bool staged = false;
    SmootherApplyStaged<T, D, CFG::GG, NT> st;
    if constexpr (CFG::on) {
        staged = (wbase + (long)kWave * a.Lc <= a.N) && (a.Lc % G == 0);
        // (tried in round 3: finishing the spine / lsuf loads before the prefetch goes out -- hipcc waits for them with
        // s_waitcnt vmcnt(9), which drains most of the 56 KiB prefetch issued after them before the fold's tree starts --
        // same pass time, 82.8 vs 83.0 us: the prefetch then starts a load latency later; profiles/r03_experiments.txt)
        if (staged) st.prefetch(a, wbase);
    }
    // smoothed state of the first step AFTER this segment (irrelevant when seg_last: E = 0 there)
    MC s;
    if (a.seg_last) {
#pragma unroll
        for (int i = 0; i < D; ++i) s.m[i] = T(0);
#pragma unroll
        for (int i = 0; i < SYM; ++i) s.P[i] = T(0);
    } else {
        // smoothed state of the next segment's first step = fold of the totals of the ranks to the right
        // (the last one has E = 0)
#pragma unroll
        for (int i = 0; i < D; ++i) s.m[i] = T(0);
#pragma unroll
        for (int i = 0; i < SYM; ++i) s.P[i] = T(0);
        const int REC = (NS + (NS & 1)) + 2;
        for (int r = a.nranks - 1; r > a.rank; --r) {
            SE e;
            rec_load(a.gathered_s + (long)r * REC, e);
            smth_apply(e, s);
        }
    }
#ifdef PGPS_STAMPS
    if ((int)blockIdx.x + 1 < a.nblocks) pin_elem(right_part);
    PGPS_STAMP(2, 4);
#endif
    if (shortcut) {
        s = s_short;
    } else if ((int)blockIdx.x + 1 < a.nblocks) {
        SE right;
        block_reduce_ordered(right_part, right, lds);
        smth_apply(right, s);
    }
    PGPS_STAMP(2, 1);
    smth_apply(ls, s);
    if (a.dform && k0 < a.N && k1 < a.N) {
        // totals in innovation form: s is (sm - m, sP - P) of the step after the chunk -- add that step's filtered moments
        T mh[D], Ph[MAT], Ps[SYM];
        load_rec<T, D>(a.fms + k1 * D, mh);
        load_rec<T, MAT>(a.fPs + k1 * MAT, Ph);
        sym_from_full<T, D>(Ph, Ps);
#pragma unroll
        for (int i = 0; i < D; ++i) s.m[i] += mh[i];
#pragma unroll
        for (int i = 0; i < SYM; ++i) s.P[i] += Ps[i];
    }
    PGPS_STAMP(2, 2);

    if constexpr (CFG::on) {
        if (staged) st.run(a, wbase, stage[wave], s);
    }
    if (!staged) lane_smoother_apply_direct<T, D>(a, k0, k1, s);
    PGPS_STAMP(2, 3);

    if (blockIdx.x == 0 && a.ll != nullptr) {
        double v = 0.0;
        if (a.gathered_s != nullptr) {
            // sharded series: every rank's partial rides in its smoother record
            const int REC = (NS + (NS & 1)) + 2;
            for (int r = threadIdx.x; r < a.nranks; r += kBlock)
                v += *reinterpret_cast<const double*>(a.gathered_s + (long)r * REC + (REC - 2));
        } else {
            for (int b = threadIdx.x; b < a.nblocks; b += kBlock) v += a.llpart[b];
        }
        const double t = block_sum_double(v, lds_ll);
        if (threadIdx.x == 0) *a.ll = t;
    }
}

// log-likelihood reduction for the filter-only entry point
static __global__ __launch_bounds__(kBlock) void k_ll_finalize(const double* llpart, int nblocks, double* ll) {
    __shared__ double lds_ll[kWaves];
    double v = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += kBlock) v += llpart[b];
    const double t = block_sum_double(v, lds_ll);
    if (threadIdx.x == 0) *ll = t;
}

// ---------------------------------------------------------------------------------------------
// Segment stitching (one segment per GPU).  Tiny single-workgroup kernels around the two
// all-gathers; see pssgp/distributed.py for the protocol.
// ---------------------------------------------------------------------------------------------
// filter record of this segment: [fold of the local spine | F_0 | Q_0]
template <typename T, int D>
__global__ __launch_bounds__(kBlock) void k_seg_filter_total(const ScanArgs<T> a) {
    constexpr int MAT = D * D, NF = Dim<D>::NFILT;
    using FE = FiltElem<T, D>;
    __shared__ T lds[kWaves * NF];
    FE total;
    fold_spine<FE>(a.spine, 0, a.nblocks, total, lds);
    if (threadIdx.x == 0) {
        rec_store(a.rec_f, total);
#pragma unroll
        for (int i = 0; i < MAT; ++i) { a.rec_f[NF + i] = a.Fs[i]; a.rec_f[NF + MAT + i] = a.Qs[i]; }
    }
}

// smoother record of this segment: [fold of the local sspine | pad | ll partial (double)]
template <typename T, int D>
__global__ __launch_bounds__(kBlock) void k_seg_smoother_total(const ScanArgs<T> a, int pad_len) {
    constexpr int NS = Dim<D>::NSMTH;
    using SE = SmthElem<T, D>;
    __shared__ T lds[kWaves * NS];
    __shared__ double lds_ll[kWaves];
    SE total;
    fold_spine<SE>(a.sspine, 0, a.nblocks, total, lds);
    double v = 0.0;
    for (int b = threadIdx.x; b < a.nblocks; b += kBlock) v += a.llpart[b];
    const double t = block_sum_double(v, lds_ll);
    if (threadIdx.x == 0) {
        rec_store(a.rec_s, total);
        *reinterpret_cast<double*>(a.rec_s + pad_len) = t;
    }
}

}  // namespace pgps
