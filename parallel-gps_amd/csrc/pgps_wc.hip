// pgps_wc.hip -- the "wave-cooperative" scan family: state dimensions up to 32 (runtime d).
//
// For d x d operands that no longer fit one lane's registers (RBF order 15, Periodic order 10, the
// d = 11 quasi-periodic config) the parallelism is turned around: ONE WAVEFRONT owns a chunk of Lw
// consecutive time steps and its 64 lanes share every matrix operation.  Operands live in a per-wave
// LDS pool as matrices padded to DP in {8, 16, 24, 32} (leading dimension DP + 2: the +2 spreads the
// eight row blocks over distinct LDS banks); the lanes form an 8 x 8 grid and lane (lr, lc) owns the
// (DP/8) x (DP/8) tile at block (lr, lc) of every result: products accumulate that tile in registers
// with the k loop unrolled by four, reading a column strip of the left operand (broadcast along lc) and
// a row strip of the right one (broadcast along lr).  Padding rows / columns are kept exactly zero.
// The scan has three levels:
//
//   level 1  one wave per chunk of Lw steps (16..64)   (N / Lw waves)     wc_reduce1 / wc_apply1 / wc_smooth1
//   level 2  one wave per group of 8..64 chunk totals                     wc_reduce2 / wc_sreduce2
//   level 3  Kogge-Stone over the group totals, one wave per element and level: wc_ks_filter + wc_fin_filter,
//            wc_ks_smoother + wc_fin_smoother (the single wave that used to walk the groups, wc_carry3 /
//            wc_scarry3, is kept behind PGPS_WC_SERIAL3=1 as a cross-check)
//
// and runs as  wc_reduce1 -> wc_reduce2 -> level 3 -> wc_apply1 (writes fms, fPs, ll partials and
// the smoothing aggregates) -> wc_sreduce2 -> level 3 -> wc_smooth1 (writes sms, sPs).
// Same algebra as pgps_math.h (filt_extend / filt_combine / filt_apply / kf_step / smth_*).  MFMA is not
// used: the contractions are at most 32 wide and interleaved with solves and rank-one updates.
//
// Reference: pssgp/kalman/parallel.py (elements 13-72, operator 100-118, log-lik 135-151, smoothing
// elements 155-173, smoothing operator 176-184).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>

#include "pgps_internal.h"
#include "pgps_math.h"
#include "pgps_wc_args.h"
#include "pgps_gradlti.h"

namespace pgps {
namespace wc {

#ifndef PGPS_WC_X
#define PGPS_WC_X 0            // diagnostic builds only (profiles/r03_experiments.txt): bit 0 no gain elimination,
#endif                         // 1 no smoothing combine, 2 no wave reductions, 3 no symmetrisation of the prediction
constexpr int kGroupMax = 64;   // level-1 totals per level-2 wave: WcArgs::kgroup, 4..64 (launch_scan_wc)

template <int DP>
struct Geo {
    // lane grid GR x GR (lanes >= GR*GR idle in tile operations); DP = 12 serves d = 9..12 (the d = 11
    // config) on a 6 x 6 grid of 2 x 2 tiles with 58 % of the LDS a 16-padding would need
    // (DP = 18 likewise serves d = 17, 18 -- the d = 18 CO2 model -- on 6 x 6 lanes of 3 x 3 tiles: 56 % of the
    // multiply-adds and of the LDS a 24-padding costs, so four waves share a CU instead of two)
    static constexpr int GR = (DP == 12 || DP == 18) ? 6 : 8;
    static constexpr int TS = DP / GR;          // tile edge of one lane
    static constexpr int KU = (DP % 4 == 0) ? 4 : 3;    // unroll of the inner-product loops; DP % KU == 0
    static_assert(DP % KU == 0, "inner-product unroll");
    // inner-product length for state dimension d: d rounded up to the unroll (never beyond the padded slot)
    // DP = 18 serves two state dimensions only: full-length inner products there, unrolled at compile time
    static constexpr bool FIXK = (DP == 18);
    __host__ __device__ static constexpr int dk(int d) { return FIXK ? DP : (d + KU - 1) / KU * KU; }
    static constexpr int LD = DP + 2;           // leading dimension of an LDS matrix
    static constexpr int MSZ = DP * LD;         // elements of one matrix slot (even)
    static constexpr int NRC = 2 * DP + GR;     // leading dimension of combine's right-hand side [A1 | C1 | w | 0..]
    static constexpr int NRA = DP + GR;         // ... of apply's [P | w | 0..] and of the smoother gain's [F P | 0..]
    static_assert(DP % GR == 0 && NRC % GR == 0 && NRA % GR == 0, "tile geometry");
    static constexpr int NFL = 3 * MSZ + 2 * DP;    // a filter 5-tuple in LDS  [A | C | J | b | eta]
    static constexpr int NSL = 2 * MSZ + DP;        // a smoother 3-tuple in LDS [E | L | g]
};

// Every kernel of this family is one wave per workgroup and its LDS pool is private to that wave: LDS executes a
// wave's instructions in issue order, so only the compiler has to be held back.  (A workgroup-scope fence also
// drains vmcnt -- it would wait for the prefetched next step at every one of the dozens of syncs per step.)
__device__ __forceinline__ void sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
template <int DP> __device__ __forceinline__ int lrow() { return (threadIdx.x & 63) / Geo<DP>::GR; }
template <int DP> __device__ __forceinline__ int lcol() { return (threadIdx.x & 63) % Geo<DP>::GR; }
template <int DP> __device__ __forceinline__ bool lactive() { return (threadIdx.x & 63) < Geo<DP>::GR * Geo<DP>::GR; }

template <typename T>
__device__ __forceinline__ T wave_sum(T x) {
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) x += __shfl_xor(x, s, 64);
    return x;
}

// fn(i, j) for every entry of this lane's tile (compile-time trip counts)
template <int DP, typename Fn>
__device__ __forceinline__ void for_tile(Fn fn) {
    constexpr int TS = Geo<DP>::TS;
    const int r0 = lrow<DP>() * TS, c0 = lcol<DP>() * TS;
    if (!lactive<DP>()) return;
#pragma unroll
    for (int ti = 0; ti < TS; ++ti)
#pragma unroll
        for (int tj = 0; tj < TS; ++tj) fn(r0 + ti, c0 + tj);
}

// ---- whole-slot helpers ---------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void slot_copy(int n, const T* src, T* dst) {
    for (int e = lane_id(); e < n; e += 64) dst[e] = src[e];
}
template <typename T>
__device__ __forceinline__ void slot_zero(int n, T* dst) {
    for (int e = lane_id(); e < n; e += 64) dst[e] = T(0);
}
template <typename T, int DP>
__device__ __forceinline__ void mat_eye(int d, T* C) {      // identity on the leading d x d block, zero padding
    for_tile<DP>([&](int i, int j) { C[i * Geo<DP>::LD + j] = (i == j && i < d) ? T(1) : T(0); });
}

// compact global (d x d) <-> padded LDS
template <typename T, int DP>
__device__ __forceinline__ void mat_g2l(int d, const T* g, T* L) {
    for_tile<DP>([&](int i, int j) { L[i * Geo<DP>::LD + j] = (i < d && j < d) ? g[i * d + j] : T(0); });
}
template <typename T, int DP>
__device__ __forceinline__ void mat_l2g(int d, const T* L, T* g) {
    for_tile<DP>([&](int i, int j) { if (i < d && j < d) g[i * d + j] = L[i * Geo<DP>::LD + j]; });
}
template <typename T, int DP>
__device__ __forceinline__ void vec_g2l(int d, const T* g, T* L) {
    if (lane_id() < DP) L[lane_id()] = lane_id() < d ? g[lane_id()] : T(0);
}
template <typename T, int DP>
__device__ __forceinline__ void vec_l2g(int d, const T* L, T* g) {
    if (lane_id() < d) g[lane_id()] = L[lane_id()];
}

// ====================================================================================================
// level 1 kernels keep a lane's tile of a matrix in registers wherever the next operation is tile-local
// (Q, the predicted and the filtered covariance, the smoothing element's L, the elimination of the smoother gain):
// LDS holds only what another lane reads -- the operands of a product and the transposed reads of a symmetrisation.
// ====================================================================================================
template <typename T, int DP>
struct Tile {
    static constexpr int TS = Geo<DP>::TS, LD = Geo<DP>::LD;
    T v[TS][TS];
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int ti = 0; ti < TS; ++ti)
#pragma unroll
            for (int tj = 0; tj < TS; ++tj) v[ti][tj] = T(0);
    }
    __device__ __forceinline__ void ld(const T* M) {            // this lane's tile of an LDS matrix
        const int r0 = lrow<DP>() * TS, c0 = lcol<DP>() * TS;
        if (!lactive<DP>()) { zero(); return; }
#pragma unroll
        for (int ti = 0; ti < TS; ++ti)
#pragma unroll
            for (int tj = 0; tj < TS; ++tj) v[ti][tj] = M[(r0 + ti) * LD + c0 + tj];
    }
    __device__ __forceinline__ void ld_t(const T* M) {          // ... of its transpose
        const int r0 = lrow<DP>() * TS, c0 = lcol<DP>() * TS;
        if (!lactive<DP>()) { zero(); return; }
#pragma unroll
        for (int ti = 0; ti < TS; ++ti)
#pragma unroll
            for (int tj = 0; tj < TS; ++tj) v[ti][tj] = M[(c0 + tj) * LD + r0 + ti];
    }
    __device__ __forceinline__ void st(T* M) const {
        const int r0 = lrow<DP>() * TS, c0 = lcol<DP>() * TS;
        if (!lactive<DP>()) return;
#pragma unroll
        for (int ti = 0; ti < TS; ++ti)
#pragma unroll
            for (int tj = 0; tj < TS; ++tj) M[(r0 + ti) * LD + c0 + tj] = v[ti][tj];
    }
    __device__ __forceinline__ void st_t(T* M) const {          // M = (this matrix)^T
        const int r0 = lrow<DP>() * TS, c0 = lcol<DP>() * TS;
        if (!lactive<DP>()) return;
#pragma unroll
        for (int ti = 0; ti < TS; ++ti)
#pragma unroll
            for (int tj = 0; tj < TS; ++tj) M[(c0 + tj) * LD + r0 + ti] = v[ti][tj];
    }
    // compact global (d x d)
    __device__ __forceinline__ void ld_g(int d, const T* g) {
        const int r0 = lrow<DP>() * TS, c0 = lcol<DP>() * TS;
        const bool act = lactive<DP>();
#pragma unroll
        for (int ti = 0; ti < TS; ++ti)
#pragma unroll
            for (int tj = 0; tj < TS; ++tj) {
                const int i = r0 + ti, j = c0 + tj;
                v[ti][tj] = (act && i < d && j < d) ? g[i * d + j] : T(0);
            }
    }
    __device__ __forceinline__ void st_g(int d, T* g) const {
        const int r0 = lrow<DP>() * TS, c0 = lcol<DP>() * TS;
        if (!lactive<DP>()) return;
#pragma unroll
        for (int ti = 0; ti < TS; ++ti)
#pragma unroll
            for (int tj = 0; tj < TS; ++tj) {
                const int i = r0 + ti, j = c0 + tj;
                if (i < d && j < d) g[i * d + j] = v[ti][tj];
            }
    }
    __device__ __forceinline__ void st_g_t(int d, T* g) const {     // g = (this matrix)^T
        const int r0 = lrow<DP>() * TS, c0 = lcol<DP>() * TS;
        if (!lactive<DP>()) return;
#pragma unroll
        for (int ti = 0; ti < TS; ++ti)
#pragma unroll
            for (int tj = 0; tj < TS; ++tj) {
                const int i = r0 + ti, j = c0 + tj;
                if (i < d && j < d) g[j * d + i] = v[ti][tj];
            }
    }
    // this <- (this + t) / 2: with t the transposed read of the stored tile, the symmetric part
    __device__ __forceinline__ void average(const Tile& t) {
#pragma unroll
        for (int ti = 0; ti < TS; ++ti)
#pragma unroll
            for (int tj = 0; tj < TS; ++tj) v[ti][tj] = T(0.5) * (v[ti][tj] + t.v[ti][tj]);
    }
};

// t += op(A) op(B), the modes of mm.  For the paddings this family really serves (DP >= 18) the operands of the next
// KU inner indices are requested before the multiply-adds of the current ones, with scheduling barriers in between:
// left alone the compiler waits for every pair of indices in turn, and one wave per SIMD has nothing else to run
// (tools/micro/wc_mm.hip: 2600 -> 1440 clocks per d = 18 product on an otherwise idle CU).
template <typename T, int DP, int MODE>
__device__ __forceinline__ void mm_acc(int dk, const T* __restrict__ A, const T* __restrict__ B, Tile<T, DP>& t) {
    constexpr int TS = Geo<DP>::TS, LD = Geo<DP>::LD, KU = Geo<DP>::KU;
    const int r0 = lrow<DP>() * TS, c0 = lcol<DP>() * TS;
    if (!lactive<DP>()) return;
    auto load = [&](T (&av)[KU][TS], T (&bv)[KU][TS], int k0) {
#pragma unroll
        for (int kk = 0; kk < KU; ++kk) {
            const int k = k0 + kk;
#pragma unroll
            for (int ti = 0; ti < TS; ++ti) av[kk][ti] = (MODE == 2) ? A[k * LD + r0 + ti] : A[(r0 + ti) * LD + k];
#pragma unroll
            for (int tj = 0; tj < TS; ++tj) bv[kk][tj] = (MODE == 1) ? B[(c0 + tj) * LD + k] : B[k * LD + c0 + tj];
        }
    };
    auto fma = [&](const T (&av)[KU][TS], const T (&bv)[KU][TS]) {
#pragma unroll
        for (int kk = 0; kk < KU; ++kk)
#pragma unroll
            for (int ti = 0; ti < TS; ++ti)
#pragma unroll
                for (int tj = 0; tj < TS; ++tj) t.v[ti][tj] += av[kk][ti] * bv[kk][tj];
    };
    if constexpr (DP % (2 * KU) == 0 && DP >= 18) {
        // two buffers, two groups per trip; the inner length rounded up to whole trips stays inside the padded slot
        const int dk2 = Geo<DP>::FIXK ? DP : (dk + 2 * KU - 1) / (2 * KU) * (2 * KU);
        T a0[KU][TS], b0[KU][TS], a1[KU][TS], b1[KU][TS];
        load(a0, b0, 0);
        auto trip = [&](int k0) {
            load(a1, b1, k0 + KU);
            __builtin_amdgcn_sched_barrier(0);
            fma(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            if (k0 + 2 * KU < dk2) load(a0, b0, k0 + 2 * KU);
            __builtin_amdgcn_sched_barrier(0);
            fma(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
        };
        if constexpr (Geo<DP>::FIXK) {
#pragma unroll
            for (int k0 = 0; k0 < DP; k0 += 2 * KU) trip(k0);
        } else {
            for (int k0 = 0; k0 < dk2; k0 += 2 * KU) trip(k0);
        }
    } else {
        T av[KU][TS], bv[KU][TS];
        for (int k0 = 0; k0 < dk; k0 += KU) {
            load(av, bv, k0);
            fma(av, bv);
        }
    }
}

// C = op(A) op(B) (+ Add).  MODE 0: A B, 1: A B^T, 2: A^T B.  C must not alias A or B.  dk = d rounded up to the unroll.
template <typename T, int DP, int MODE>
__device__ __forceinline__ void mm(int dk, const T* __restrict__ A, const T* __restrict__ B, T* __restrict__ C,
                                   const T* __restrict__ Add = nullptr) {
    Tile<T, DP> t;
    if (Add) t.ld(Add);
    else t.zero();
    mm_acc<T, DP, MODE>(dk, A, B, t);
    t.st(C);
}

// C = sym_part(C) in place (all reads precede all writes in program order; one wave = lockstep)
template <typename T, int DP>
__device__ __forceinline__ void symmetrise(T* C) {
    constexpr int TS = Geo<DP>::TS, LD = Geo<DP>::LD;
    const int r0 = lrow<DP>() * TS, c0 = lcol<DP>() * TS;
    const bool act = lactive<DP>();
    T v[TS][TS];
    if (act) {
#pragma unroll
        for (int ti = 0; ti < TS; ++ti)
#pragma unroll
            for (int tj = 0; tj < TS; ++tj) v[ti][tj] = T(0.5) * (C[(r0 + ti) * LD + c0 + tj] + C[(c0 + tj) * LD + r0 + ti]);
    }
    sync();
    if (act) {
#pragma unroll
        for (int ti = 0; ti < TS; ++ti)
#pragma unroll
            for (int tj = 0; tj < TS; ++tj) C[(r0 + ti) * LD + c0 + tj] = v[ti][tj];
    }
}

// y = A x (TRANS = false) or A^T x (+ add); y must not alias x.  Lanes 0..DP-1 own one element each.
template <typename T, int DP, bool TRANS>
__device__ __forceinline__ void mv(int dk, const T* __restrict__ A, const T* __restrict__ x, T* __restrict__ y,
                                   const T* __restrict__ add = nullptr) {
    constexpr int LD = Geo<DP>::LD;
    const int i = lane_id();
    if (i < DP) {
        constexpr int KU = Geo<DP>::KU;
        T acc[2] = {add ? add[i] : T(0), T(0)};     // even and odd terms of the sum
        for (int k0 = 0; k0 < dk; k0 += KU) {
            T av[KU];
#pragma unroll
            for (int kk = 0; kk < KU; ++kk) av[kk] = TRANS ? A[(k0 + kk) * LD + i] : A[i * LD + k0 + kk];
#pragma unroll
            for (int kk = 0; kk < KU; ++kk) acc[kk & 1] += av[kk] * x[k0 + kk];
        }
        y[i] = acc[0] + acc[1];
    }
}

template <typename T, int DP>
__device__ __forceinline__ T dot(const T* a, const T* b) {
    const int i = lane_id();
    return wave_sum(i < DP ? a[i] * b[i] : T(0));
}

// Gauss-Jordan: M X = B in place.  M: DP x DP slot (padding = identity rows), B: DP x NRL (NRL multiple
// of 8; lane (lr, lc) owns TS rows x NRL/8 columns of it).  Partial pivoting when PIVOT.  Ends synchronised.
template <typename T, int DP, int NRL, bool PIVOT>
__device__ __forceinline__ void solve(int d, T* M, T* B) {
    constexpr int TS = Geo<DP>::TS, LD = Geo<DP>::LD, TB = NRL / Geo<DP>::GR;
    const int lane = lane_id();
    const int r0 = lrow<DP>() * TS, c0 = lcol<DP>() * TS, b0 = lcol<DP>() * TB;
    const bool act = lactive<DP>();
    for (int c = 0; c < d; ++c) {
        if (PIVOT) {
            T best = T(-1);
            int row = c;
            if (lane >= c && lane < d) { best = fabs(M[lane * LD + c]); row = lane; }
#pragma unroll
            for (int s = 32; s > 0; s >>= 1) {
                const T ob = __shfl_xor(best, s, 64);
                const int orow = __shfl_xor(row, s, 64);
                if (ob > best || (ob == best && orow < row)) { best = ob; row = orow; }
            }
            if (row != c) {
                for (int q = lane; q < LD + NRL; q += 64) {
                    T* pc = q < LD ? M + c * LD + q : B + c * NRL + (q - LD);
                    T* pr = q < LD ? M + row * LD + q : B + row * NRL + (q - LD);
                    const T t = *pc;
                    *pc = *pr;
                    *pr = t;
                }
                sync();
            }
        }
        // everything this lane needs is read before anything is written
        const T inv = T(1) / M[c * LD + c];
        T f[TS], pm[TS], pb[TB], tm[TS][TS], tb[TS][TB];
        if (act) {
#pragma unroll
            for (int ti = 0; ti < TS; ++ti) f[ti] = M[(r0 + ti) * LD + c];
#pragma unroll
            for (int tj = 0; tj < TS; ++tj) pm[tj] = M[c * LD + c0 + tj] * inv;
#pragma unroll
            for (int tj = 0; tj < TB; ++tj) pb[tj] = B[c * NRL + b0 + tj] * inv;
#pragma unroll
            for (int ti = 0; ti < TS; ++ti) {
#pragma unroll
                for (int tj = 0; tj < TS; ++tj) tm[ti][tj] = M[(r0 + ti) * LD + c0 + tj];
#pragma unroll
                for (int tj = 0; tj < TB; ++tj) tb[ti][tj] = B[(r0 + ti) * NRL + b0 + tj];
            }
        }
        sync();
        if (act) {
#pragma unroll
            for (int ti = 0; ti < TS; ++ti) {
                const bool piv = (r0 + ti == c);
#pragma unroll
                for (int tj = 0; tj < TS; ++tj) M[(r0 + ti) * LD + c0 + tj] = piv ? pm[tj] : tm[ti][tj] - f[ti] * pm[tj];
#pragma unroll
                for (int tj = 0; tj < TB; ++tj) B[(r0 + ti) * NRL + b0 + tj] = piv ? pb[tj] : tb[ti][tj] - f[ti] * pb[tj];
            }
        }
        sync();
    }
}

// Gauss-Jordan with partial pivoting on register tiles: M X = B, X over B.  M: DP x DP slot (padding = identity rows), B:
// DP x NRL (lane (lr, lc) owns TS rows x NRL / GR columns of it).  Per elimination step only column c and the pivot row
// travel through LDS -- through the M slot itself, free once the tiles are loaded (its contents are NOT restored).  The
// GR lanes that own column c each offer their best row among those that have not served yet, every lane reads the GR offers
// and finds the same pivot; no rows are exchanged: the row that served column c is written back as row c of X.  Ends
// synchronised.  (Until round 3 the whole of M and B went through LDS twice per step and the pivot search was a wave
// reduction of shuffles: 70 of the 98 thousand clocks of a d = 18 combine.)
template <typename T, int DP, int NRL>
__device__ __forceinline__ void solve_piv(int d, T* M, T* B) {
    constexpr int TS = Geo<DP>::TS, LD = Geo<DP>::LD, GR = Geo<DP>::GR, TB = NRL / GR;
    static_assert(DP <= 32, "the mask of used rows is one word");
    static_assert(DP + 2 * GR + LD + NRL <= DP * LD, "column, offers and pivot row must fit the M slot");
    const int lr = lrow<DP>(), lc = lcol<DP>();
    const int r0 = lr * TS, c0 = lc * TS, b0 = lc * TB;
    const bool act = lactive<DP>();
    T tm[TS][TS], tb[TS][TB];
    int slot[TS];                       // the column each of this lane's rows has served as pivot of; -1: none yet
#pragma unroll
    for (int ti = 0; ti < TS; ++ti) {
        slot[ti] = -1;
#pragma unroll
        for (int tj = 0; tj < TS; ++tj) tm[ti][tj] = act ? M[(r0 + ti) * LD + c0 + tj] : T(0);
#pragma unroll
        for (int tj = 0; tj < TB; ++tj) tb[ti][tj] = act ? B[(r0 + ti) * NRL + b0 + tj] : T(0);
    }
    sync();
    T* col = M;                         // DP values: column c
    T* offer = M + DP;                  // GR pairs: (|entry|, row) of each owner's best row, -1 when it has none
    T* prow = M + DP + 2 * GR;          // LD + NRL values: the pivot row of M, then of B
    unsigned used = 0;
    for (int cb = 0; cb < GR; ++cb) {
#pragma unroll
        for (int cs = 0; cs < TS; ++cs) {
            const int c = cb * TS + cs;
            if (c < d) {
                if (act && lc == cb) {
                    T best = T(-1);
                    int row = r0;
#pragma unroll
                    for (int ti = 0; ti < TS; ++ti) {
                        col[r0 + ti] = tm[ti][cs];
                        const T v = fabs(tm[ti][cs]);
                        const bool ok = r0 + ti < d && !((used >> (r0 + ti)) & 1u);
                        if (ok && v > best) { best = v; row = r0 + ti; }
                    }
                    offer[2 * lr] = best;
                    offer[2 * lr + 1] = T(row);
                }
                sync();
                T best = T(-1);
                int p = 0;
#pragma unroll
                for (int g = 0; g < GR; ++g) {          // ties: the lowest row (the offers come in row order)
                    const T v = offer[2 * g];
                    const int row = (int)offer[2 * g + 1];
                    if (v > best) { best = v; p = row; }
                }
                used |= 1u << p;
                T f[TS];
#pragma unroll
                for (int ti = 0; ti < TS; ++ti) f[ti] = col[act ? r0 + ti : 0];
                if (act && lr == p / TS) {
#pragma unroll
                    for (int ti = 0; ti < TS; ++ti)
                        if (r0 + ti == p) {
#pragma unroll
                            for (int tj = 0; tj < TS; ++tj) prow[c0 + tj] = tm[ti][tj];
#pragma unroll
                            for (int tj = 0; tj < TB; ++tj) prow[LD + b0 + tj] = tb[ti][tj];
                        }
                }
                sync();
                if (act) {
                    const T inv = T(1) / prow[c];
                    T pm[TS], pb[TB];
#pragma unroll
                    for (int tj = 0; tj < TS; ++tj) pm[tj] = prow[c0 + tj] * inv;
#pragma unroll
                    for (int tj = 0; tj < TB; ++tj) pb[tj] = prow[LD + b0 + tj] * inv;
#pragma unroll
                    for (int ti = 0; ti < TS; ++ti) {
                        const bool piv = (r0 + ti == p);
                        slot[ti] = piv ? c : slot[ti];
#pragma unroll
                        for (int tj = 0; tj < TS; ++tj) tm[ti][tj] = piv ? pm[tj] : tm[ti][tj] - f[ti] * pm[tj];
#pragma unroll
                        for (int tj = 0; tj < TB; ++tj) tb[ti][tj] = piv ? pb[tj] : tb[ti][tj] - f[ti] * pb[tj];
                    }
                }
                sync();
            }
        }
    }
    if (act) {
#pragma unroll
        for (int ti = 0; ti < TS; ++ti) {
            const int r = slot[ti] >= 0 ? slot[ti] : r0 + ti;       // (padding rows never serve: they stay where they are)
#pragma unroll
            for (int tj = 0; tj < TB; ++tj) B[r * NRL + b0 + tj] = tb[ti][tj];
        }
    }
    sync();
}

// ---- element records --------------------------------------------------------------------------------
// global (compact): filter [A d^2 | C d^2 | J d^2 | b d | eta d], smoother [E d^2 | L d^2 | g d]
// LDS (padded):     filter [A MSZ | C MSZ | J MSZ | b DP | eta DP], smoother [E MSZ | L MSZ | g DP]

template <typename T, int DP>
struct Filt {
    T *A, *C, *J, *b, *eta;
    __device__ explicit Filt(T* p) : A(p), C(p + Geo<DP>::MSZ), J(p + 2 * Geo<DP>::MSZ), b(p + 3 * Geo<DP>::MSZ),
                                     eta(p + 3 * Geo<DP>::MSZ + DP) {}
};
template <typename T, int DP>
struct Smth {
    T *E, *L, *g;
    __device__ explicit Smth(T* p) : E(p), L(p + Geo<DP>::MSZ), g(p + 2 * Geo<DP>::MSZ) {}
};

template <typename T, int DP>
__device__ __forceinline__ void filt_g2l(int d, const T* g, T* l) {
    Filt<T, DP> f(l);
    mat_g2l<T, DP>(d, g, f.A);
    mat_g2l<T, DP>(d, g + d * d, f.C);
    mat_g2l<T, DP>(d, g + 2 * d * d, f.J);
    vec_g2l<T, DP>(d, g + 3 * d * d, f.b);
    vec_g2l<T, DP>(d, g + 3 * d * d + d, f.eta);
}
template <typename T, int DP>
__device__ __forceinline__ void filt_l2g(int d, const T* l, T* g) {
    Filt<T, DP> f(const_cast<T*>(l));
    mat_l2g<T, DP>(d, f.A, g);
    mat_l2g<T, DP>(d, f.C, g + d * d);
    mat_l2g<T, DP>(d, f.J, g + 2 * d * d);
    vec_l2g<T, DP>(d, f.b, g + 3 * d * d);
    vec_l2g<T, DP>(d, f.eta, g + 3 * d * d + d);
}
template <typename T, int DP>
__device__ __forceinline__ void smth_g2l(int d, const T* g, T* l) {
    Smth<T, DP> s(l);
    mat_g2l<T, DP>(d, g, s.E);
    mat_g2l<T, DP>(d, g + d * d, s.L);
    vec_g2l<T, DP>(d, g + 2 * d * d, s.g);
}
template <typename T, int DP>
__device__ __forceinline__ void smth_l2g(int d, const T* l, T* g) {
    Smth<T, DP> s(const_cast<T*>(l));
    mat_l2g<T, DP>(d, s.E, g);
    mat_l2g<T, DP>(d, s.L, g + d * d);
    vec_l2g<T, DP>(d, s.g, g + 2 * d * d);
}
template <typename T, int DP>
__device__ __forceinline__ void filt_set_identity(int d, T* rec) {
    Filt<T, DP> f(rec);
    slot_zero(Geo<DP>::NFL, rec);
    sync();
    mat_eye<T, DP>(d, f.A);
}
template <typename T, int DP>
__device__ __forceinline__ void smth_set_identity(int d, T* rec) {
    Smth<T, DP> s(rec);
    slot_zero(Geo<DP>::NSL, rec);
    sync();
    mat_eye<T, DP>(d, s.E);
}


// One step's (F, Q) as register tiles: fetched a step ahead, parked in LDS when needed.
template <typename T, int DP>
struct StepTiles {
    static constexpr int TS = Geo<DP>::TS;
    T f[TS][TS], q[TS][TS];
    __device__ __forceinline__ void fetch(int d, const T* Fg, const T* Qg) {
        const int r0 = lrow<DP>() * TS, c0 = lcol<DP>() * TS;
        const bool act = lactive<DP>();
#pragma unroll
        for (int ti = 0; ti < TS; ++ti)
#pragma unroll
            for (int tj = 0; tj < TS; ++tj) {
                const int i = r0 + ti, j = c0 + tj;
                const bool in = act && (i < d && j < d);
                f[ti][tj] = in ? Fg[i * d + j] : T(0);
                // symmetric part of Q straight from global memory (both (i,j) and (j,i) are read)
                q[ti][tj] = in ? T(0.5) * (Qg[i * d + j] + Qg[j * d + i]) : T(0);
            }
    }
    __device__ __forceinline__ void park_f(T* F) const {      // F alone: Q is only ever added tile by tile
        const int r0 = lrow<DP>() * TS, c0 = lcol<DP>() * TS;
        if (!lactive<DP>()) return;
#pragma unroll
        for (int ti = 0; ti < TS; ++ti)
#pragma unroll
            for (int tj = 0; tj < TS; ++tj) F[(r0 + ti) * Geo<DP>::LD + c0 + tj] = f[ti][tj];
    }
};

// first element of the series: update of the prior without a predict (filt_first); P0 is an LDS matrix
template <typename T, int DP>
__device__ __forceinline__ void first_element(int dk, T* acc, const T* P0, T y, const T* h, T R, T* v2) {
    constexpr int LD = Geo<DP>::LD;
    Filt<T, DP> a(acc);
    slot_zero(Geo<DP>::NFL, acc);
    sync();
    slot_copy(Geo<DP>::MSZ, P0, a.C);
    sync();
    symmetrise<T, DP>(a.C);
    sync();
    if (y != y) return;
    mv<T, DP, false>(dk, a.C, h, v2);
    sync();
    const T S = dot<T, DP>(h, v2) + R;
    const T inv = T(1) / S;
    for_tile<DP>([&](int i, int j) { a.C[i * LD + j] -= v2[i] * v2[j] * inv; });
    if (lane_id() < DP) a.b[lane_id()] = v2[lane_id()] * y * inv;
    sync();
}

// out = e1 (x) e2 (filt_combine).  scratch: M (MSZ), rhs (DP x NRC), X (MSZ), vt (DP)
template <typename T, int DP>
__device__ __forceinline__ void combine(int d, int dk, const T* r1, const T* r2, T* rout, T* M, T* rhs, T* X, T* vt) {
    constexpr int LD = Geo<DP>::LD, NR = Geo<DP>::NRC, TS = Geo<DP>::TS;
    Filt<T, DP> e1(const_cast<T*>(r1)), e2(const_cast<T*>(r2)), o(rout);
    // M = I + C1 J2 (identity also on the padding, so the elimination leaves padded rows alone)
    mm<T, DP, 0>(dk, e1.C, e2.J, M);
    sync();
    for_tile<DP>([&](int i, int j) {
        if (i == j) M[i * LD + j] += T(1);
        rhs[i * NR + j] = e1.A[i * LD + j];
        rhs[i * NR + DP + j] = e1.C[i * LD + j];
    });
    if (lane_id() < DP) {
        const int i = lane_id();
        T acc = e1.b[i];
        for (int k = 0; k < dk; ++k) acc += e1.C[i * LD + k] * e2.eta[k];
        rhs[i * NR + 2 * DP] = acc;
#pragma unroll
        for (int q = 1; q < Geo<DP>::GR; ++q) rhs[i * NR + 2 * DP + q] = T(0);
    }
    sync();
    solve_piv<T, DP, NR>(d, M, rhs);        // rhs = [G | Nm | w]
    // A = A2 G ; X = A2 Nm
    {
        const int r0 = lrow<DP>() * TS, c0 = lcol<DP>() * TS;
        const bool act = lactive<DP>();
        T accA[TS][TS], accX[TS][TS];
#pragma unroll
        for (int ti = 0; ti < TS; ++ti)
#pragma unroll
            for (int tj = 0; tj < TS; ++tj) { accA[ti][tj] = T(0); accX[ti][tj] = T(0); }
        for (int k0 = 0; act && k0 < dk; k0 += Geo<DP>::KU) {
#pragma unroll
            for (int kk = 0; kk < Geo<DP>::KU; ++kk) {
                const int k = k0 + kk;
                T a2[TS], g[TS], nm[TS];
#pragma unroll
                for (int ti = 0; ti < TS; ++ti) a2[ti] = e2.A[(r0 + ti) * LD + k];
#pragma unroll
                for (int tj = 0; tj < TS; ++tj) { g[tj] = rhs[k * NR + c0 + tj]; nm[tj] = rhs[k * NR + DP + c0 + tj]; }
#pragma unroll
                for (int ti = 0; ti < TS; ++ti)
#pragma unroll
                    for (int tj = 0; tj < TS; ++tj) { accA[ti][tj] += a2[ti] * g[tj]; accX[ti][tj] += a2[ti] * nm[tj]; }
            }
        }
        if (act) {
#pragma unroll
            for (int ti = 0; ti < TS; ++ti)
#pragma unroll
                for (int tj = 0; tj < TS; ++tj) {
                    o.A[(r0 + ti) * LD + c0 + tj] = accA[ti][tj];
                    X[(r0 + ti) * LD + c0 + tj] = accX[ti][tj];
                }
        }
    }
    if (lane_id() < DP) {
        const int i = lane_id();
        T acc = e2.b[i], z = e2.eta[i];
        for (int k = 0; k < dk; ++k) { acc += e2.A[i * LD + k] * rhs[k * NR + 2 * DP]; z -= e2.J[i * LD + k] * e1.b[k]; }
        o.b[i] = acc;
        vt[i] = z;                          // z = eta2 - J2 b1
    }
    sync();
    mm<T, DP, 1>(dk, X, e2.A, o.C, e2.C);   // C = X A2^T + C2
    mm<T, DP, 0>(dk, e2.J, e1.A, M);        // M <- J2 A1
    sync();
    // J = G^T (J2 A1) + J1 ; eta = G^T z + eta1      (G = leading block of rhs, leading dimension NR)
    {
        const int r0 = lrow<DP>() * TS, c0 = lcol<DP>() * TS;
        const bool act = lactive<DP>();
        T acc[TS][TS];
#pragma unroll
        for (int ti = 0; ti < TS; ++ti)
#pragma unroll
            for (int tj = 0; tj < TS; ++tj) acc[ti][tj] = act ? e1.J[(r0 + ti) * LD + c0 + tj] : T(0);
        for (int k0 = 0; act && k0 < dk; k0 += Geo<DP>::KU) {
#pragma unroll
            for (int kk = 0; kk < Geo<DP>::KU; ++kk) {
                const int k = k0 + kk;
                T g[TS], m[TS];
#pragma unroll
                for (int ti = 0; ti < TS; ++ti) g[ti] = rhs[k * NR + r0 + ti];
#pragma unroll
                for (int tj = 0; tj < TS; ++tj) m[tj] = M[k * LD + c0 + tj];
#pragma unroll
                for (int ti = 0; ti < TS; ++ti)
#pragma unroll
                    for (int tj = 0; tj < TS; ++tj) acc[ti][tj] += g[ti] * m[tj];
            }
        }
        if (act) {
#pragma unroll
            for (int ti = 0; ti < TS; ++ti)
#pragma unroll
                for (int tj = 0; tj < TS; ++tj) o.J[(r0 + ti) * LD + c0 + tj] = acc[ti][tj];
        }
    }
    if (lane_id() < DP) {
        const int i = lane_id();
        T acc = e1.eta[i];
        for (int k = 0; k < dk; ++k) acc += rhs[k * NR + i] * vt[k];
        o.eta[i] = acc;
    }
    sync();
    symmetrise<T, DP>(o.C);
    symmetrise<T, DP>(o.J);
    sync();
}

// (m, P) <- (m, P) pushed through aggregate r2 (filt_apply).  scratch: M (MSZ), rhs (DP x NRA), X (MSZ)
template <typename T, int DP>
__device__ __forceinline__ void apply(int d, int dk, T* m, T* P, const T* r2, T* M, T* rhs, T* X) {
    constexpr int LD = Geo<DP>::LD, NR = Geo<DP>::NRA;
    Filt<T, DP> e2(const_cast<T*>(r2));
    mm<T, DP, 0>(dk, P, e2.J, M);
    sync();
    for_tile<DP>([&](int i, int j) {
        if (i == j) M[i * LD + j] += T(1);
        rhs[i * NR + j] = P[i * LD + j];
    });
    if (lane_id() < DP) {
        const int i = lane_id();
        T acc = m[i];
        for (int k = 0; k < dk; ++k) acc += P[i * LD + k] * e2.eta[k];
        rhs[i * NR + DP] = acc;
#pragma unroll
        for (int q = 1; q < Geo<DP>::GR; ++q) rhs[i * NR + DP + q] = T(0);
    }
    sync();
    solve_piv<T, DP, NR>(d, M, rhs);        // rhs = [Nm | w]
    for_tile<DP>([&](int i, int j) { M[i * LD + j] = rhs[i * NR + j]; });   // Nm as a regular matrix
    if (lane_id() < DP) {
        const int i = lane_id();
        T acc = e2.b[i];
        for (int k = 0; k < dk; ++k) acc += e2.A[i * LD + k] * rhs[k * NR + DP];
        m[i] = acc;
    }
    sync();
    mm<T, DP, 0>(dk, e2.A, M, X);           // X = A2 Nm
    sync();
    mm<T, DP, 1>(dk, X, e2.A, P, e2.C);     // P = X A2^T + C2
    sync();
    symmetrise<T, DP>(P);
    sync();
}

// out = a (x) b in time order (a earlier): E = Ea Eb, g = Ea gb + ga, L = Ea Lb Ea^T + La.  scratch X
template <typename T, int DP>
__device__ __forceinline__ void scombine(int dk, const T* ra, const T* rb, T* rout, T* X) {
    Smth<T, DP> a(const_cast<T*>(ra)), b(const_cast<T*>(rb)), o(rout);
    mm<T, DP, 0>(dk, a.E, b.E, o.E);
    mm<T, DP, 0>(dk, a.E, b.L, X);
    mv<T, DP, false>(dk, a.E, b.g, o.g, a.g);
    sync();
    mm<T, DP, 1>(dk, X, a.E, o.L, a.L);
    sync();
    symmetrise<T, DP>(o.L);
    sync();
}

// (sm, sP) after an aggregate -> at its first step: sm = E sm + g, sP = E sP E^T + L.  scratch X, v, Y
template <typename T, int DP>
__device__ __forceinline__ void sapply(int dk, const T* ra, T* sm, T* sP, T* X, T* v, T* Y) {
    Smth<T, DP> a(const_cast<T*>(ra));
    mv<T, DP, false>(dk, a.E, sm, v, a.g);
    mm<T, DP, 0>(dk, a.E, sP, X);
    sync();
    if (lane_id() < DP) sm[lane_id()] = v[lane_id()];
    mm<T, DP, 1>(dk, X, a.E, Y, a.L);
    sync();
    slot_copy(Geo<DP>::MSZ, Y, sP);
    sync();
    symmetrise<T, DP>(sP);
    sync();
}

// ---- LDS pool ---------------------------------------------------------------------------------------
template <typename T>
struct Pool {
    T* base;
    int off = 0;
    __device__ explicit Pool(T* b) : base(b) {}
    __device__ T* take(int n) { T* p = base + off; off += (n + 1) & ~1; return p; }
};

// level 2: serial combine of a group's chunk totals; stores every chunk's exclusive in-group prefix
template <typename T, int DP>
__global__ __launch_bounds__(64) void wc_reduce2(const WcArgs<T> a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MSZ = Geo<DP>::MSZ, NFL = Geo<DP>::NFL;
    const int d = a.d, nf = nfilt(d), dk = Geo<DP>::dk(d);
    Pool<T> pool(reinterpret_cast<T*>(smem));
    T* acc = pool.take(NFL); T* cur = pool.take(NFL); T* out = pool.take(NFL);
    T* M = pool.take(MSZ); T* rhs = pool.take(DP * Geo<DP>::NRC); T* X = pool.take(MSZ); T* vt = pool.take(DP);
    const int g = blockIdx.x;
    const long c0 = (long)g * a.kgroup, c1 = min(a.nchunk, c0 + a.kgroup);
    filt_set_identity<T, DP>(d, acc);
    sync();
    for (long c = c0; c < c1; ++c) {
        filt_l2g<T, DP>(d, acc, a.lpre1 + c * nf);
        filt_g2l<T, DP>(d, a.agg1 + c * nf, cur);
        sync();
        if (c == c0) {
            slot_copy(NFL, cur, acc);
        } else {
            combine<T, DP>(d, dk, acc, cur, out, M, rhs, X, vt);
            slot_copy(NFL, out, acc);
        }
        sync();
    }
    filt_l2g<T, DP>(d, acc, a.agg2 + (long)g * nf);
}

// level 3: one wave carries (m, P) across the groups
template <typename T, int DP>
__global__ __launch_bounds__(64) void wc_carry3(const WcArgs<T> a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MSZ = Geo<DP>::MSZ;
    const int d = a.d, dd = d * d, nf = nfilt(d), dk = Geo<DP>::dk(d);
    Pool<T> pool(reinterpret_cast<T*>(smem));
    T* m = pool.take(DP); T* P = pool.take(MSZ); T* cur = pool.take(Geo<DP>::NFL);
    T* M = pool.take(MSZ); T* rhs = pool.take(DP * Geo<DP>::NRA); T* X = pool.take(MSZ);
    if (a.carry_in) {
        vec_g2l<T, DP>(d, a.carry_in, m);
        mat_g2l<T, DP>(d, a.carry_in + d, P);
    } else {
        if (lane_id() < DP) m[lane_id()] = T(0);
        mat_g2l<T, DP>(d, a.P0, P);
    }
    sync();
    symmetrise<T, DP>(P);
    sync();
    for (int g = 0; g < a.ngroup; ++g) {
        T* out = a.carry2 + (long)g * (d + dd);
        vec_l2g<T, DP>(d, m, out);
        mat_l2g<T, DP>(d, P, out + d);
        filt_g2l<T, DP>(d, a.agg2 + (long)g * nf, cur);
        sync();
        apply<T, DP>(d, dk, m, P, cur, M, rhs, X);
    }
}

// Level 3 as a Kogge-Stone scan over the group totals (the serial wc_carry3 above costs one `apply` per group on ONE
// wave: 7 ms at 2^20 steps, d = 18): out[i] = in[i - stride] (x) in[i], one wave per element and level.
template <typename T, int DP>
__global__ __launch_bounds__(64) void wc_ks_filter(int d, long n, long stride, const T* in, T* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MSZ = Geo<DP>::MSZ, NFL = Geo<DP>::NFL;
    const int nf = nfilt(d), dk = Geo<DP>::dk(d);
    const long i = blockIdx.x;
    if (i >= n) return;
    if (i < stride) {
        for (int e = lane_id(); e < nf; e += 64) out[i * nf + e] = in[i * nf + e];
        return;
    }
    Pool<T> pool(reinterpret_cast<T*>(smem));
    T* r1 = pool.take(NFL); T* r2 = pool.take(NFL); T* o = pool.take(NFL);
    T* M = pool.take(MSZ); T* rhs = pool.take(DP * Geo<DP>::NRC); T* X = pool.take(MSZ); T* vt = pool.take(DP);
    filt_g2l<T, DP>(d, in + (i - stride) * nf, r1);
    filt_g2l<T, DP>(d, in + i * nf, r2);
    sync();
    combine<T, DP>(d, dk, r1, r2, o, M, rhs, X, vt);
    sync();
    filt_l2g<T, DP>(d, o, out + i * nf);
}

// (m, P) entering group g = the inclusive prefix of the groups before it applied to (0, P0) -- or to the state entering
// the segment; one wave per group
template <typename T, int DP>
__global__ __launch_bounds__(64) void wc_fin_filter(const WcArgs<T> a, const T* incl) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MSZ = Geo<DP>::MSZ;
    const int d = a.d, dd = d * d, nf = nfilt(d), dk = Geo<DP>::dk(d);
    Pool<T> pool(reinterpret_cast<T*>(smem));
    T* m = pool.take(DP); T* P = pool.take(MSZ); T* cur = pool.take(Geo<DP>::NFL);
    T* M = pool.take(MSZ); T* rhs = pool.take(DP * Geo<DP>::NRA); T* X = pool.take(MSZ);
    const int g = blockIdx.x;
    if (g >= a.ngroup) return;
    if (a.carry_in) {                       // a later segment: the state the segments before it leave behind
        vec_g2l<T, DP>(d, a.carry_in, m);
        mat_g2l<T, DP>(d, a.carry_in + d, P);
    } else {
        if (lane_id() < DP) m[lane_id()] = T(0);
        mat_g2l<T, DP>(d, a.P0, P);
    }
    sync();
    symmetrise<T, DP>(P);
    sync();
    if (g > 0) {
        filt_g2l<T, DP>(d, incl + (long)(g - 1) * nf, cur);
        sync();
        apply<T, DP>(d, dk, m, P, cur, M, rhs, X);
        sync();
    }
    T* out = a.carry2 + (long)g * (d + dd);
    vec_l2g<T, DP>(d, m, out);
    mat_l2g<T, DP>(d, P, out + d);
}

// The state entering every chunk, for the two-rows level-1 kernels (pgps_rc2.hip.h): the group's carry pushed through
// the chunk's in-group prefix -- the prologue of wc_apply1 as a kernel of its own (one wave per chunk)
template <typename T, int DP>
__global__ __launch_bounds__(64) void wc_enter1(const WcArgs<T> a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MSZ = Geo<DP>::MSZ;
    const int d = a.d, dd = d * d, nf = nfilt(d), dk = Geo<DP>::dk(d);
    Pool<T> pool(reinterpret_cast<T*>(smem));
    T* m = pool.take(DP); T* P = pool.take(MSZ); T* cur = pool.take(Geo<DP>::NFL);
    T* M = pool.take(MSZ); T* rhs = pool.take(DP * Geo<DP>::NRA); T* X = pool.take(MSZ);
    const long c = blockIdx.x;
    if (c >= a.nchunk) return;
    const int g = (int)(c / a.kgroup);
    const T* cg = a.carry2 + (long)g * (d + dd);
    vec_g2l<T, DP>(d, cg, m);
    mat_g2l<T, DP>(d, cg + d, P);
    filt_g2l<T, DP>(d, a.lpre1 + c * nf, cur);
    sync();
    apply<T, DP>(d, dk, m, P, cur, M, rhs, X);
    sync();
    T* out = a.enter1 + c * (d + dd);
    vec_l2g<T, DP>(d, m, out);
    mat_l2g<T, DP>(d, P, out + d);
}

// ... and the smoothed state of the first step after every chunk (the prologue of wc_smooth1)
template <typename T, int DP>
__global__ __launch_bounds__(64) void wc_senter1(const WcArgs<T> a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MSZ = Geo<DP>::MSZ;
    const int d = a.d, dd = d * d, ns = nsmth(d), dk = Geo<DP>::dk(d);
    Pool<T> pool(reinterpret_cast<T*>(smem));
    T* sm = pool.take(DP); T* sP = pool.take(MSZ); T* cur = pool.take(Geo<DP>::NSL);
    T* X = pool.take(MSZ); T* Y = pool.take(MSZ); T* v = pool.take(DP);
    const long c = blockIdx.x;
    if (c >= a.nchunk) return;
    const int g = (int)(c / a.kgroup);
    const T* cg = a.scarry2 + (long)g * (d + dd);
    vec_g2l<T, DP>(d, cg, sm);
    mat_g2l<T, DP>(d, cg + d, sP);
    smth_g2l<T, DP>(d, a.lsuf1 + c * ns, cur);
    sync();
    sapply<T, DP>(dk, cur, sm, sP, X, v, Y);
    sync();
    T* out = a.senter1 + c * (d + dd);
    vec_l2g<T, DP>(d, sm, out);
    mat_l2g<T, DP>(d, sP, out + d);
}

// suffix scan of the smoothing totals: out[i] = in[i] (x) in[i + stride] (the later element is applied first)
template <typename T, int DP>
__global__ __launch_bounds__(64) void wc_ks_smoother(int d, long n, long stride, const T* in, T* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NSL = Geo<DP>::NSL;
    const int ns = nsmth(d), dk = Geo<DP>::dk(d);
    const long i = blockIdx.x;
    if (i >= n) return;
    if (i + stride >= n) {
        for (int e = lane_id(); e < ns; e += 64) out[i * ns + e] = in[i * ns + e];
        return;
    }
    Pool<T> pool(reinterpret_cast<T*>(smem));
    T* ra = pool.take(NSL); T* rb = pool.take(NSL); T* o = pool.take(NSL); T* X = pool.take(Geo<DP>::MSZ);
    smth_g2l<T, DP>(d, in + i * ns, ra);
    smth_g2l<T, DP>(d, in + (i + stride) * ns, rb);
    sync();
    scombine<T, DP>(dk, ra, rb, o, X);
    sync();
    smth_l2g<T, DP>(d, o, out + i * ns);
}

// (sm, sP) of the first step after group g = (g, L) of the suffix total that starts at group g + 1 (its E multiplies the
// zero state beyond the end of the series); zero after the last group.  Records are [E | L | g], carries [sm | sP].
template <typename T>
__global__ __launch_bounds__(64) void wc_fin_smoother(const WcArgs<T> a, const T* sfx) {
    const int d = a.d, dd = d * d, ns = nsmth(d);
    const int g = blockIdx.x;
    if (g >= a.ngroup) return;
    T* out = a.scarry2 + (long)g * (d + dd);
    const bool last = (g == a.ngroup - 1);
    const T* rec = sfx + (long)(g + 1) * ns;
    for (int e = lane_id(); e < d + dd; e += 64)
        out[e] = last ? T(0) : (e < d ? rec[2 * dd + e] : rec[dd + (e - d)]);
}

// The same with a state behind the segment (a series sharded over several GPUs): (sm, sP) of the next segment's first
// step pushed back through the suffix total that starts at group g + 1; one wave per group.
template <typename T, int DP>
__global__ __launch_bounds__(64) void wc_fin_smoother_cb(const WcArgs<T> a, const T* sfx) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MSZ = Geo<DP>::MSZ;
    const int d = a.d, dd = d * d, ns = nsmth(d), dk = Geo<DP>::dk(d);
    Pool<T> pool(reinterpret_cast<T*>(smem));
    T* sm = pool.take(DP); T* sP = pool.take(MSZ); T* cur = pool.take(Geo<DP>::NSL);
    T* X = pool.take(MSZ); T* Y = pool.take(MSZ); T* v = pool.take(DP);
    const int g = blockIdx.x;
    if (g >= a.ngroup) return;
    vec_g2l<T, DP>(d, a.carry_back, sm);
    mat_g2l<T, DP>(d, a.carry_back + d, sP);
    sync();
    if (g < a.ngroup - 1) {
        smth_g2l<T, DP>(d, sfx + (long)(g + 1) * ns, cur);
        sync();
        sapply<T, DP>(dk, cur, sm, sP, X, v, Y);
    }
    T* out = a.scarry2 + (long)g * (d + dd);
    vec_l2g<T, DP>(d, sm, out);
    mat_l2g<T, DP>(d, sP, out + d);
}

// Segment stitching, one wave: the filtered (m, P) entering segment `rank` = the totals of the segments before it
// (compact records [A | C | J | b | eta], `rank` of them) applied in order to (0, P0) -- segment 0 holds the series'
// first element, so its total already forgets the start.
template <typename T, int DP>
__global__ __launch_bounds__(64) void wc_seg_carry_f(int d, const T* P0, const T* recs, int rank, T* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MSZ = Geo<DP>::MSZ;
    const int nf = nfilt(d), dk = Geo<DP>::dk(d);
    Pool<T> pool(reinterpret_cast<T*>(smem));
    T* m = pool.take(DP); T* P = pool.take(MSZ); T* cur = pool.take(Geo<DP>::NFL);
    T* M = pool.take(MSZ); T* rhs = pool.take(DP * Geo<DP>::NRA); T* X = pool.take(MSZ);
    if (lane_id() < DP) m[lane_id()] = T(0);
    mat_g2l<T, DP>(d, P0, P);
    sync();
    symmetrise<T, DP>(P);
    sync();
    for (int r = 0; r < rank; ++r) {
        filt_g2l<T, DP>(d, recs + (long)r * nf, cur);
        sync();
        apply<T, DP>(d, dk, m, P, cur, M, rhs, X);
        sync();
    }
    vec_l2g<T, DP>(d, m, out);
    mat_l2g<T, DP>(d, P, out + d);
}

// ... and the smoothed (sm, sP) of the first step of segment rank + 1 = the smoothing totals (compact [E | L | g]) of the
// segments behind, applied from the last one backwards to the zero state beyond the end of the series.
template <typename T, int DP>
__global__ __launch_bounds__(64) void wc_seg_carry_s(int d, const T* recs, int rank, int nranks, T* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MSZ = Geo<DP>::MSZ;
    const int ns = nsmth(d), dk = Geo<DP>::dk(d);
    Pool<T> pool(reinterpret_cast<T*>(smem));
    T* sm = pool.take(DP); T* sP = pool.take(MSZ); T* cur = pool.take(Geo<DP>::NSL);
    T* X = pool.take(MSZ); T* Y = pool.take(MSZ); T* v = pool.take(DP);
    if (lane_id() < DP) sm[lane_id()] = T(0);
    slot_zero(MSZ, sP);
    sync();
    for (int r = nranks - 1; r > rank; --r) {
        smth_g2l<T, DP>(d, recs + (long)r * ns, cur);
        sync();
        sapply<T, DP>(dk, cur, sm, sP, X, v, Y);
    }
    vec_l2g<T, DP>(d, sm, out);
    mat_l2g<T, DP>(d, sP, out + d);
}

// smoother gain E = (Pp^-1 F P)^T from Pp and FP (both LDS matrices); M, rhs scratch
template <typename T, int DP>
__device__ __forceinline__ void gain(int d, const T* Pp, const T* FP, T* E, T* M, T* rhs) {
    constexpr int LD = Geo<DP>::LD, NR = Geo<DP>::NRA;
    for_tile<DP>([&](int i, int j) {
        // padding rows of Pp are zero: give the elimination an identity there
        M[i * LD + j] = (i == j && i >= d) ? T(1) : Pp[i * LD + j];
        rhs[i * NR + j] = FP[i * LD + j];
    });
    if (lane_id() < DP) {
#pragma unroll
        for (int q = 0; q < Geo<DP>::GR; ++q) rhs[lane_id() * NR + DP + q] = T(0);
    }
    sync();
    solve<T, DP, NR, false>(d, M, rhs);     // rhs = Pp^-1 F P = E^T
    for_tile<DP>([&](int i, int j) { E[i * LD + j] = rhs[j * NR + i]; });
    sync();
}

// Gauss-Jordan without pivoting on register tiles: tb <- tm^-1 tb for a symmetric positive definite tm (destroyed).
// Per elimination step only column c and row c travel through LDS (gj: 3 DP values), not the matrices.
template <typename T, int DP>
__device__ __forceinline__ void solve_tiles(int d, Tile<T, DP>& tm, Tile<T, DP>& tb, T* gj) {
    constexpr int TS = Geo<DP>::TS, GR = Geo<DP>::GR;
    const int lr = lrow<DP>(), lc = lcol<DP>(), r0 = lr * TS, c0 = lc * TS;
    const bool act = lactive<DP>();
    for (int cb = 0; cb < GR; ++cb) {
#pragma unroll
        for (int cs = 0; cs < TS; ++cs) {
            const int c = cb * TS + cs;
            if (c < d) {
                if (act && lc == cb) {
#pragma unroll
                    for (int ti = 0; ti < TS; ++ti) gj[r0 + ti] = tm.v[ti][cs];
                }
                if (act && lr == cb) {
#pragma unroll
                    for (int tj = 0; tj < TS; ++tj) { gj[DP + c0 + tj] = tm.v[cs][tj]; gj[2 * DP + c0 + tj] = tb.v[cs][tj]; }
                }
                sync();
                if (act) {
                    const T inv = T(1) / gj[DP + c];
                    T f[TS], pm[TS], pb[TS];
#pragma unroll
                    for (int ti = 0; ti < TS; ++ti) f[ti] = gj[r0 + ti];
#pragma unroll
                    for (int tj = 0; tj < TS; ++tj) { pm[tj] = gj[DP + c0 + tj] * inv; pb[tj] = gj[2 * DP + c0 + tj] * inv; }
#pragma unroll
                    for (int ti = 0; ti < TS; ++ti) {
                        const bool piv = (ti == cs) && (lr == cb);
#pragma unroll
                        for (int tj = 0; tj < TS; ++tj) {
                            tm.v[ti][tj] = piv ? pm[tj] : tm.v[ti][tj] - f[ti] * pm[tj];
                            tb.v[ti][tj] = piv ? pb[tj] : tb.v[ti][tj] - f[ti] * pb[tj];
                        }
                    }
                }
                sync();
            }
        }
    }
}

// acc <- acc (x) e in time order (acc earlier), in place: scombine with the results held in registers until every
// operand has been read.  scratch: X (MSZ), u (DP)
template <typename T, int DP>
__device__ __forceinline__ void scombine_tiles(int dk, T* racc, const T* re, T* X, T* u) {
    Smth<T, DP> s(racc), e(const_cast<T*>(re));
    Tile<T, DP> oE, xx, oL;
    oE.zero();
    xx.zero();
    mm_acc<T, DP, 0>(dk, s.E, e.E, oE);
    mm_acc<T, DP, 1>(dk, s.E, e.L, xx);        // L is symmetric: read as L^T, both operands along the inner index
    mv<T, DP, false>(dk, s.E, e.g, u, s.g);
    xx.st(X);
    sync();
    oL.ld(s.L);
    mm_acc<T, DP, 1>(dk, X, s.E, oL);
    sync();
    oE.st(s.E);
    oL.st(s.L);
    if (lane_id() < DP) s.g[lane_id()] = u[lane_id()];
    sync();
    Tile<T, DP> t;
    t.ld_t(s.L);
    oL.average(t);
    sync();
    oL.st(s.L);
    sync();
}

// acc <- acc (x) raw step (F, Q, y): predict the conditional, scalar-innovation update (filt_extend); F A, F C F^T + Q
// and the rank-one updates on register tiles.
// scratch: t2 (MSZ), v1..v3 (DP)
template <typename T, int DP>
__device__ __forceinline__ void extend_tiles(int dk, T* acc, const T* F, const Tile<T, DP>& qt, T y, const T* h, T R,
                                             T* t2, T* v1, T* v2, T* v3) {
    constexpr int TS = Geo<DP>::TS;
    Filt<T, DP> a(acc);
    const int lane = lane_id();
    const int r0 = lrow<DP>() * TS, c0 = lcol<DP>() * TS;
    Tile<T, DP> ta, tc;
    ta.zero();
    tc.zero();
    mm_acc<T, DP, 0>(dk, F, a.A, ta);           // A' = F A
    mv<T, DP, false>(dk, F, a.b, v1);           // b' = F b
    mm_acc<T, DP, 1>(dk, F, a.C, tc);           // F C (C symmetric, read as C^T)
    tc.st(t2);
    sync();
    ta.st(a.A);
    tc = qt;
    mm_acc<T, DP, 1>(dk, t2, F, tc);            // C' = F C F^T + Q
    tc.st(a.C);
    if (lane < DP) a.b[lane] = v1[lane];
    sync();
    {
        Tile<T, DP> t;
        t.ld_t(a.C);
        tc.average(t);
    }
    sync();
    tc.st(a.C);
    sync();
    if (y != y) return;
    mv<T, DP, false>(dk, a.C, h, v2);           // u = C' h
    mv<T, DP, true>(dk, a.A, h, v3);            // v = (h A')^T
    sync();
    const T S = dot<T, DP>(h, v2) + R;
    const T hb = dot<T, DP>(h, a.b);
    const T inv = T(1) / S;
    const T res = y - hb;
    if (lactive<DP>()) {
        Tile<T, DP> tj;
        tj.ld(a.J);
#pragma unroll
        for (int ti = 0; ti < TS; ++ti)
#pragma unroll
            for (int tj_ = 0; tj_ < TS; ++tj_) {
                const T ui = v2[r0 + ti], uj = v2[c0 + tj_], vi = v3[r0 + ti], vj = v3[c0 + tj_];
                ta.v[ti][tj_] -= ui * inv * vj;
                tc.v[ti][tj_] -= ui * uj * inv;
                tj.v[ti][tj_] += vi * vj * inv;
            }
        ta.st(a.A);
        tc.st(a.C);
        tj.st(a.J);
    }
    if (lane < DP) {
        a.b[lane] += v2[lane] * inv * res;
        a.eta[lane] += v3[lane] * res * inv;
    }
    sync();
}

// ====================================================================================================
// level 1: reduce
// ====================================================================================================
template <typename T, int DP>
__global__ __launch_bounds__(64) void wc_reduce1(const WcArgs<T> a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MSZ = Geo<DP>::MSZ, TS = Geo<DP>::TS;
    const int d = a.d, dd = d * d, dk = Geo<DP>::dk(d);
    Pool<T> pool(reinterpret_cast<T*>(smem));
    T* acc = pool.take(Geo<DP>::NFL);
    T* F = pool.take(MSZ); T* t2 = pool.take(MSZ);
    T* h = pool.take(DP); T* v1 = pool.take(DP); T* v2 = pool.take(DP); T* v3 = pool.take(DP);
    const long c = blockIdx.x;
    const long k0 = c * a.Lw, k1 = min(a.N, k0 + a.Lw);
    vec_g2l<T, DP>(d, a.H, h);
    filt_set_identity<T, DP>(d, acc);
    StepTiles<T, DP> st;
    st.fetch(d, a.Fs + k0 * dd, a.Qs + k0 * dd);
    T yn = a.ys[k0];
    sync();
    for (long k = k0; k < k1; ++k) {
        st.park_f(F);
        Tile<T, DP> qt;
#pragma unroll
        for (int ti = 0; ti < TS; ++ti)
#pragma unroll
            for (int tj = 0; tj < TS; ++tj) qt.v[ti][tj] = st.q[ti][tj];
        const T y = yn;
        if (k + 1 < k1) { st.fetch(d, a.Fs + (k + 1) * dd, a.Qs + (k + 1) * dd); yn = a.ys[k + 1]; }
        sync();
        if (k == 0 && a.seg_first) {
            mat_g2l<T, DP>(d, a.P0, t2);
            sync();
            first_element<T, DP>(dk, acc, t2, y, h, a.R, v2);
        } else {
            extend_tiles<T, DP>(dk, acc, F, qt, y, h, a.R, t2, v1, v2, v3);
        }
    }
    filt_l2g<T, DP>(d, acc, a.agg1 + c * nfilt(d));
}

// ====================================================================================================
// level 1: apply -- Kalman pass over the chunk, log-likelihood, smoothing aggregate; the smoother gain of every step
// is left in a.Es for wc_smooth1 (the smoothed covariances' own buffer, overwritten there step by step)
// ====================================================================================================
template <typename T, int DP, bool SMOOTH>
__global__ __launch_bounds__(64) void wc_apply1(const WcArgs<T> a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MSZ = Geo<DP>::MSZ, NSL = Geo<DP>::NSL, TS = Geo<DP>::TS;
    const int d = a.d, dd = d * d, nf = nfilt(d), ns = nsmth(d), dk = Geo<DP>::dk(d);
    Pool<T> pool(reinterpret_cast<T*>(smem));
    T* m = pool.take(DP); T* P = pool.take(MSZ);
    T* F = pool.take(MSZ); T* FP = pool.take(MSZ);      // contiguous: the prologue's right-hand side lies over them
    T* Pp = pool.take(MSZ); T* X = pool.take(MSZ);
    T* Ee = pool.take(NSL); T* sacc = pool.take(NSL);   // contiguous: the prologue's prefix record lies over them
    T* h = pool.take(DP); T* mp = pool.take(DP); T* u = pool.take(DP); T* gj = pool.take(3 * DP);
    T* rhsA = F;
    T* rec = Ee;
    static_assert(DP * Geo<DP>::NRA <= 2 * MSZ, "the prologue's right-hand side must fit two matrix slots");
    static_assert(Geo<DP>::NFL <= 2 * ((NSL + 1) & ~1), "the prefix record must fit the smoothing scratch");
    const long c = blockIdx.x;
    const int g = (int)(c / a.kgroup);
    const long k0 = c * a.Lw, k1 = min(a.N, k0 + a.Lw);
    const int lane = lane_id();
    const int r0 = lrow<DP>() * TS, c0 = lcol<DP>() * TS;
    vec_g2l<T, DP>(d, a.H, h);
    {   // state entering the chunk: group carry pushed through the in-group prefix
        const T* cg = a.carry2 + (long)g * (d + dd);
        vec_g2l<T, DP>(d, cg, m);
        mat_g2l<T, DP>(d, cg + d, P);
        filt_g2l<T, DP>(d, a.lpre1 + c * nf, rec);
        sync();
        apply<T, DP>(d, dk, m, P, rec, Pp, rhsA, X);    // Pp, X, F, FP are free until the loop starts
    }
    if (SMOOTH) smth_set_identity<T, DP>(d, sacc);
    StepTiles<T, DP> st;
    st.fetch(d, a.Fs + k0 * dd, a.Qs + k0 * dd);
    sync();
    double quad = 0.0, mant = 1.0;
    long long expo = 0, count = 0;
    Tile<T, DP> pt;                                      // the filtered covariance (also in LDS: P)
    for (long k = k0; k <= k1; ++k) {
        const bool halo = (k == k1);
        if (halo && (!SMOOTH || (k == a.N && a.seg_last))) break;
        st.park_f(F);
        Tile<T, DP> pp;                                  // Q now, the predicted covariance below
#pragma unroll
        for (int ti = 0; ti < TS; ++ti)
#pragma unroll
            for (int tj = 0; tj < TS; ++tj) pp.v[ti][tj] = st.q[ti][tj];
        if (k + 1 < a.N && (k + 1 < k1 || SMOOTH)) st.fetch(d, a.Fs + (k + 1) * dd, a.Qs + (k + 1) * dd);
        else if (SMOOTH && k + 1 == a.N && !a.seg_last) st.fetch(d, a.halo_F, a.halo_Q);      // the next segment's first step
        Tile<T, DP> pprev;
        T mprev = T(0);
        if (SMOOTH) {
            pprev.ld(P);
            if (lane < DP) mprev = m[lane];
        }
        sync();
        // predict
        Tile<T, DP> fp;
        fp.zero();
        mv<T, DP, false>(dk, F, m, mp);
        mm_acc<T, DP, 1>(dk, F, P, fp);                 // P is symmetric: read as P^T
        fp.st(FP);
        sync();
        mm_acc<T, DP, 1>(dk, FP, F, pp);
        pp.st(Pp);
        sync();
        if (!(PGPS_WC_X & 8)) {
            {
                Tile<T, DP> t;
                t.ld_t(Pp);
                pp.average(t);
            }
            sync();
            pp.st(Pp);
            sync();
        }
        if (SMOOTH && k > k0) {
            // element of step k-1: E = (Pp^-1 F P)^T, g = m - E mp, L = P - sym(E F P)
            Smth<T, DP> e(Ee);
            {
                Tile<T, DP> tm = pp;
                if (!(PGPS_WC_X & 1)) solve_tiles<T, DP>(d, tm, fp, gj);       // fp <- Pp^-1 F P = E^T
            }
            fp.st_t(e.E);
            fp.st_g_t(d, a.Es + (k - 1) * dd);
            sync();
            Tile<T, DP> x;
            x.zero();
            mv<T, DP, false>(dk, e.E, mp, u);
            mm_acc<T, DP, 0>(dk, e.E, FP, x);
            x.st(X);
            sync();
            if (lane < DP) e.g[lane] = mprev - u[lane];
            {
                Tile<T, DP> xt;
                xt.ld_t(X);
#pragma unroll
                for (int ti = 0; ti < TS; ++ti)
#pragma unroll
                    for (int tj = 0; tj < TS; ++tj) pprev.v[ti][tj] -= T(0.5) * (x.v[ti][tj] + xt.v[ti][tj]);
                pprev.st(e.L);
            }
            sync();
            if (!(PGPS_WC_X & 2)) scombine_tiles<T, DP>(dk, sacc, Ee, X, u);
        }
        if (halo) break;
        const T y = a.ys[k];
        const bool obs = !(y != y);
        const bool first = (k == 0 && a.seg_first);
        // log-likelihood term from the predicted moments (also for the first step)
        mv<T, DP, false>(dk, Pp, h, u);
        sync();
        const T S = (PGPS_WC_X & 4) ? a.R + T(1) : dot<T, DP>(h, u) + a.R;
        const T mu = (PGPS_WC_X & 4) ? T(0) : dot<T, DP>(h, mp);
        if (obs) {
            const double r = double(y) - double(mu);
            quad += r * r / double(S);
            int ex;
            mant = frexp(mant * double(S), &ex);
            expo += ex;
            count += 1;
        }
        const bool act = lactive<DP>();
        if (first) {
            // update straight from the prior (m, P still hold m0 = 0, P0)
            mv<T, DP, false>(dk, P, h, u);
            sync();
            const T S0 = dot<T, DP>(h, u) + a.R;
            const T mu0 = dot<T, DP>(h, m);
            pt.ld(P);
            if (obs) {
                const T inv = T(1) / S0;
                if (act) {
#pragma unroll
                    for (int ti = 0; ti < TS; ++ti)
#pragma unroll
                        for (int tj = 0; tj < TS; ++tj) pt.v[ti][tj] -= u[r0 + ti] * u[c0 + tj] * inv;
                }
                pt.st(P);
                if (lane < DP) m[lane] += u[lane] * (y - mu0) * inv;
            }
        } else if (obs) {
            const T inv = T(1) / S;
            if (act) {
#pragma unroll
                for (int ti = 0; ti < TS; ++ti)
#pragma unroll
                    for (int tj = 0; tj < TS; ++tj) pt.v[ti][tj] = pp.v[ti][tj] - u[r0 + ti] * u[c0 + tj] * inv;
            }
            pt.st(P);
            if (lane < DP) m[lane] = mp[lane] + u[lane] * (y - mu) * inv;
        } else {
            pt = pp;
            pt.st(P);
            if (lane < DP) m[lane] = mp[lane];
        }
        sync();
        vec_l2g<T, DP>(d, m, a.fms + k * d);
        pt.st_g(d, a.fPs + k * dd);
    }
    if (SMOOTH && k1 == a.N && a.seg_last) {
        // last element of the series: (0, m_N, P_N)
        Smth<T, DP> e(Ee);
        slot_zero(MSZ, e.E);
        slot_copy(MSZ, P, e.L);
        if (lane < DP) e.g[lane] = m[lane];
        sync();
        scombine_tiles<T, DP>(dk, sacc, Ee, X, u);
    }
    if (SMOOTH) smth_l2g<T, DP>(d, sacc, a.sagg1 + c * ns);
    if (lane == 0) {
        const double logdet = log(mant) + double(expo) * 0.6931471805599453;
        a.llpart[c] = -0.5 * (double(count) * 1.8378770664093453 + logdet + quad);
    }
}

// level 2 (smoother): serial suffix combine of a group's chunk aggregates
template <typename T, int DP>
__global__ __launch_bounds__(64) void wc_sreduce2(const WcArgs<T> a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NSL = Geo<DP>::NSL;
    const int d = a.d, ns = nsmth(d), dk = Geo<DP>::dk(d);
    Pool<T> pool(reinterpret_cast<T*>(smem));
    T* acc = pool.take(NSL); T* cur = pool.take(NSL); T* out = pool.take(NSL); T* X = pool.take(Geo<DP>::MSZ);
    const int g = blockIdx.x;
    const long c0 = (long)g * a.kgroup, c1 = min(a.nchunk, c0 + a.kgroup);
    smth_set_identity<T, DP>(d, acc);
    sync();
    for (long c = c1 - 1; c >= c0; --c) {
        smth_l2g<T, DP>(d, acc, a.lsuf1 + c * ns);
        smth_g2l<T, DP>(d, a.sagg1 + c * ns, cur);
        sync();
        if (c == c1 - 1) {
            slot_copy(NSL, cur, acc);
        } else {
            scombine<T, DP>(dk, cur, acc, out, X);
            slot_copy(NSL, out, acc);
        }
        sync();
    }
    smth_l2g<T, DP>(d, acc, a.sagg2 + (long)g * ns);
}

// level 3 (smoother): one wave carries (sm, sP) from the right; also sums the log-likelihood partials
template <typename T, int DP>
__global__ __launch_bounds__(64) void wc_scarry3(const WcArgs<T> a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MSZ = Geo<DP>::MSZ;
    const int d = a.d, dd = d * d, ns = nsmth(d), dk = Geo<DP>::dk(d);
    Pool<T> pool(reinterpret_cast<T*>(smem));
    T* sm = pool.take(DP); T* sP = pool.take(MSZ); T* cur = pool.take(Geo<DP>::NSL);
    T* X = pool.take(MSZ); T* Y = pool.take(MSZ); T* v = pool.take(DP);
    if (lane_id() < DP) sm[lane_id()] = T(0);
    slot_zero(MSZ, sP);
    sync();
    for (int g = a.ngroup - 1; g >= 0; --g) {
        T* out = a.scarry2 + (long)g * (d + dd);
        vec_l2g<T, DP>(d, sm, out);
        mat_l2g<T, DP>(d, sP, out + d);
        smth_g2l<T, DP>(d, a.sagg2 + (long)g * ns, cur);
        sync();
        sapply<T, DP>(dk, cur, sm, sP, X, v, Y);
    }
    if (a.ll) {
        double t = 0.0;
        for (long c = lane_id(); c < a.nchunk; c += 64) t += a.llpart[c];
        t = wave_sum(t);
        if (lane_id() == 0) *a.ll = t;
    }
}

// level 1 (smoother): RTS pass backwards over the chunk with the gains wc_apply1 left in a.Es
template <typename T, int DP>
__global__ __launch_bounds__(64) void wc_smooth1(const WcArgs<T> a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MSZ = Geo<DP>::MSZ, TS = Geo<DP>::TS;
    const int d = a.d, dd = d * d, ns = nsmth(d), dk = Geo<DP>::dk(d);
    Pool<T> pool(reinterpret_cast<T*>(smem));
    T* sm = pool.take(DP); T* sP = pool.take(MSZ);
    T* F = pool.take(MSZ); T* P = pool.take(MSZ); T* FP = pool.take(MSZ);   // contiguous: the prologue's suffix record
    T* E = pool.take(MSZ); T* X = pool.take(MSZ); T* Y = pool.take(MSZ);
    T* m = pool.take(DP); T* mp = pool.take(DP); T* v = pool.take(DP);
    T* rec = F;
    static_assert(Geo<DP>::NSL <= 3 * MSZ, "the suffix record must fit three matrix slots");
    const long c = blockIdx.x;
    const int g = (int)(c / a.kgroup);
    const long k0 = c * a.Lw, k1 = min(a.N, k0 + a.Lw);
    const int lane = lane_id();
    {
        const T* cg = a.scarry2 + (long)g * (d + dd);
        vec_g2l<T, DP>(d, cg, sm);
        mat_g2l<T, DP>(d, cg + d, sP);
        smth_g2l<T, DP>(d, a.lsuf1 + c * ns, rec);
        sync();
        sapply<T, DP>(dk, rec, sm, sP, X, v, Y);
    }
    StepTiles<T, DP> st;
    if (k1 < a.N) st.fetch(d, a.Fs + k1 * dd, a.Qs + k1 * dd);
    else if (!a.seg_last) st.fetch(d, a.halo_F, a.halo_Q);
    for (long k = k1 - 1; k >= k0; --k) {
        const bool terminal = (k == a.N - 1 && a.seg_last);
        Tile<T, DP> pt, sp;
        pt.ld_g(d, a.fPs + k * dd);
        pt.st(P);
        vec_g2l<T, DP>(d, a.fms + k * d, m);
        sp.ld(sP);                                       // the step after this one
        if (!terminal) {
            mat_g2l<T, DP>(d, a.Es + k * dd, E);
            st.park_f(F);                                // (F, Q) of step k+1
#pragma unroll
            for (int ti = 0; ti < TS; ++ti)
#pragma unroll
                for (int tj = 0; tj < TS; ++tj) sp.v[ti][tj] -= st.q[ti][tj];
        }
        if (k > k0) st.fetch(d, a.Fs + k * dd, a.Qs + k * dd);
        sync();
        if (terminal) {
            if (lane < DP) sm[lane] = m[lane];
            sp = pt;
            sp.st(sP);
            sync();
        } else {
            Tile<T, DP> fp;
            fp.zero();
            mv<T, DP, false>(dk, F, m, mp);
            mm_acc<T, DP, 1>(dk, F, P, fp);             // P is symmetric: read as P^T
            fp.st(FP);
            sync();
            // X = sP' - (F P F^T + Q): the predicted covariance is not symmetrised on its own here -- the smoothed
            // one is, and sym(E X E^T) = E sym(X) E^T
            fp.zero();
            mm_acc<T, DP, 1>(dk, FP, F, fp);
#pragma unroll
            for (int ti = 0; ti < TS; ++ti)
#pragma unroll
                for (int tj = 0; tj < TS; ++tj) sp.v[ti][tj] -= fp.v[ti][tj];
            sp.st(X);
            if (lane < DP) v[lane] = sm[lane] - mp[lane];
            sync();
            Tile<T, DP> y;
            y.zero();
            mv<T, DP, false>(dk, E, v, sm, m);          // sm = m + E (sm' - mp)
            mm_acc<T, DP, 1>(dk, E, X, y);              // X^T for X: the same after the symmetrisation below
            y.st(Y);
            sync();
            sp = pt;
            mm_acc<T, DP, 1>(dk, Y, E, sp);             // sP = P + E (sP' - Pp) E^T
            sp.st(sP);
            sync();
            {
                Tile<T, DP> t;
                t.ld_t(sP);
                sp.average(t);
            }
            sync();
            sp.st(sP);
            sync();
        }
        vec_l2g<T, DP>(d, sm, a.sms + k * d);
        sp.st_g(d, a.sPs + k * dd);
    }
}

// ====================================================================================================
// discretisation for d > 6: one wave per time step, Pade-13 scaling and squaring in LDS (fp64)
// ====================================================================================================
template <typename T, int DP>
__global__ __launch_bounds__(64) void wc_discretise(long N, int d, int steps_per_wave, const T* Fg, const T* Pg,
                                                    const T* ts, T t_prev, T* Fs, T* Qs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MSZ = Geo<DP>::MSZ, LD = Geo<DP>::LD;
    const int dd = d * d, dk = Geo<DP>::dk(d);
    Pool<double> pool(reinterpret_cast<double*>(smem));
    double* A = pool.take(MSZ); double* A2 = pool.take(MSZ); double* A4 = pool.take(MSZ); double* A6 = pool.take(MSZ);
    double* W = pool.take(MSZ); double* U = pool.take(MSZ); double* V = pool.take(MSZ); double* Pm = pool.take(MSZ);
    double* Fm = pool.take(MSZ); double* R2 = pool.take(DP * Geo<DP>::NRA);
    const double b[14] = {64764752532480000., 32382376266240000., 7771770303897600., 1187353796428800.,
                          129060195264000., 10559470521600., 670442572800., 33522128640.,
                          1323241920., 40840800., 960960., 16380., 182., 1.};
    for_tile<DP>([&](int i, int j) {
        const bool in = (i < d && j < d);
        Pm[i * LD + j] = in ? double(Pg[i * d + j]) : 0.0;
        Fm[i * LD + j] = in ? double(Fg[i * d + j]) : 0.0;
    });
    sync();
    for (int q = 0; q < steps_per_wave; ++q) {
        const long k = (long)blockIdx.x * steps_per_wave + q;
        if (k >= N) break;
        const double dt = double(ts[k] - (k > 0 ? ts[k - 1] : t_prev));
        for_tile<DP>([&](int i, int j) { A[i * LD + j] = dt * Fm[i * LD + j]; });
        sync();
        // 1-norm -> scaling
        double col = 0.0;
        if (lane_id() < d) for (int i = 0; i < d; ++i) col += fabs(A[i * LD + lane_id()]);
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) col = fmax(col, __shfl_xor(col, s, 64));
        int sq = 0;
        if (col > 5.371920351148152) {
            sq = (int)ceil(log2(col / 5.371920351148152));
            sq = sq < 0 ? 0 : (sq > 60 ? 60 : sq);
        }
        const double sc = ldexp(1.0, -sq);
        for_tile<DP>([&](int i, int j) { A[i * LD + j] *= sc; });
        sync();
        mm<double, DP, 0>(dk, A, A, A2);
        sync();
        mm<double, DP, 0>(dk, A2, A2, A4);
        sync();
        mm<double, DP, 0>(dk, A4, A2, A6);
        sync();
        for_tile<DP>([&](int i, int j) { const int e = i * LD + j; W[e] = b[13] * A6[e] + b[11] * A4[e] + b[9] * A2[e]; });
        sync();
        mm<double, DP, 0>(dk, A6, W, V);
        sync();
        for_tile<DP>([&](int i, int j) {
            const int e = i * LD + j;
            W[e] = V[e] + b[7] * A6[e] + b[5] * A4[e] + b[3] * A2[e] + ((i == j && i < d) ? b[1] : 0.0);
        });
        sync();
        mm<double, DP, 0>(dk, A, W, U);
        sync();
        for_tile<DP>([&](int i, int j) { const int e = i * LD + j; W[e] = b[12] * A6[e] + b[10] * A4[e] + b[8] * A2[e]; });
        sync();
        mm<double, DP, 0>(dk, A6, W, V);
        sync();
        for_tile<DP>([&](int i, int j) {
            const int e = i * LD + j;
            const double v = V[e] + b[6] * A6[e] + b[4] * A4[e] + b[2] * A2[e] + ((i == j && i < d) ? b[0] : 0.0);
            W[e] = (i == j && i >= d) ? 1.0 : v - U[e];     // M = V - U (identity on the padding)
            R2[i * Geo<DP>::NRA + j] = v + U[e];            // rhs = V + U
        });
        if (lane_id() < DP) {
#pragma unroll
            for (int z = 0; z < Geo<DP>::GR; ++z) R2[lane_id() * Geo<DP>::NRA + DP + z] = 0.0;
        }
        sync();
        solve<double, DP, Geo<DP>::NRA, true>(d, W, R2);    // R2 = expm(A / 2^sq)
        for_tile<DP>([&](int i, int j) { A2[i * LD + j] = R2[i * Geo<DP>::NRA + j]; });
        sync();
        double* R = A2;
        double* Rn = A4;
        for (int t = 0; t < sq; ++t) {
            mm<double, DP, 0>(dk, R, R, Rn);
            sync();
            double* tmp = R; R = Rn; Rn = tmp;
        }
        // Q = Pinf - sym(R Pinf R^T)
        mm<double, DP, 0>(dk, R, Pm, U);
        sync();
        mm<double, DP, 1>(dk, U, R, V);
        sync();
        for_tile<DP>([&](int i, int j) {
            if (i < d && j < d) {
                const double qv = 0.5 * (Pm[i * LD + j] + Pm[j * LD + i]) - 0.5 * (V[i * LD + j] + V[j * LD + i]);
                Qs[k * dd + i * d + j] = T(qv);
                Fs[k * dd + i * d + j] = T(R[i * LD + j]);
            }
        });
        sync();
    }
}

static __global__ __launch_bounds__(64) void wc_ll_finalize(const double* llpart, long n, double* ll) {
    double t = 0.0;
    for (long c = threadIdx.x; c < n; c += 64) t += llpart[c];
    t = wave_sum(t);
    if (threadIdx.x == 0) *ll = t;
}

}  // namespace wc
}  // namespace pgps
#include "pgps_wcgrad.hip.h"
namespace pgps {

// ---- segment records (pgps_seg_*): shared by the row-cooperative and the wave-cooperative drivers ------------
namespace rc {
__host__ __device__ inline int sym_index(int d, int i, int j) {
    return i <= j ? (i * d - (i * (i - 1)) / 2 + (j - i)) : (j * d - (j * (j - 1)) / 2 + (i - j));
}
// this segment's filter record [A | b | C sym | J sym | eta | F_0 | Q_0] from the compact total [A | C | J | b | eta]
template <typename Real>
static __global__ __launch_bounds__(256) void seg_pack_f(int d, const Real* tot, const Real* Fs, const Real* Qs, Real* rec) {
    const int dd = d * d, sym = d * (d + 1) / 2;
    for (int e = threadIdx.x; e < dd; e += 256) {
        const int i = e / d, j = e % d;
        rec[e] = tot[e];
        if (i <= j) {
            rec[dd + d + sym_index(d, i, j)] = tot[dd + e];
            rec[dd + d + sym + sym_index(d, i, j)] = tot[2 * dd + e];
        }
        rec[dd + 2 * d + 2 * sym + e] = Fs[e];
        rec[2 * dd + 2 * d + 2 * sym + e] = Qs[e];
    }
    for (int e = threadIdx.x; e < d; e += 256) { rec[dd + e] = tot[3 * dd + e]; rec[dd + d + 2 * sym + e] = tot[3 * dd + d + e]; }
}
// this segment's smoother record [E | g | L sym | pad | ll partial] from the compact total [E | L | g]
template <typename Real>
static __global__ __launch_bounds__(256) void seg_pack_s(int d, const Real* tot, const double* llpart, long nchunk, int pad,
                                                         Real* rec) {
    __shared__ double part[4];
    const int dd = d * d;
    for (int e = threadIdx.x; e < dd; e += 256) {
        const int i = e / d, j = e % d;
        rec[e] = tot[e];
        if (i <= j) rec[dd + d + sym_index(d, i, j)] = tot[dd + e];
    }
    for (int e = threadIdx.x; e < d; e += 256) rec[dd + e] = tot[2 * dd + e];
    double t = 0.0;
    for (long c = threadIdx.x; c < nchunk; c += 256) t += llpart[c];
    t = wc::wave_sum(t);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = t;
    __syncthreads();
    // the log-likelihood partial travels as a double whatever the record's type (two floats' worth of room: pad is even)
    if (threadIdx.x == 0) *reinterpret_cast<double*>(rec + pad) = part[0] + part[1] + part[2] + part[3];
}
template <typename Real>
static __global__ void seg_ll_sum(const Real* gathered_s, int nranks, int reclen, int pad, double* ll) {
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int r = 0; r < nranks; ++r) t += *reinterpret_cast<const double*>(gathered_s + (long)r * reclen + pad);
        *ll = t;
    }
}

// the inverse of the two packers, one workgroup per rank: gathered records -> compact totals (wave-cooperative layouts)
template <typename Real>
static __global__ __launch_bounds__(256) void seg_unpack_f(int d, const Real* gathered, int reclen, Real* out) {
    const int dd = d * d, sym = d * (d + 1) / 2;
    const Real* rec = gathered + (long)blockIdx.x * reclen;
    Real* tot = out + (long)blockIdx.x * (3 * dd + 2 * d);
    for (int e = threadIdx.x; e < dd; e += 256) {
        const int i = e / d, j = e % d;
        tot[e] = rec[e];
        tot[dd + e] = rec[dd + d + sym_index(d, i, j)];
        tot[2 * dd + e] = rec[dd + d + sym + sym_index(d, i, j)];
    }
    for (int e = threadIdx.x; e < d; e += 256) { tot[3 * dd + e] = rec[dd + e]; tot[3 * dd + d + e] = rec[dd + d + 2 * sym + e]; }
}
template <typename Real>
static __global__ __launch_bounds__(256) void seg_unpack_s(int d, const Real* gathered, int reclen, Real* out) {
    const int dd = d * d;
    const Real* rec = gathered + (long)blockIdx.x * reclen;
    Real* tot = out + (long)blockIdx.x * (2 * dd + d);
    for (int e = threadIdx.x; e < dd; e += 256) {
        const int i = e / d, j = e % d;
        tot[e] = rec[e];
        tot[dd + e] = rec[dd + d + sym_index(d, i, j)];
    }
    for (int e = threadIdx.x; e < d; e += 256) tot[2 * dd + e] = rec[dd + e];
}
}  // namespace rc

// ---- host side ----------------------------------------------------------------------------------------
static inline size_t wc_align(size_t x) { return (x + 255) / 256 * 256; }

// one segment of a sharded series (pgps_seg_*): the records exchanged between the ranks and the scratch of the stitching
template <typename T>
struct WcSeg {
    int rank, nranks;
    T* rec_f; const T* gathered_f;
    T* rec_s; const T* gathered_s;
    T *tot_f, *tot_s;           // (nranks, nfilt) / (nranks, nsmth): the gathered totals, unpacked
    T *carry_in, *carry_back;   // (d + d^2) each
    T* halo;                    // (2, d, d): F, Q of the next segment's first step, kept for the smoother phase
    double* ll;
};

// level-1 kernels of the two-rows family (pgps_rc2.hip.h; pgps_rc2_inst.hip, one unit per padded dimension).
// which: 0 reduce1, 1 apply1 (filter only), 2 apply1 (with the smoothing total), 3 smooth1
// Their padding is their own: the state dimension itself from 18 on (no padding at all), 18 for d = 17.
#define PGPS_RC2_DIMS(X) X(18) X(19) X(20) X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31) X(32)
#define PGPS_RC2_DECL(DPV)                                                       \
    int launch_rc2_##DPV(pgps_ctx*, int which, const wc::WcArgs<double>&);       \
    int launch_rc2_##DPV(pgps_ctx*, int which, const wc::WcArgs<float>&);
PGPS_RC2_DIMS(PGPS_RC2_DECL)
#undef PGPS_RC2_DECL
template <typename T>
static int launch_rc2(pgps_ctx* ctx, int which, const wc::WcArgs<T>& a) {
    switch (a.d < 18 ? 18 : a.d) {
#define PGPS_RC2_CASE(DPV) \
    case DPV: return launch_rc2_##DPV(ctx, which, a);
        PGPS_RC2_DIMS(PGPS_RC2_CASE)
#undef PGPS_RC2_CASE
        default: return PGPS_E_UNSUPPORTED_DIM;
    }
}

template <typename T, int DP>
static int launch_scan_wc_dp(pgps_ctx* ctx, wc::WcArgs<T> a, Mode mode, const WcSeg<T>* sg = nullptr,
                             const GradLtiArgs* gr = nullptr) {
    using namespace wc;
    using GE = Geo<DP>;
    const size_t pad = 64;
    const size_t l_reduce1 = GE::NFL + 2 * GE::MSZ + 4 * DP + pad;
    const size_t l_reduce2 = 3 * GE::NFL + 2 * GE::MSZ + (size_t)DP * GE::NRC + DP + pad;
    const size_t l_carry3 = DP + 3 * GE::MSZ + GE::NFL + (size_t)DP * GE::NRA + pad;
    const size_t l_apply1 = 7 * DP + 5 * GE::MSZ + 2 * (GE::NSL + 1) + pad;
    const size_t l_sred2 = 3 * GE::NSL + GE::MSZ + pad;
    const size_t l_scarry3 = 2 * DP + 3 * GE::MSZ + GE::NSL + pad;
    const size_t l_smooth1 = 4 * DP + 7 * GE::MSZ + pad;
    auto bytes = [](size_t n) { return n * sizeof(T); };
    size_t need = 0;
    for (size_t v : {l_reduce1, l_reduce2, l_carry3, l_apply1, l_sred2, l_scarry3, l_smooth1}) need = need > v ? need : v;
    if (bytes(need) > 160 * 1024) return PGPS_E_UNSUPPORTED_DIM;
    // the dynamic-LDS ceilings are set once per context and instantiation (a dozen runtime calls per scan otherwise:
    // they show at the few-thousand-step series of the experiment drivers)
    bool& attr_done = ctx->wc_attr_done[sizeof(T) == 8 ? 1 : 0][DP / 2];
#define WC_ATTR(K, L)                                                                                              \
    if (!attr_done)                                                                                                \
    HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(K), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                    (int)bytes(L)))
    WC_ATTR((wc_reduce1<T, DP>), l_reduce1);
    WC_ATTR((wc_reduce2<T, DP>), l_reduce2);
    WC_ATTR((wc_carry3<T, DP>), l_carry3);
    WC_ATTR((wc_ks_filter<T, DP>), l_reduce2);
    WC_ATTR((wc_fin_filter<T, DP>), l_carry3);
    WC_ATTR((wc_ks_smoother<T, DP>), l_sred2);
    WC_ATTR((wc_apply1<T, DP, true>), l_apply1);
    WC_ATTR((wc_apply1<T, DP, false>), l_apply1);
    WC_ATTR((wc_sreduce2<T, DP>), l_sred2);
    WC_ATTR((wc_scarry3<T, DP>), l_scarry3);
    WC_ATTR((wc_smooth1<T, DP>), l_smooth1);
    WC_ATTR((wc_fin_smoother_cb<T, DP>), l_scarry3);
    WC_ATTR((wc_seg_carry_f<T, DP>), l_carry3);
    WC_ATTR((wc_seg_carry_s<T, DP>), l_scarry3);
    WC_ATTR((wc_enter1<T, DP>), l_carry3);
    WC_ATTR((wc_senter1<T, DP>), l_scarry3);
#undef WC_ATTR
    attr_done = true;
    const dim3 blk(64), g1((unsigned)a.nchunk), g2((unsigned)a.ngroup);
    // Level 1 (which: 0 reduce, 1 apply filter only, 2 apply with the smoothing total, 3 smooth): the two-rows kernels for
    // d = 17..32 -- every state dimension this family is chosen for -- unless PGPS_WC_ROWS2=0 (or a mask of `which` bits) asks for the
    // LDS-tile kernels of this file (the cross-check of the tests, and d <= 16 when this family is forced)
    auto level1 = [&](int which) -> int {
        if (DP >= 18 && rc2_covers<T>(a.d) && ((ctx->wc_rows2 >> which) & 1)) {
            // (the states entering the chunks close the scan: timed with its slots, so that the apply slots stay one
            //  launch per pass -- the launch bench.py prices against the roofline)
            if (which == 1 || which == 2)
                timed_launch(ctx, PGPS_K_FILTER_REDUCE, wc_enter1<T, DP>, g1, blk, (unsigned)bytes(l_carry3), a);
            if (which == 3) timed_launch(ctx, PGPS_K_SMOOTHER_REDUCE, wc_senter1<T, DP>, g1, blk, (unsigned)bytes(l_scarry3), a);
            return launch_rc2<T>(ctx, which, a);
        }
        switch (which) {
            case 0: timed_launch(ctx, PGPS_K_FILTER_REDUCE, wc_reduce1<T, DP>, g1, blk, (unsigned)bytes(l_reduce1), a); break;
            case 1: timed_launch(ctx, PGPS_K_FILTER_APPLY, wc_apply1<T, DP, false>, g1, blk, (unsigned)bytes(l_apply1), a); break;
            case 2: timed_launch(ctx, PGPS_K_FILTER_APPLY, wc_apply1<T, DP, true>, g1, blk, (unsigned)bytes(l_apply1), a); break;
            default: timed_launch(ctx, PGPS_K_SMOOTHER_APPLY, wc_smooth1<T, DP>, g1, blk, (unsigned)bytes(l_smooth1), a); break;
        }
        return PGPS_OK;
    };
    // one Kogge-Stone level over the group totals: the two-rows combine kernel (two elements per wave, elimination in
    // registers) wherever the two-rows level-1 kernels run; PGPS_WC_KS2=0 (diagnostic, read once) keeps the LDS-tile one
    static const bool ks2_env = [] { const char* e = std::getenv("PGPS_WC_KS2"); return !(e && e[0] == '0'); }();
    auto ks_filter_level = [&](long stride, const T* cur, T* nxt) -> int {
        if (DP >= 18 && rc2_covers<T>(a.d) && (ctx->wc_rows2 & 1) && ks2_env) {
            a.ks_n = a.ngroup; a.ks_stride = stride; a.ks_in = cur; a.ks_out = nxt;
            return launch_rc2<T>(ctx, 4, a);
        }
        timed_launch(ctx, PGPS_K_FILTER_REDUCE, wc_ks_filter<T, DP>, g2, blk, (unsigned)bytes(l_reduce2), a.d, (long)a.ngroup,
                     stride, cur, nxt);
        return PGPS_OK;
    };
    auto ks_smoother_level = [&](long stride, const T* cur, T* nxt) -> int {
        if (DP >= 18 && rc2_covers<T>(a.d) && (ctx->wc_rows2 & 8) && ks2_env) {
            a.ks_n = a.ngroup; a.ks_stride = stride; a.ks_in = cur; a.ks_out = nxt;
            return launch_rc2<T>(ctx, 5, a);
        }
        timed_launch(ctx, PGPS_K_SMOOTHER_REDUCE, wc_ks_smoother<T, DP>, g2, blk, (unsigned)bytes(l_sred2), a.d, (long)a.ngroup,
                     stride, cur, nxt);
        return PGPS_OK;
    };
#define WC_LEVEL1(which)                 \
    do {                                 \
        const int r1_ = level1(which);   \
        if (r1_) return r1_;             \
    } while (0)
    if (sg) {
        // Three phases with the ranks' records exchanged in between (pssgp/distributed.py); chunk / group totals, their
        // scans and the in-group prefixes stay in the workspace from one phase to the next.
        if (ctx->wc_serial3) return PGPS_E_UNSUPPORTED_DIM;     // the diagnostic serial walk keeps no inclusive totals
        const int d = a.d, dd = d * d, nf = nfilt(d), ns = nsmth(d);
        const int rf = seg_rec_f_len(d), rs = seg_rec_s_len(d), spad = seg_rec_s_pad(d);
        const int nfp = dd + d + d * (d + 1) + d;               // packed five-tuple ahead of (F_0, Q_0) in a filter record
        // Kogge-Stone over the group totals: first -> ksA -> ksB -> ksA ...; where a finished scan lies
        auto ks_where = [&](T* first) {
            int steps = 0;
            for (long st = 1; st < a.ngroup; st <<= 1) ++steps;
            return steps == 0 ? first : ((steps & 1) ? a.ksA : a.ksB);
        };
        if (mode == MODE_SEG_REDUCE) {
            WC_LEVEL1(0);
            timed_launch(ctx, PGPS_K_FILTER_REDUCE, wc_reduce2<T, DP>, g2, blk, (unsigned)bytes(l_reduce2), a);
            const T* cur = a.agg2;
            T* nxt = a.ksA;
            for (long stride = 1; stride < a.ngroup; stride <<= 1) {
                { const int rk_ = ks_filter_level(stride, cur, nxt); if (rk_) return rk_; }
                cur = nxt;
                nxt = (nxt == a.ksA) ? a.ksB : a.ksA;
            }
            hipLaunchKernelGGL(rc::seg_pack_f<T>, dim3(1), dim3(256), 0, ctx->stream, d, cur + (long)(a.ngroup - 1) * nf, a.Fs,
                               a.Qs, sg->rec_f);
            HIPCHK(ctx, hipGetLastError());
            return PGPS_OK;
        }
        if (mode == MODE_SEG_FILTER) {
            const T* incl = ks_where(a.agg2);
            if (!a.seg_first) {
                hipLaunchKernelGGL(rc::seg_unpack_f<T>, dim3((unsigned)sg->rank), dim3(256), 0, ctx->stream, d, sg->gathered_f, rf,
                                   sg->tot_f);
                hipLaunchKernelGGL((wc_seg_carry_f<T, DP>), dim3(1), blk, (unsigned)bytes(l_carry3), ctx->stream, d, a.P0,
                                   (const T*)sg->tot_f, sg->rank, sg->carry_in);
                a.carry_in = sg->carry_in;
            }
            if (!a.seg_last) {
                HIPCHK(ctx, hipMemcpyAsync(sg->halo, sg->gathered_f + (long)(sg->rank + 1) * rf + nfp, 2 * (size_t)dd * sizeof(T),
                                           hipMemcpyDeviceToDevice, ctx->stream));
                a.halo_F = sg->halo;
                a.halo_Q = sg->halo + dd;
            }
            timed_launch(ctx, PGPS_K_FILTER_REDUCE, wc_fin_filter<T, DP>, g2, blk, (unsigned)bytes(l_carry3), a, incl);
            WC_LEVEL1(2);
            timed_launch(ctx, PGPS_K_SMOOTHER_REDUCE, wc_sreduce2<T, DP>, g2, blk, (unsigned)bytes(l_sred2), a);
            const T* cur = a.sagg2;
            T* nxt = a.ksA;
            for (long stride = 1; stride < a.ngroup; stride <<= 1) {
                { const int rk_ = ks_smoother_level(stride, cur, nxt); if (rk_) return rk_; }
                cur = nxt;
                nxt = (nxt == a.ksA) ? a.ksB : a.ksA;
            }
            // the suffix that starts at group 0 is the whole segment
            hipLaunchKernelGGL(rc::seg_pack_s<T>, dim3(1), dim3(256), 0, ctx->stream, d, cur, (const double*)a.llpart,
                               (long)a.nchunk, spad, sg->rec_s);
            HIPCHK(ctx, hipGetLastError());
            return PGPS_OK;
        }
        // MODE_SEG_SMOOTHER
        const T* sfx = ks_where(a.sagg2);
        if (!a.seg_last) {
            hipLaunchKernelGGL(rc::seg_unpack_s<T>, dim3((unsigned)sg->nranks), dim3(256), 0, ctx->stream, d, sg->gathered_s, rs,
                               sg->tot_s);
            hipLaunchKernelGGL((wc_seg_carry_s<T, DP>), dim3(1), blk, (unsigned)bytes(l_scarry3), ctx->stream, d,
                               (const T*)sg->tot_s, sg->rank, sg->nranks, sg->carry_back);
            a.carry_back = sg->carry_back;
            a.halo_F = sg->halo;
            a.halo_Q = sg->halo + dd;
            timed_launch(ctx, PGPS_K_SMOOTHER_REDUCE, wc_fin_smoother_cb<T, DP>, g2, blk, (unsigned)bytes(l_scarry3), a, sfx);
        } else {
            timed_launch(ctx, PGPS_K_SMOOTHER_REDUCE, wc_fin_smoother<T>, g2, blk, 0u, a, sfx);
        }
        WC_LEVEL1(3);
        if (sg->ll)
            hipLaunchKernelGGL(rc::seg_ll_sum<T>, dim3(1), dim3(64), 0, ctx->stream, sg->gathered_s, sg->nranks, rs, spad, sg->ll);
        HIPCHK(ctx, hipGetLastError());
        return PGPS_OK;
    }
    WC_LEVEL1(0);
    timed_launch(ctx, PGPS_K_FILTER_REDUCE, wc_reduce2<T, DP>, g2, blk, (unsigned)bytes(l_reduce2), a);
    // level 3: Kogge-Stone over the group totals, ping-pong agg2 -> ksA -> ksB -> ...; serial wc_carry3 / wc_scarry3
    // (one wave walking the groups) only when forced (PGPS_WC_SERIAL3, the cross-check of the tests)
    const bool serial3 = ctx->wc_serial3 != 0;
    if (serial3) {
        timed_launch(ctx, PGPS_K_FILTER_REDUCE, wc_carry3<T, DP>, dim3(1), blk, (unsigned)bytes(l_carry3), a);
    } else {
        const T* cur = a.agg2;
        T* nxt = a.ksA;
        for (long stride = 1; stride < a.ngroup; stride <<= 1) {
            { const int rk_ = ks_filter_level(stride, cur, nxt); if (rk_) return rk_; }
            cur = nxt;
            nxt = (nxt == a.ksA) ? a.ksB : a.ksA;
        }
        timed_launch(ctx, PGPS_K_FILTER_REDUCE, wc_fin_filter<T, DP>, g2, blk, (unsigned)bytes(l_carry3), a, cur);
    }
    if constexpr (sizeof(T) == 8) {
        if (gr) {
            // log-likelihood + the model's adjoints (pgps_gradlti.h): forward pass with the adjoint elements folded, the
            // smoother's scan over their totals, backward pass, finalize
            if (serial3 || mode != MODE_PKF) return PGPS_E_INVALID;
            const size_t l_gapply = 7 * GE::MSZ + 12 * DP + pad, l_gback = 9 * GE::MSZ + 14 * DP + pad;
            if (bytes(l_gback) > 160 * 1024) return PGPS_E_UNSUPPORTED_DIM;
            HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(wg_apply1<DP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes(l_gapply)));
            HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(wg_back1<DP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes(l_gback)));
            timed_launch(ctx, PGPS_K_FILTER_REDUCE, wc_enter1<T, DP>, g1, blk, (unsigned)bytes(l_carry3), a);
            timed_launch(ctx, PGPS_K_FILTER_APPLY, wg_apply1<DP>, g1, blk, (unsigned)bytes(l_gapply), a);
            timed_launch(ctx, PGPS_K_SMOOTHER_REDUCE, wc_sreduce2<T, DP>, g2, blk, (unsigned)bytes(l_sred2), a);
            const T* cur = a.sagg2;
            T* nxt = a.ksA;
            for (long stride = 1; stride < a.ngroup; stride <<= 1) {
                timed_launch(ctx, PGPS_K_SMOOTHER_REDUCE, wc_ks_smoother<T, DP>, g2, blk, (unsigned)bytes(l_sred2), a.d, (long)a.ngroup,
                             stride, cur, nxt);
                cur = nxt;
                nxt = (nxt == a.ksA) ? a.ksB : a.ksA;
            }
            timed_launch(ctx, PGPS_K_SMOOTHER_REDUCE, wc_fin_smoother<T>, g2, blk, 0u, a, cur);
            timed_launch(ctx, PGPS_K_SMOOTHER_REDUCE, wc_senter1<T, DP>, g1, blk, (unsigned)bytes(l_scarry3), a);
            timed_launch(ctx, PGPS_K_SMOOTHER_APPLY, wg_back1<DP>, g1, blk, (unsigned)bytes(l_gback), a, *gr);
            hipLaunchKernelGGL(k_grad_lti_finalize, dim3((unsigned)(1 + grad_lti_nstat(a.d))), dim3(256), 0, ctx->stream,
                               (long)a.nchunk, grad_lti_nstat(a.d), (const double*)a.llpart, (const double*)gr->gpart, gr->out);
            HIPCHK(ctx, hipGetLastError());
            return PGPS_OK;
        }
    }
    if (mode == MODE_PKFS) {
        WC_LEVEL1(2);
        timed_launch(ctx, PGPS_K_SMOOTHER_REDUCE, wc_sreduce2<T, DP>, g2, blk, (unsigned)bytes(l_sred2), a);
        if (serial3) {
            timed_launch(ctx, PGPS_K_SMOOTHER_REDUCE, wc_scarry3<T, DP>, dim3(1), blk, (unsigned)bytes(l_scarry3), a);
        } else {
            const T* cur = a.sagg2;
            T* nxt = a.ksA;
            for (long stride = 1; stride < a.ngroup; stride <<= 1) {
                { const int rk_ = ks_smoother_level(stride, cur, nxt); if (rk_) return rk_; }
                cur = nxt;
                nxt = (nxt == a.ksA) ? a.ksB : a.ksA;
            }
            timed_launch(ctx, PGPS_K_SMOOTHER_REDUCE, wc_fin_smoother<T>, g2, blk, 0u, a, cur);
            if (a.ll)
                timed_launch(ctx, PGPS_K_LL_FINALIZE, wc::wc_ll_finalize, dim3(1), blk, 0u, (const double*)a.llpart,
                             (long)a.nchunk, a.ll);
        }
        WC_LEVEL1(3);
    } else {
        WC_LEVEL1(1);
        if (a.ll)
            timed_launch(ctx, PGPS_K_LL_FINALIZE, wc::wc_ll_finalize, dim3(1), blk, 0u, (const double*)a.llpart,
                         (long)a.nchunk, a.ll);
    }
#undef WC_LEVEL1
    HIPCHK(ctx, hipGetLastError());
    return PGPS_OK;
}

template <typename T>
static int launch_scan_wc_impl(pgps_ctx* ctx, ScanArgs<T> sa, int d, Mode mode, GradLtiArgs* gr) {
    using namespace wc;
    if (mode == MODE_PKS) return PGPS_E_UNSUPPORTED_DIM;      // stand-alone pks: the other two families (d <= 16)
    const bool seg = mode == MODE_SEG_REDUCE || mode == MODE_SEG_FILTER || mode == MODE_SEG_SMOOTHER;
    if (d < 1 || d > 32) return PGPS_E_UNSUPPORTED_DIM;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    WcArgs<T> a{};
    a.N = sa.N; a.d = d;
    // one wave per chunk: about a thousand chunks before the chunks grow to 64 steps; groups of 4..64 chunks so that the
    // Kogge-Stone levels over the group totals stay at one round of waves (<= 512 groups up to 2^21 steps)
    // (the two-rows level-1 kernels of d >= 17 carry two chunks per wave: twice the chunks for a wave on every SIMD)
    const long waves = (rc2_covers<T>(d) && ctx->wc_rows2) ? 2048 : 1024;
    long lw = ctx->chunk > 0 ? ctx->chunk : (sa.N + waves - 1) / waves;
    if (ctx->chunk <= 0) lw = lw < 16 ? 16 : (lw > 64 ? 64 : lw);
    a.Lw = (int)lw;
    a.nchunk = (sa.N + a.Lw - 1) / a.Lw;
    long kg = (a.nchunk + 511) / 512;
    // (PGPS_WC_KGROUP_MIN: diagnostic, the smallest group; read once)
    static const long kgmin_env = [] { const char* e = std::getenv("PGPS_WC_KGROUP_MIN"); return e ? std::atol(e) : 0L; }();
    // (groups of ONE chunk up to 512 chunks: a serial in-group combine costs more than the Kogge-Stone level it saves --
    // 32 against 26 us at d = 18 -- the CO2 kernel at its 3192 points: ll 526 -> 495 us, ll + gradient 843 -> 808 us)
    const long kgmin = kgmin_env > 0 ? kgmin_env : 1;
    a.kgroup = (int)(kg < kgmin ? kgmin : (kg > kGroupMax ? kGroupMax : kg));
    a.ngroup = (int)((a.nchunk + a.kgroup - 1) / a.kgroup);
    a.P0 = sa.P0; a.H = sa.H; a.R = sa.R; a.Fs = sa.Fs; a.Qs = sa.Qs; a.ys = sa.ys;
    a.fms = sa.fms; a.fPs = sa.fPs; a.sms = sa.sms; a.sPs = sa.sPs; a.ll = seg ? nullptr : sa.ll;
    a.seg_first = seg ? (sa.rank == 0) : 1;
    a.seg_last = seg ? (sa.rank == sa.nranks - 1) : 1;
    const size_t dd = (size_t)d * d, nf = nfilt(d), ns = nsmth(d), nc = (size_t)a.nchunk, ng = (size_t)a.ngroup;
    size_t off = 0;
    const size_t o_agg1 = off;   off = wc_align(off + nc * nf * sizeof(T));
    const size_t o_lpre1 = off;  off = wc_align(off + nc * nf * sizeof(T));
    const size_t o_agg2 = off;   off = wc_align(off + ng * nf * sizeof(T));
    const size_t o_carry2 = off; off = wc_align(off + ng * (d + dd) * sizeof(T));
    const size_t o_sagg1 = off;  off = wc_align(off + nc * ns * sizeof(T));
    const size_t o_lsuf1 = off;  off = wc_align(off + nc * ns * sizeof(T));
    const size_t o_sagg2 = off;  off = wc_align(off + ng * ns * sizeof(T));
    const size_t o_sc2 = off;    off = wc_align(off + ng * (d + dd) * sizeof(T));
    const size_t o_ll = off;     off = wc_align(off + nc * sizeof(double));
    const size_t o_ksA = off;    off = wc_align(off + ng * nf * sizeof(T));
    const size_t o_ksB = off;    off = wc_align(off + ng * nf * sizeof(T));
    // segments: the same layout in all three phases (the workspace must not move between them)
    const size_t nr = seg ? (size_t)sa.nranks : 0;
    const size_t o_totf = off;   off = wc_align(off + nr * nf * sizeof(T));
    const size_t o_tots = off;   off = wc_align(off + nr * ns * sizeof(T));
    const size_t o_cin = off;    if (seg) off = wc_align(off + (d + dd) * sizeof(T));
    const size_t o_cback = off;  if (seg) off = wc_align(off + (d + dd) * sizeof(T));
    const size_t o_halo = off;   if (seg) off = wc_align(off + 2 * dd * sizeof(T));
    const size_t o_E = off;      if (seg) off = wc_align(off + (size_t)sa.N * dd * sizeof(T));
    const size_t o_en = off;     off = wc_align(off + nc * (d + dd) * sizeof(T));
    const size_t o_sen = off;    off = wc_align(off + nc * (d + dd) * sizeof(T));
    const size_t o_gp = off;     if (gr) off = wc_align(off + nc * (size_t)grad_lti_nstat(d) * sizeof(double));
    int rc = ensure(ctx, ctx->ws, off);
    if (rc) return rc;
    char* base = (char*)ctx->ws.p;
    if (gr) gr->gpart = (double*)(base + o_gp);
    a.agg1 = (T*)(base + o_agg1); a.lpre1 = (T*)(base + o_lpre1); a.agg2 = (T*)(base + o_agg2);
    a.carry2 = (T*)(base + o_carry2); a.sagg1 = (T*)(base + o_sagg1); a.lsuf1 = (T*)(base + o_lsuf1);
    a.sagg2 = (T*)(base + o_sagg2); a.scarry2 = (T*)(base + o_sc2); a.llpart = (double*)(base + o_ll);
    a.Es = seg ? (T*)(base + o_E) : a.sPs;
    a.enter1 = (T*)(base + o_en); a.senter1 = (T*)(base + o_sen);
    a.ksA = (T*)(base + o_ksA); a.ksB = (T*)(base + o_ksB);
    WcSeg<T> sgv{};
    if (seg) {
        sgv.rank = sa.rank; sgv.nranks = sa.nranks;
        sgv.rec_f = sa.rec_f; sgv.gathered_f = sa.gathered_f; sgv.rec_s = sa.rec_s; sgv.gathered_s = sa.gathered_s;
        sgv.tot_f = (T*)(base + o_totf); sgv.tot_s = (T*)(base + o_tots);
        sgv.carry_in = (T*)(base + o_cin); sgv.carry_back = (T*)(base + o_cback); sgv.halo = (T*)(base + o_halo);
        sgv.ll = sa.ll;
    }
    const WcSeg<T>* sg = seg ? &sgv : nullptr;
    if (d <= 8) return launch_scan_wc_dp<T, 8>(ctx, a, mode, sg, gr);
    if (d <= 12) return launch_scan_wc_dp<T, 12>(ctx, a, mode, sg, gr);
    if (d <= 16) return launch_scan_wc_dp<T, 16>(ctx, a, mode, sg, gr);
    if (d <= 18) return launch_scan_wc_dp<T, 18>(ctx, a, mode, sg, gr);
    if (d <= 24) return launch_scan_wc_dp<T, 24>(ctx, a, mode, sg, gr);
    return launch_scan_wc_dp<T, 32>(ctx, a, mode, sg, gr);
}
template <typename T>
int launch_scan_wc(pgps_ctx* ctx, ScanArgs<T> sa, int d, Mode mode) {
    return launch_scan_wc_impl<T>(ctx, sa, d, mode, nullptr);
}

int launch_ll_grad_lti_wc(pgps_ctx* ctx, long N, int d, const double* model, double R, const double* Fs, const double* Qs,
                          const double* ts, double t0, const double* ys, double* out) {
    RoctxRange range_("parallel_filter");
    if (!ctx || N < 1 || !model || !Fs || !Qs || !ts || !ys || !out) return PGPS_E_INVALID;
    if (d < 1 || d > 32) return PGPS_E_UNSUPPORTED_DIM;
    const size_t dd = (size_t)d * d, n = (size_t)N;
    int rcode;
    if ((rcode = ensure(ctx, ctx->lti[6], n * dd * sizeof(double)))) return rcode;
    if ((rcode = ensure(ctx, ctx->lti[7], n * d * sizeof(double)))) return rcode;
    ScanArgs<double> a{};
    a.N = N; a.seg_first = 1; a.seg_last = 1;
    a.P0 = model + dd; a.H = model + 2 * dd; a.R = R; a.Fs = Fs; a.Qs = Qs; a.ys = ys;
    a.fPs = (double*)ctx->lti[6].p; a.fms = (double*)ctx->lti[7].p;
    a.ll = nullptr;
    GradLtiArgs g{};
    g.N = N; g.d = d; g.ts = ts; g.t0 = t0; g.out = out;
    return launch_scan_wc_impl<double>(ctx, a, d, MODE_PKF, &g);
}

template <typename T, int DP>
static int launch_disc_wc_dp(pgps_ctx* ctx, long N, int d, const T* F, const T* Pinf, const T* ts, T t0, T* Fs, T* Qs) {
    const int spw = 8;
    const size_t lds = ((size_t)9 * wc::Geo<DP>::MSZ + (size_t)DP * wc::Geo<DP>::NRA + 64) * sizeof(double);
    if (lds > 160 * 1024) return PGPS_E_UNSUPPORTED_DIM;
    HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(wc::wc_discretise<T, DP>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const long grid = (N + spw - 1) / spw;
    timed_launch(ctx, PGPS_K_DISCRETISE, wc::wc_discretise<T, DP>, dim3((unsigned)grid), dim3(64), (unsigned)lds, N, d, spw,
                 F, Pinf, ts, t0, Fs, Qs);
    HIPCHK(ctx, hipGetLastError());
    return PGPS_OK;
}

template <typename T>
int launch_disc_wc(pgps_ctx* ctx, long N, int d, const T* F, const T* Pinf, const T* ts, T t0, T* Fs, T* Qs) {
    RoctxRange range_("make_model");
    if (d < 1 || d > 32) return PGPS_E_UNSUPPORTED_DIM;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (d <= 8) return launch_disc_wc_dp<T, 8>(ctx, N, d, F, Pinf, ts, t0, Fs, Qs);
    if (d <= 12) return launch_disc_wc_dp<T, 12>(ctx, N, d, F, Pinf, ts, t0, Fs, Qs);
    if (d <= 16) return launch_disc_wc_dp<T, 16>(ctx, N, d, F, Pinf, ts, t0, Fs, Qs);
    if (d <= 18) return launch_disc_wc_dp<T, 18>(ctx, N, d, F, Pinf, ts, t0, Fs, Qs);
    if (d <= 24) return launch_disc_wc_dp<T, 24>(ctx, N, d, F, Pinf, ts, t0, Fs, Qs);
    return launch_disc_wc_dp<T, 32>(ctx, N, d, F, Pinf, ts, t0, Fs, Qs);
}

template int launch_disc_wc<double>(pgps_ctx*, long, int, const double*, const double*, const double*, double, double*,
                                    double*);
template int launch_disc_wc<float>(pgps_ctx*, long, int, const float*, const float*, const float*, float, float*, float*);
template int launch_scan_wc<double>(pgps_ctx*, ScanArgs<double>, int, Mode);
template int launch_scan_wc<float>(pgps_ctx*, ScanArgs<float>, int, Mode);

// ====================================================================================================
// Host driver of the row-cooperative family (kernels: pgps_rc.hip.h, one instantiation per state dimension):
// level-1 reduce, Kogge-Stone steps over the chain totals, level-1 apply, Kogge-Stone over the smoothing totals,
// level-1 smoother.
// ====================================================================================================
namespace rc {

// sum of the chains' log-likelihood partials: one workgroup of 16 waves, fixed order (bit-reproducible)
static __global__ __launch_bounds__(1024) void ll_finalize(const double* llpart, long n, double* ll) {
    __shared__ double part[16];
    llpart += blockIdx.x * n;           // batched: one model per workgroup
    ll += blockIdx.x;
    double t = 0.0;
    // eight loads in flight per thread (one at a time, the sixteen dependent round trips of 16 384 chains were 9 us);
    // the order of the additions stays a function of n alone
    for (long c0 = threadIdx.x; c0 < n; c0 += 8 * 1024) {
        double v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const long c = c0 + j * 1024L; v[j] = c < n ? llpart[c] : 0.0; }
#pragma unroll
        for (int j = 0; j < 8; ++j) t += v[j];
    }
    t = wc::wave_sum(t);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < 16; ++w) s += part[w];
        *ll = s;
    }
}

template <typename Real>
static int level1(pgps_ctx* ctx, int d, const RcArgsT<Real>& a, int phase) {
    if constexpr (sizeof(Real) == 4) {
        if (a.quad) {                           // quad-cooperative level-1 kernels (pgps_qc.hip.h), same records
            switch (d) {
                case 5: return qc::launch_qc_level1<5>(ctx, a, phase);
                case 6: return qc::launch_qc_level1<6>(ctx, a, phase);
                case 7: return qc::launch_qc_level1<7>(ctx, a, phase);
                case 8: return qc::launch_qc_level1<8>(ctx, a, phase);
            }
            return PGPS_E_UNSUPPORTED_DIM;
        }
    }
    switch (d) {
#define PGPS_RC_CASE(DV) case DV: return launch_rc_level1<Real, DV>(ctx, a, phase);
        PGPS_RC_CASE(2) PGPS_RC_CASE(3) PGPS_RC_CASE(4) PGPS_RC_CASE(5) PGPS_RC_CASE(6) PGPS_RC_CASE(7) PGPS_RC_CASE(8)
        PGPS_RC_CASE(9) PGPS_RC_CASE(10) PGPS_RC_CASE(11) PGPS_RC_CASE(12) PGPS_RC_CASE(13) PGPS_RC_CASE(14)
        PGPS_RC_CASE(15) PGPS_RC_CASE(16)
#undef PGPS_RC_CASE
    }
    return PGPS_E_UNSUPPORTED_DIM;
}

template <typename Real>
static int ks_step(pgps_ctx* ctx, int d, int which, long n, long stride, const Real* in, Real* out, int batch = 1,
                   long bstride = 0, const Real* fixed = nullptr) {
    switch (d) {
#define PGPS_RC_CASE(DV) case DV: return launch_rc_ks<Real, DV>(ctx, which, n, stride, in, out, batch, bstride, fixed);
        PGPS_RC_CASE(2) PGPS_RC_CASE(3) PGPS_RC_CASE(4) PGPS_RC_CASE(5) PGPS_RC_CASE(6) PGPS_RC_CASE(7) PGPS_RC_CASE(8)
        PGPS_RC_CASE(9) PGPS_RC_CASE(10) PGPS_RC_CASE(11) PGPS_RC_CASE(12) PGPS_RC_CASE(13) PGPS_RC_CASE(14)
        PGPS_RC_CASE(15) PGPS_RC_CASE(16)
#undef PGPS_RC_CASE
    }
    return PGPS_E_UNSUPPORTED_DIM;
}

template <typename Real>
static int scan_blocked(pgps_ctx* ctx, int d, int which, long n, Real* data, Real* scratch) {
    switch (d) {
#define PGPS_RC_CASE(DV) case DV: return launch_rc_scan_blocked<Real, DV>(ctx, which, n, data, scratch);
        PGPS_RC_CASE(2) PGPS_RC_CASE(3) PGPS_RC_CASE(4) PGPS_RC_CASE(5) PGPS_RC_CASE(6) PGPS_RC_CASE(7) PGPS_RC_CASE(8)
        PGPS_RC_CASE(9) PGPS_RC_CASE(10) PGPS_RC_CASE(11) PGPS_RC_CASE(12) PGPS_RC_CASE(13) PGPS_RC_CASE(14)
        PGPS_RC_CASE(15) PGPS_RC_CASE(16)
#undef PGPS_RC_CASE
    }
    return PGPS_E_UNSUPPORTED_DIM;
}

// Which scan runs over n chain totals: the blocked one (in place, a handful of launches: pgps_rc.hip.h) or one launch per
// Kogge-Stone level (ping-pong between the two buffers).  A pure function of the context's setting and n, so the three
// phases of a segment pass agree on where the result lives.
static inline bool scan_is_blocked(const pgps_ctx* ctx, long n, int batch, int d) {
    return batch <= 1 && d <= kScanBlockedDimMax && (ctx->rc_scan == 1 || (ctx->rc_scan < 0 && n >= 64));
}
// inclusive scan of the n records in A (B: second buffer / scratch); *res = where the result is
template <typename Real>
static int ks_scan(pgps_ctx* ctx, int d, int which, long n, Real* A, Real* B, Real** res, int batch = 1, long bstride = 0) {
    if (scan_is_blocked(ctx, n, batch, d)) {
        *res = A;
        return scan_blocked(ctx, d, which, n, A, B);
    }
    Real *src = A, *dst = B;
    for (long s = 1; s < n; s *= 2) {
        int rcode = ks_step(ctx, d, which, n, s, src, dst, batch, bstride);
        if (rcode) return rcode;
        Real* t = src; src = dst; dst = t;
    }
    *res = src;
    return PGPS_OK;
}

template <typename Real>
static int seg_carry(pgps_ctx* ctx, int d, int which, const Real* gathered, int rank, int nranks, int reclen, Real* out) {
    switch (d) {
#define PGPS_RC_CASE(DV) case DV: return launch_rc_seg_carry<Real, DV>(ctx, which, gathered, rank, nranks, reclen, out);
        PGPS_RC_CASE(2) PGPS_RC_CASE(3) PGPS_RC_CASE(4) PGPS_RC_CASE(5) PGPS_RC_CASE(6) PGPS_RC_CASE(7) PGPS_RC_CASE(8)
        PGPS_RC_CASE(9) PGPS_RC_CASE(10) PGPS_RC_CASE(11) PGPS_RC_CASE(12) PGPS_RC_CASE(13) PGPS_RC_CASE(14)
        PGPS_RC_CASE(15) PGPS_RC_CASE(16)
#undef PGPS_RC_CASE
    }
    return PGPS_E_UNSUPPORTED_DIM;
}

template <typename Real>
struct SegInfo {
    int rank, nranks;
    Real* rec_f; const Real* gathered_f;
    Real* rec_s; const Real* gathered_s;
    Real *carry_rec, *cb_rec;         // scratch: compact carry-in / carry-back records
};

// number of Kogge-Stone steps over n records, and the buffer the result ends up in
template <typename Real>
static inline Real* ks_result(const pgps_ctx* ctx, int d, long n, Real* A, Real* B) {
    if (scan_is_blocked(ctx, n, 1, d)) return A;
    int steps = 0;
    for (long s = 1; s < n; s *= 2) ++steps;
    return (steps & 1) ? B : A;
}

// One phase of the segment protocol (pssgp/distributed.py): scratch (chain totals, their scans, the stored smoothing
// elements) stays in the context's workspace between the three calls of a pass.
template <typename Real>
static int scan_rc_seg(pgps_ctx* ctx, int d, RcArgsT<Real> a, Mode mode, Real* aggA, Real* aggB, Real* saggA, Real* saggB,
                       const SegInfo<Real>& sg, double* ll) {
    int rcode;
    const int nf = wc::nfilt(d), ns = wc::nsmth(d);
    const int rf = seg_rec_f_len(d), rs = seg_rec_s_len(d), pad = seg_rec_s_pad(d);
    const int nfp = d * d + d + d * (d + 1) + d;            // packed 5-tuple inside the filter record
    a.seg_first = sg.rank == 0;
    a.seg_last = sg.rank == sg.nranks - 1;
    if (mode == MODE_SEG_REDUCE) {
        a.agg1 = aggA;
        if ((rcode = level1(ctx, d, a, 0))) return rcode;
        Real* src = aggA;
        if ((rcode = ks_scan(ctx, d, 0, a.nchunk, aggA, aggB, &src))) return rcode;
        hipLaunchKernelGGL(seg_pack_f<Real>, dim3(1), dim3(256), 0, ctx->stream, d, (const Real*)(src + (a.nchunk - 1) * nf), a.Fs,
                           a.Qs, sg.rec_f);
        HIPCHK(ctx, hipGetLastError());
        return PGPS_OK;
    }
    if (mode == MODE_SEG_FILTER) {
        Real* pre = ks_result(ctx, d, a.nchunk, aggA, aggB);
        if (!a.seg_first) {
            // everything before this segment, combined into every local prefix (and the entry state of chain 0)
            if ((rcode = seg_carry(ctx, d, 0, sg.gathered_f, sg.rank, sg.nranks, rf, sg.carry_rec))) return rcode;
            Real* other = pre == aggA ? aggB : aggA;
            if ((rcode = ks_step(ctx, d, 0, a.nchunk, 0, pre, other, 1, 0, sg.carry_rec))) return rcode;
            pre = other;
            a.carry = sg.carry_rec;
        }
        if (!a.seg_last) {
            a.halo_F = sg.gathered_f + (long)(sg.rank + 1) * rf + nfp;
            a.halo_Q = a.halo_F + (long)d * d;
        }
        a.pre = pre;
        a.sagg1 = saggA;
        if ((rcode = level1(ctx, d, a, 1))) return rcode;
        Real* src = saggA;
        if ((rcode = ks_scan(ctx, d, 1, a.nchunk, saggA, saggB, &src))) return rcode;
        hipLaunchKernelGGL(seg_pack_s<Real>, dim3(1), dim3(256), 0, ctx->stream, d, (const Real*)src, (const double*)a.llpart,
                           (long)a.nchunk, pad, sg.rec_s);
        HIPCHK(ctx, hipGetLastError());
        return PGPS_OK;
    }
    // MODE_SEG_SMOOTHER
    Real* suf = ks_result(ctx, d, a.nchunk, saggA, saggB);
    if (!a.seg_last) {
        if ((rcode = seg_carry(ctx, d, 1, sg.gathered_s, sg.rank, sg.nranks, rs, sg.cb_rec))) return rcode;
        Real* other = suf == saggA ? saggB : saggA;
        if ((rcode = ks_step(ctx, d, 1, a.nchunk, 0, suf, other, 1, 0, sg.cb_rec))) return rcode;
        suf = other;
        a.carry_back = sg.cb_rec;
    }
    a.suf = suf;
    if ((rcode = level1(ctx, d, a, 3))) return rcode;
    if (ll) hipLaunchKernelGGL(seg_ll_sum<Real>, dim3(1), dim3(64), 0, ctx->stream, sg.gathered_s, sg.nranks, rs, pad, ll);
    HIPCHK(ctx, hipGetLastError());
    (void)ns;
    return PGPS_OK;
}

template <typename Real>
static int scan_rc(pgps_ctx* ctx, int d, RcArgsT<Real> a, Mode mode, Real* aggA, Real* aggB, Real* saggA, Real* saggB,
                   double* ll) {
    int rcode;
    if (mode == MODE_PKS) {                     // stand-alone smoother: elements from the given filtered moments
        a.sagg1 = saggA;
        if ((rcode = level1(ctx, d, a, 5))) return rcode;
        Real* ssrc = saggA;
        if ((rcode = ks_scan(ctx, d, 1, a.nchunk, saggA, saggB, &ssrc))) return rcode;
        a.suf = ssrc;
        return level1(ctx, d, a, 3);
    }
    a.agg1 = aggA;
    if ((rcode = level1(ctx, d, a, 0))) return rcode;
    Real* src = aggA;
    const int nb = a.batch > 1 ? a.batch : 1;
    if ((rcode = ks_scan(ctx, d, 0, a.nchunk, aggA, aggB, &src, nb, a.bs_agg))) return rcode;
    a.pre = src;
    if (mode == MODE_PKFS) {
        a.sagg1 = saggA;
        if ((rcode = level1(ctx, d, a, 1))) return rcode;
        src = saggA;
        if ((rcode = ks_scan(ctx, d, 1, a.nchunk, saggA, saggB, &src))) return rcode;
        a.suf = src;
        if ((rcode = level1(ctx, d, a, a.qslot ? 4 : 3))) return rcode;
    } else {
        if ((rcode = level1(ctx, d, a, 2))) return rcode;
    }
    if (ll)
        timed_launch(ctx, PGPS_K_LL_FINALIZE, ll_finalize, dim3((unsigned)nb), dim3(1024), 0u, (const double*)a.llpart,
                     (long)a.nchunk, ll);
    HIPCHK(ctx, hipGetLastError());
    return PGPS_OK;
}

}  // namespace rc

int launch_disc_rc(pgps_ctx* ctx, long N, int d, const double* F, const double* Pinf, const double* ts, double t0,
                   double* Fs, double* Qs, int batch, long bs_model) {
    RoctxRange range_("make_model");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    switch (d) {
#define PGPS_RC_CASE(DV) case DV: return rc::launch_rc_disc<DV>(ctx, N, F, Pinf, ts, t0, Fs, Qs, batch, bs_model);
        PGPS_RC_CASE(2) PGPS_RC_CASE(3) PGPS_RC_CASE(4) PGPS_RC_CASE(5) PGPS_RC_CASE(6) PGPS_RC_CASE(7) PGPS_RC_CASE(8)
        PGPS_RC_CASE(9) PGPS_RC_CASE(10) PGPS_RC_CASE(11) PGPS_RC_CASE(12) PGPS_RC_CASE(13) PGPS_RC_CASE(14)
        PGPS_RC_CASE(15) PGPS_RC_CASE(16)
#undef PGPS_RC_CASE
    }
    return PGPS_E_UNSUPPORTED_DIM;
}

static inline size_t rc_align(size_t x) { return (x + 255) / 256 * 256; }

template <typename Real>
static int scan_rc_entry(pgps_ctx* ctx, ScanArgs<Real> sa, int d, Mode mode, int store_f, const int* qslot, Real* pmean,
                         Real* pvar, int batch = 1, long bs_model = 0);

// 2 <= d <= 16, pkf / pks / pkfs on one device and the three segment phases
template <typename Real>
int launch_scan_rc(pgps_ctx* ctx, ScanArgs<Real> sa, int d, Mode mode) {
    return scan_rc_entry<Real>(ctx, sa, d, mode, 1, nullptr, nullptr, nullptr);
}
template int launch_scan_rc<double>(pgps_ctx*, ScanArgs<double>, int, Mode);
template int launch_scan_rc<float>(pgps_ctx*, ScanArgs<float>, int, Mode);
int launch_scan_rc_proj(pgps_ctx* ctx, ScanArgs<double> sa, int d, Mode mode, const int* qslot, double* pmean, double* pvar) {
    RoctxRange range_("parallel_filter");
    if (mode == MODE_PKFS && (!qslot || !pmean || !pvar)) return PGPS_E_INVALID;
    return scan_rc_entry<double>(ctx, sa, d, mode, 0, mode == MODE_PKFS ? qslot : nullptr, pmean, pvar);
}

int launch_ll_batch_rc(pgps_ctx* ctx, long N, int d, int batch, const double* table, long bs_model, const double* Fs,
                       const double* Qs, const double* ys, double* ll) {
    RoctxRange range_("parallel_filter");
    if (batch < 1 || !table || !Fs || !ys || !ll) return PGPS_E_INVALID;      // Qs == nullptr: implicit process noise
    const long dd = (long)d * d;
    ScanArgs<double> a{};
    a.N = N; a.seg_first = 1; a.seg_last = 1;
    a.P0 = table + dd; a.H = table + 2 * dd; a.R = 0.0; a.Fs = Fs; a.Qs = Qs; a.ys = ys;
    a.ll = ll;
    // R of model b sits at table[b * bs_model + 2 dd + d]; scan_rc_entry turns that into RcArgs::Rs
    a.carry_in = table + 2 * dd + d;
    return scan_rc_entry<double>(ctx, a, d, MODE_PKF, 0, nullptr, nullptr, nullptr, batch, bs_model);
}

template <typename Real>
static int scan_rc_entry(pgps_ctx* ctx, ScanArgs<Real> sa, int d, Mode mode, int store_f, const int* qslot, Real* pmean,
                         Real* pvar, int batch, long bs_model) {
    const bool seg = mode == MODE_SEG_REDUCE || mode == MODE_SEG_FILTER || mode == MODE_SEG_SMOOTHER;
    if (d < rc::kDimMin || d > rc::kDimMax) return PGPS_E_UNSUPPORTED_DIM;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    rc::RcArgsT<Real> a{};
    a.N = sa.N;
    // quad-cooperative level-1 kernels: fp32, 5 <= d <= 8, filter and filter + smoother (family 4)
    // (automatic at d = 8, where half of a 16-lane row idles: 0.91 against 1.10 ms at 2^20 steps; odd d goes lane by
    // lane and loses.  d = 6 reaches this driver only when the caller -- dispatch_scan in pgps_core.hip -- prefers it to
    // the lane-chunk kernels: profiles/r03_experiments.txt)
    const bool quad = sizeof(Real) == 4 && (ctx->family == 4 || (ctx->family == 0 && (d == 8 || d == 6))) &&
                      d >= qc::kDimMin && d <= qc::kDimMax && mode != MODE_PKS && batch <= 1 && bs_model == 0 && !qslot;
    a.quad = quad ? 1 : 0;
    if (ctx->chunk > 0) {
        // (the wide accesses address a workgroup's chains with 32-bit offsets: 4096 steps per chain keep the span of
        // sixteen d = 8 fp32 records, or of four d = 16 fp64 ones, far below 2^31 bytes)
        a.Lw = ctx->chunk > 4096 ? 4096 : ctx->chunk;
    } else if (quad) {
        // sixteen chains per wave: 64 steps per chain put one wave on every SIMD at 2^20 steps
        long lw = (sa.N + 16383) / 16384;
        a.Lw = (int)(lw < 8 ? 8 : lw > 64 ? 64 : lw);
    } else {
        // four chains per wave: 4096 chains put one wave on every SIMD, and every doubling adds a Kogge-Stone
        // launch to both scans.  Measured at d = 11: 2^17 steps 0.67 ms with 4096 chains against 0.79 ms with 8192;
        // 2^20 steps 3.57 ms with 8192 (two waves per SIMD where the registers allow) against 3.73 ms with 4096.
        // From d = 10 the filter + smoother with every moment written is better off with 4096 chains up to 2^20 steps
        // (d = 11: 3.24 -> 3.11 ms, d = 15: 5.45 -> 5.19 ms; d = 6 loses 12 %, the log-likelihood-only and projected
        // calls lose 2-10 %, so they keep 8192).
        const long cap = (d >= 10 && store_f && !qslot && mode == MODE_PKFS) ? 256 : 128;
        long lw = (sa.N + 4095) / 4096;
        if (lw > cap) { lw = (sa.N + 8191) / 8192; if (lw < 128) lw = 128; }     // 8192 chains, two full rounds of blocks
        if (batch > 1) {                        // the models multiply the chains: keep about 8192 in flight
            lw = ((long)sa.N * batch + 8191) / 8192;
            if (lw > 128) lw = 128;
        }
        a.Lw = (int)(lw < 8 ? 8 : lw > 512 ? 512 : lw);
    }
    a.nchunk = (sa.N + a.Lw - 1) / a.Lw;
    // (w + 1) 4 Lw + 2 <= N: the wide loads of a FAST wave reach at most 8 bytes into the record after its halo step
    a.wfast = sa.N >= 2 ? (sa.N - 2) / (4L * a.Lw) : 0;
    a.P0 = sa.P0; a.H = sa.H; a.R = sa.R; a.Fs = sa.Fs; a.Qs = sa.Qs; a.ys = sa.ys;
    a.fms = sa.fms; a.fPs = sa.fPs; a.sms = sa.sms; a.sPs = sa.sPs;
    a.store_f = store_f; a.qslot = qslot; a.pmean = pmean; a.pvar = pvar;
    // whole-series filter + smoother with every moment stored: the chains' smoothing totals in innovation form (pgps_rc.hip.h,
    // apply1_body DFORM); -DPGPS_RC_DFORM=0 keeps the reference's form everywhere (A/B)
#ifndef PGPS_RC_DFORM
#define PGPS_RC_DFORM 1
#endif
    a.dform = (PGPS_RC_DFORM != 0 && mode == MODE_PKFS && !seg && store_f && !quad && !qslot && batch <= 1 && sa.Qs != nullptr) ? 1 : 0;
    a.seg_first = 1; a.seg_last = 1;
    a.implicit_q = (sa.Qs == nullptr) ? 1 : 0;
    // (Qs may be absent where nothing per step is written: the log-likelihood call and the projected smoother)
    if (a.implicit_q && (store_f || (mode != MODE_PKF && !(mode == MODE_PKFS && qslot)))) return PGPS_E_INVALID;
    // projected-posterior calls come from the general-LTI entry points, whose Qs is Pinf - F Pinf F^T by
    // construction: the reduce pass need not read it
    if (qslot && !a.implicit_q) a.implicit_q = 2;
    const size_t dd = (size_t)d * d, nf = wc::nfilt(d), ns = wc::nsmth(d), nc = (size_t)a.nchunk;
    const size_t nbm = batch > 1 ? (size_t)batch : 1;
    if (bs_model > 0) {                                             // the batch entry point, B >= 1
        if (mode != MODE_PKF || store_f) return PGPS_E_INVALID;     // batched: log-likelihood only
        a.batch = batch; a.bs_F = (long)sa.N * (long)dd; a.bs_agg = (long)(nc * nf); a.bs_model = bs_model;
        a.Rs = sa.carry_in;
    }
    size_t off = 0;
    const size_t o_aggA = off;  off = rc_align(off + nbm * nc * nf * sizeof(Real));
    const size_t o_aggB = off;  off = rc_align(off + nbm * nc * nf * sizeof(Real));
    const size_t o_sagA = off;  off = rc_align(off + nc * ns * sizeof(Real));
    const size_t o_sagB = off;  off = rc_align(off + nc * ns * sizeof(Real));
    const size_t o_ll = off;    off = rc_align(off + nbm * nc * sizeof(double));
    const size_t o_cf = off;    off = rc_align(off + nf * sizeof(Real));
    const size_t o_cs = off;    off = rc_align(off + ns * sizeof(Real));
    const size_t o_L = off;     if (mode != MODE_PKF) off = rc_align(off + (size_t)sa.N * dd * sizeof(Real));
    // segments: the smoothing elements wait in scratch until the smoother phase brings sms / sPs
    const size_t o_E = off;     if (seg) off = rc_align(off + (size_t)sa.N * dd * sizeof(Real));
    const size_t o_g = off;     if (seg) off = rc_align(off + (size_t)sa.N * d * sizeof(Real));
    int rcode = ensure(ctx, ctx->ws, off);
    if (rcode) return rcode;
    char* base = (char*)ctx->ws.p;
    a.llpart = (double*)(base + o_ll);
    a.Lws = (Real*)(base + o_L);
    a.Es = seg ? (Real*)(base + o_E) : a.sPs;
    a.gs = seg ? (Real*)(base + o_g) : a.sms;
    Real* aggA = (Real*)(base + o_aggA); Real* aggB = (Real*)(base + o_aggB);
    Real* sagA = (Real*)(base + o_sagA); Real* sagB = (Real*)(base + o_sagB);
    if (seg) {
        rc::SegInfo<Real> sg{};
        sg.rank = sa.rank; sg.nranks = sa.nranks;
        sg.rec_f = sa.rec_f; sg.gathered_f = sa.gathered_f; sg.rec_s = sa.rec_s; sg.gathered_s = sa.gathered_s;
        sg.carry_rec = (Real*)(base + o_cf); sg.cb_rec = (Real*)(base + o_cs);
        return rc::scan_rc_seg(ctx, d, a, mode, aggA, aggB, sagA, sagB, sg, sa.ll);
    }
    return rc::scan_rc(ctx, d, a, mode, aggA, aggB, sagA, sagB, sa.ll);
}

// ====================================================================================================
// Log-likelihood and the model's adjoints (pgps_gradlti.h): discretise -> rc_reduce1 -> scan of the filter totals ->
// forward pass with the adjoint elements folded (rc_gapply1) -> suffix scan of the adjoint totals (the smoother's scan
// kernels: the reverse sweep composes under the smoothing operator) -> backward pass (rc_gback1) -> finalize.
// Everything is enqueued on the context's stream; `out` holds 1 + d d + 2 d + 1 doubles.
// ====================================================================================================
static int grad_level1_rc(pgps_ctx* ctx, int d, const GradLtiArgs& g, int phase) {
    switch (d) {
#define PGPS_RC_CASE(DV) case DV: return rc::launch_rc_grad<DV>(ctx, g, phase);
        PGPS_RC_CASE(2) PGPS_RC_CASE(3) PGPS_RC_CASE(4) PGPS_RC_CASE(5) PGPS_RC_CASE(6) PGPS_RC_CASE(7) PGPS_RC_CASE(8)
        PGPS_RC_CASE(9) PGPS_RC_CASE(10) PGPS_RC_CASE(11) PGPS_RC_CASE(12) PGPS_RC_CASE(13) PGPS_RC_CASE(14)
        PGPS_RC_CASE(15) PGPS_RC_CASE(16)
#undef PGPS_RC_CASE
    }
    return PGPS_E_UNSUPPORTED_DIM;
}

int launch_ll_grad_lti(pgps_ctx* ctx, long N, int d, const double* model, double R, const double* ts, double t0,
                       const double* ys, double* out) {
    RoctxRange range_("parallel_filter");
    if (!ctx || N < 1 || !model || !ts || !ys || !out) return PGPS_E_INVALID;
    if (d < rc::kDimMin || d > rc::kDimMax) return PGPS_E_UNSUPPORTED_DIM;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t dd = (size_t)d * d, n = (size_t)N;
    int rcode;
    // transition matrices and the filtered moments the backward pass re-reads: the general-LTI scratch of the context
    if ((rcode = ensure(ctx, ctx->lti[4], n * dd * sizeof(double)))) return rcode;
    if ((rcode = ensure(ctx, ctx->lti[6], n * dd * sizeof(double)))) return rcode;
    if ((rcode = ensure(ctx, ctx->lti[7], n * d * sizeof(double)))) return rcode;
    double* Fs = (double*)ctx->lti[4].p;
    if ((rcode = launch_disc_rc(ctx, N, d, model, model + dd, ts, t0, Fs, nullptr))) return rcode;     // implicit process noise
    rc::RcArgsT<double> a{};
    a.N = N;
    {
        long lw = ctx->chunk > 0 ? (ctx->chunk > 4096 ? 4096 : ctx->chunk) : (N + 4095) / 4096;
        if (ctx->chunk <= 0 && lw > 128) { lw = (N + 8191) / 8192; if (lw < 128) lw = 128; }
        a.Lw = (int)(lw < 8 && ctx->chunk <= 0 ? 8 : lw > 512 && ctx->chunk <= 0 ? 512 : lw);
    }
    a.nchunk = (N + a.Lw - 1) / a.Lw;
    a.wfast = N >= 2 ? (N - 2) / (4L * a.Lw) : 0;
    a.P0 = model + dd; a.H = model + 2 * dd; a.R = R; a.Fs = Fs; a.Qs = nullptr; a.ys = ys;
    a.seg_first = 1; a.seg_last = 1; a.implicit_q = 1; a.store_f = 0;
    const size_t nf = wc::nfilt(d), ns = wc::nsmth(d), nc = (size_t)a.nchunk, nst = (size_t)grad_lti_nstat(d);
    size_t off = 0;
    const size_t o_aggA = off;  off = rc_align(off + nc * nf * sizeof(double));
    const size_t o_aggB = off;  off = rc_align(off + nc * nf * sizeof(double));
    const size_t o_sagA = off;  off = rc_align(off + nc * ns * sizeof(double));
    const size_t o_sagB = off;  off = rc_align(off + nc * ns * sizeof(double));
    const size_t o_ll = off;    off = rc_align(off + nc * sizeof(double));
    const size_t o_gp = off;    off = rc_align(off + nc * nst * sizeof(double));
    if ((rcode = ensure(ctx, ctx->ws, off))) return rcode;
    char* base = (char*)ctx->ws.p;
    double *aggA = (double*)(base + o_aggA), *aggB = (double*)(base + o_aggB);
    double *sagA = (double*)(base + o_sagA), *sagB = (double*)(base + o_sagB);
    a.llpart = (double*)(base + o_ll);
    a.agg1 = aggA;
    if ((rcode = rc::level1(ctx, d, a, 0))) return rcode;
    double* pre = aggA;
    if ((rcode = rc::ks_scan(ctx, d, 0, a.nchunk, aggA, aggB, &pre))) return rcode;
    GradLtiArgs g{};
    g.N = N; g.d = d; g.Lw = a.Lw; g.nchunk = a.nchunk;
    g.Pinf = model + dd; g.H = model + 2 * dd; g.R = R; g.Fs = Fs; g.ys = ys; g.ts = ts; g.t0 = t0;
    g.fPs = (double*)ctx->lti[6].p; g.fms = (double*)ctx->lti[7].p;
    g.pre = pre; g.sagg = sagA; g.llpart = a.llpart; g.gpart = (double*)(base + o_gp); g.out = out;
    if ((rcode = grad_level1_rc(ctx, d, g, 0))) return rcode;
    double* suf = sagA;
    if ((rcode = rc::ks_scan(ctx, d, 1, a.nchunk, sagA, sagB, &suf))) return rcode;
    g.suf = suf;
    if ((rcode = grad_level1_rc(ctx, d, g, 1))) return rcode;
    return grad_level1_rc(ctx, d, g, 2);
}

}  // namespace pgps
