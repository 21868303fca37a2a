// pgps_rc2.hip.h -- "two-rows" level-1 kernels of the d <= 32 family: state dimensions 17..32 with the products in
// registers.
//
// A chain of time steps (one level-1 chunk) is owned by 32 lanes -- two 16-lane DPP rows -- and a wave carries two
// chains.  Lane i of a chain holds ROW i of every DP x DP operand (DP registers; DP is the state dimension itself
// from 18 on -- one instantiation each, no padding -- and 18 for d = 17), a distributed vector has element i in lane i.  All products are v_fmac_*_dpp with a `row_newbcast`
// operand, exactly the instruction of the row-cooperative family (pgps_rc.hip.h, d <= 16) -- the broadcast only
// reaches the 16 lanes of a DPP row, so the operand that is broadcast is first SPLIT: `lo` holds rows 0..15 of it in
// both DPP rows of the chain, `hi` rows 16..31 (one v_permlane16_swap per 32-bit half, gfx950).  Then
//     (A B)_i   = sum_{k<16} A_i[k] bcast_k(lo) + sum_{k<DP-16} A_i[16+k] bcast_k(hi)      (blocks of pgps_rc_asm.h)
//     (A B^T)_i[j] = sum_k A_i[k] bcast_j(lo|hi [k])                                        (pgps_rc2_asm.h)
// so neither product touches LDS; LDS carries the transposes (symmetrisations, the smoother gain) and the pivot row
// of the gain's elimination.  The wave-cooperative kernels this replaces (pgps_wc.hip: a lane grid of register tiles,
// operands in LDS) are bound by the CU's 128 B/clk of LDS: tools/micro/wc_mm.hip.
//
// Record formats, workspace and levels 2 and 3 are those of pgps_wc.hip (WcArgs); the state entering a chunk comes
// from wc_enter1 / wc_senter1 there.  Same algorithm as the reference's parallel filter and smoother elements
// (pssgp/kalman/parallel.py:155-184) chunked as in DESIGN.md section 4b.
//
// Both chains of a wave execute every instruction: a DPP operand read from a lane that EXEC has switched off is not
// defined, so nothing that contains an asm block sits under a lane- or chain-dependent branch -- a chain that has run
// out of steps, the first step of a series and a missing observation are all handled by selecting results.
#pragma once
#include <hip/hip_runtime.h>

#include "pgps_rc2_asm.h"
#include "pgps_rc_asm.h"
#include "pgps_wc_args.h"

namespace pgps {
namespace rc2 {

using wc::nfilt;
using wc::nsmth;
using wc::WcArgs;

// one wave per workgroup: LDS is in order for a wave, only the compiler has to be held back
__device__ __forceinline__ void wsync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ double rcp_nr(double x) {    // v_rcp_f64 and two Newton steps (pivots and innovation variances)
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ float rcp_nr(float x) {
    float r = __builtin_amdgcn_rcpf(x);
    r = __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
    return r;
}

// lo: the value the lane's partner in DPP row 0 (2) of the chain holds, hi: ... in DPP row 1 (3)
__device__ __forceinline__ void split(float b, float& lo, float& hi) {
    const unsigned u = __builtin_bit_cast(unsigned, b);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    lo = __builtin_bit_cast(float, (unsigned)r[0]);
    hi = __builtin_bit_cast(float, (unsigned)r[1]);
}
__device__ __forceinline__ void split(double b, double& lo, double& hi) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, b);
    const unsigned l = (unsigned)u, h = (unsigned)(u >> 32);
    const auto rl = __builtin_amdgcn_permlane16_swap(l, l, false, false);
    const auto rh = __builtin_amdgcn_permlane16_swap(h, h, false, false);
    lo = __builtin_bit_cast(double, (unsigned long long)(unsigned)rl[0] | ((unsigned long long)(unsigned)rh[0] << 32));
    hi = __builtin_bit_cast(double, (unsigned long long)(unsigned)rl[1] | ((unsigned long long)(unsigned)rh[1] << 32));
}

template <typename T, int DP>
struct Ops {
    static_assert(DP > 16 && DP <= 32, "two DPP rows per chain");
    static constexpr int HI = DP - 16;
    static constexpr int TLD = DP + 1;                  // leading dimension of the transposition buffer
    static constexpr int TLN = DP * TLD;
    using A16 = rc::Asm<T, 16>;
    using AH = rc::Asm<T, HI>;

    static __device__ __forceinline__ void split_mat(const T* b, T* lo, T* hi) {
#pragma unroll
        for (int j = 0; j < DP; ++j) split(b[j], lo[j], hi[j]);
    }
    static __device__ __forceinline__ void zero(T* z) {
#pragma unroll
        for (int j = 0; j < DP; ++j) z[j] = T(0);
    }
    static __device__ __forceinline__ void copy(T* z, const T* x) {
#pragma unroll
        for (int j = 0; j < DP; ++j) z[j] = x[j];
    }
    static __device__ __forceinline__ void select(T* z, bool take, const T* x) {       // z = take ? x : z
#pragma unroll
        for (int j = 0; j < DP; ++j) z[j] = take ? x[j] : z[j];
    }

    // C += A B, B split
    template <int J0 = 0>
    static __device__ __forceinline__ void mm0(T* C, const T* A, const T* Blo, const T* Bhi) {
        if constexpr (J0 + 4 <= DP) {
            A16::rows4(C + J0, Blo + J0, A);
            AH::rows4(C + J0, Bhi + J0, A + 16);
            mm0<J0 + 4>(C, A, Blo, Bhi);
        } else if constexpr (DP - J0 == 3) {
            A16::rows3(C + J0, Blo + J0, A);
            AH::rows3(C + J0, Bhi + J0, A + 16);
        } else if constexpr (DP - J0 == 2) {
            A16::rows2(C + J0, Blo + J0, A);
            AH::rows2(C + J0, Bhi + J0, A + 16);
        } else if constexpr (DP - J0 == 1) {
            A16::rows1(C + J0, Blo + J0, A);
            AH::rows1(C + J0, Bhi + J0, A + 16);
        }
    }

    // c[0..NJ) += sum_{k<DP} bcast_{J0+jj}(b[k]) a[k]
    template <int NJ, int J0, int K0 = 0>
    static __device__ __forceinline__ void cols_k(T* c, const T* b, const T* a) {
        if constexpr (K0 < DP) {
            constexpr int KC = (DP - K0 < 8) ? DP - K0 : 8;
            Cols<T, NJ, KC>::template run<J0>(c, b + K0, a + K0);
            cols_k<NJ, J0, K0 + KC>(c, b, a);
        }
    }
    template <int J0 = 0>
    static __device__ __forceinline__ void mm1_lo(T* C, const T* A, const T* Blo) {
        if constexpr (J0 < 16) {
            cols_k<4, J0>(C + J0, Blo, A);
            mm1_lo<J0 + 4>(C, A, Blo);
        }
    }
    template <int J0 = 0>
    static __device__ __forceinline__ void mm1_hi(T* C, const T* A, const T* Bhi) {
        if constexpr (J0 < HI) {
            constexpr int NJ = (HI - J0 < 4) ? HI - J0 : 4;
            cols_k<NJ, J0>(C + 16 + J0, Bhi, A);
            mm1_hi<J0 + NJ>(C, A, Bhi);
        }
    }
    // C += A B^T, B split
    static __device__ __forceinline__ void mm1(T* C, const T* A, const T* Blo, const T* Bhi) {
        mm1_lo(C, A, Blo);
        mm1_hi(C, A, Bhi);
    }

    // (A x)_i for a distributed vector x
    static __device__ __forceinline__ T mv(const T* A, T x) {
        T xl, xh, a0 = T(0), a1 = T(0);
        split(x, xl, xh);
        A16::mv(a0, a1, xl, A);
        AH::mv(a0, a1, xh, A + 16);
        return a0 + a1;
    }
    // sum over the chain of a distributed value (every lane gets it)
    static __device__ __forceinline__ T chain_sum(T p) {
        T pl, ph, a0 = T(0), a1 = T(0);
        split(p, pl, ph);
        A16::rsum(a0, a1, pl + ph, T(1));
        return a0 + a1;
    }
    // Z_i[j] += p_j q_i (p, q distributed)
    static __device__ __forceinline__ void rank1(T* Z, T p, T q) {
        T pl, ph;
        split(p, pl, ph);
        A16::rank1(Z, pl, q);
        AH::rank1(Z + 16, ph, q);
    }

    // at = (the matrix whose rows the lanes hold)^T through the chain's LDS buffer tl (TLN values)
    static __device__ __forceinline__ void transpose(const T* a, T* at, T* tl, int i) {
        if (i < DP) {
#pragma unroll
            for (int j = 0; j < DP; ++j) tl[i * TLD + j] = a[j];
        }
        wsync();
        const int ii = i < DP ? i : 0;
#pragma unroll
        for (int j = 0; j < DP; ++j) at[j] = tl[j * TLD + ii];
        wsync();
    }
    static __device__ __forceinline__ void symmetrise(T* a, T* tl, int i) {
        T at[DP];
        transpose(a, at, tl, i);
#pragma unroll
        for (int j = 0; j < DP; ++j) a[j] = T(0.5) * (a[j] + at[j]);
    }

    // row i of a compact (d x d) matrix in global memory; zero padding.  Every lane reads valid addresses (clamped
    // indices) and the padding is selected afterwards: a load under a lane-dependent condition becomes a branch of
    // its own with its own wait for memory, DP of them one after the other.
    //
    // FULL (d == DP, e.g. the CO2 kernel's d = 18): no column is padding, and the lanes beyond row d - 1 are left with
    // a copy of row 0 instead of zeros -- no select at all.  Nothing an active lane reads comes from those lanes (the
    // broadcasts name lanes below DP, the transposes and stores are guarded by the row, a chain sum multiplies their
    // term by a zero of h) and what they compute stays finite, being the arithmetic of a real row.
    template <bool FULL = false>
    static __device__ __forceinline__ void ld_row(const T* g, int d, int i, T* a) {
        const bool in = i < d;
        const T* r = g + (in ? i : 0) * d;
        if constexpr (FULL) {
#pragma unroll
            for (int j = 0; j < DP; ++j) a[j] = r[j];
        } else {
            T v[DP];
#pragma unroll
            for (int j = 0; j < DP; ++j) v[j] = r[j < d ? j : 0];
#pragma unroll
            for (int j = 0; j < DP; ++j) a[j] = (in && j < d) ? v[j] : T(0);
        }
    }
    // ... of its symmetric part
    template <bool FULL = false>
    static __device__ __forceinline__ void ld_row_sym(const T* g, int d, int i, T* a) {
        const bool in = i < d;
        const int ii = in ? i : 0;
        T v[DP], w[DP];
#pragma unroll
        for (int j = 0; j < DP; ++j) {
            const int jj = (FULL || j < d) ? j : 0;
            v[j] = g[ii * d + jj];
            w[j] = g[jj * d + ii];
        }
#pragma unroll
        for (int j = 0; j < DP; ++j) a[j] = (FULL || (in && j < d)) ? T(0.5) * (v[j] + w[j]) : T(0);
    }
    template <bool FULL = false>
    static __device__ __forceinline__ void st_row(T* g, int d, int i, bool ok, const T* a) {
        if (ok && i < d) {
            T* r = g + i * d;
#pragma unroll
            for (int j = 0; j < DP; ++j)
                if (FULL || j < d) r[j] = a[j];
        }
    }

    // B <- M^-1 B for a symmetric positive definite M (destroyed): Gauss-Jordan without pivoting and without scaling
    // the pivot rows (one multiply-add per entry: the rows are divided by their diagonal at the end).  The pivot row
    // travels through the chain's LDS buffer rb (2 DP values), which every lane reads at the same address.
    template <int C = 0>
    static __device__ __forceinline__ void solve(int d, T* M, T* B, T& diag, T* rb, int i) {
        if constexpr (C < DP) {
            if (C < d) {
                if (i == C) {
#pragma unroll
                    for (int j = C; j < DP; ++j) rb[j] = M[j];
#pragma unroll
                    for (int j = 0; j < DP; ++j) rb[DP + j] = B[j];
                }
                wsync();
                const T piv = rb[C];
                const T inv = rcp_nr(piv);
                diag = (i == C) ? piv : diag;
                const T fs = (i == C) ? T(0) : M[C] * inv;
#pragma unroll
                for (int j = C + 1; j < DP; ++j) M[j] = __builtin_fma(-fs, rb[j], M[j]);
#pragma unroll
                for (int j = 0; j < DP; ++j) B[j] = __builtin_fma(-fs, rb[DP + j], B[j]);
                wsync();
                solve<C + 1>(d, M, B, diag, rb, i);
            }
        }
    }
};

__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }

// ====================================================================================================
// level 1: reduce -- the chunk's filtering total (wc_reduce1)
// ====================================================================================================
template <typename T, int DP, bool FULL>
__global__ __launch_bounds__(64) void rc2_reduce1(const WcArgs<T> a) {
    using O = Ops<T, DP>;
    __shared__ T tl_all[2 * O::TLN];
    const int lane = threadIdx.x, ch = lane >> 5, i = lane & 31;
    T* tl = tl_all + ch * O::TLN;
    const int d = a.d;
    const long dd = (long)d * d;
    const long c_raw = (long)blockIdx.x * 2 + ch;
    const bool valid = c_raw < a.nchunk;
    const long c = valid ? c_raw : a.nchunk - 1;
    const long k0 = c * a.Lw, k1 = min(a.N, k0 + a.Lw);
    const bool row = i < d;
    const T hi_ = row ? a.H[row ? i : 0] : T(0);
    // identity element
    T A[DP], C[DP], J[DP], b = T(0), eta = T(0);
#pragma unroll
    for (int j = 0; j < DP; ++j) { A[j] = (row && j == i) ? T(1) : T(0); C[j] = T(0); J[j] = T(0); }
    for (int s = 0; s < a.Lw; ++s) {
        const long kk = k0 + s;
        const bool live = valid && kk < k1;
        const long k = kk < k1 ? kk : k1 - 1;
        const bool first = (k == 0 && a.seg_first != 0);
        T F[DP], Q[DP];
        O::template ld_row<FULL>(a.Fs + k * dd, d, i, F);
        O::template ld_row_sym<FULL>(a.Qs + k * dd, d, i, Q);
        const T y = a.ys[k];
        // predict: A' = F A, b' = F b, C' = F C F^T + Q
        T An[DP], Cn[DP], bn;
        {
            T lo[DP], hi[DP], FC[DP];
            O::split_mat(A, lo, hi);
            O::zero(An);
            O::mm0(An, F, lo, hi);
            O::split_mat(C, lo, hi);
            O::zero(FC);
            O::mm0(FC, F, lo, hi);
            bn = O::mv(F, b);
            O::split_mat(F, lo, hi);
            O::copy(Cn, Q);
            O::mm1(Cn, FC, lo, hi);
        }
        if (first) {        // the first element of the series: the prior itself, nothing to propagate
            T P0r[DP];
            O::template ld_row<FULL>(a.P0, d, i, P0r);
#pragma unroll
            for (int j = 0; j < DP; ++j) { An[j] = T(0); Cn[j] = P0r[j]; }
            bn = T(0);
        }
        O::symmetrise(Cn, tl, i);
        // scalar-innovation update (all zero multipliers when the observation is missing)
        // (h is a distributed vector like any other: element i in lane i)
        const T u = O::mv(Cn, hi_);                     // u = C' h
        T v;
        {
            T At[DP];
            O::transpose(An, At, tl, i);
            v = O::mv(At, hi_);                         // v = A'^T h
        }
        const T S = O::chain_sum(hi_ * u) + a.R;
        const T hb = O::chain_sum(hi_ * bn);
        const bool obs = !(y != y);
        const T inv = obs ? rcp_nr(S) : T(0);
        const T res = obs ? y - hb : T(0);
        O::rank1(An, v, -u * inv);
        O::rank1(Cn, u, -u * inv);
        O::rank1(J, v, live ? v * inv : T(0));          // (in place: a chain past its last step adds zeros)
        bn += u * inv * res;
        const T etan = eta + v * res * inv;
        // a chain past its last step keeps its total
        O::select(A, live, An);
        O::select(C, live, Cn);
        b = live ? bn : b;
        eta = live ? etan : eta;
    }
    T* out = a.agg1 + c * nfilt(d);
    O::template st_row<FULL>(out, d, i, valid, A);
    O::template st_row<FULL>(out + dd, d, i, valid, C);
    O::template st_row<FULL>(out + 2 * dd, d, i, valid, J);
    if (valid && row) {
        out[3 * dd + i] = b;
        out[3 * dd + d + i] = eta;
    }
}

// ====================================================================================================
// level 1: apply -- Kalman pass over the chunk, log-likelihood, smoothing total, the smoother gains (wc_apply1)
// ====================================================================================================
template <typename T, int DP>
struct SmthAcc {        // smoothing total of the chunk so far: rows of E and L, element of g
    T E[DP], L[DP], g;
};

// acc <- acc (x) e in time order (acc earlier): E = Ea Eb, g = Ea gb + ga, L = Ea Lb Ea^T + La; taken when `take`
template <typename T, int DP>
__device__ __forceinline__ void scombine(SmthAcc<T, DP>& s, const T* eE, const T* eL, T eg, bool take, T* tl, int i) {
    using O = Ops<T, DP>;
    T oE[DP], oL[DP], X[DP], lo[DP], hi[DP];
    O::split_mat(eE, lo, hi);
    O::zero(oE);
    O::mm0(oE, s.E, lo, hi);
    O::split_mat(eL, lo, hi);
    O::zero(X);
    O::mm0(X, s.E, lo, hi);
    const T og = O::mv(s.E, eg) + s.g;
    O::split_mat(s.E, lo, hi);
    O::copy(oL, s.L);
    O::mm1(oL, X, lo, hi);
    O::symmetrise(oL, tl, i);
    O::select(s.E, take, oE);
    O::select(s.L, take, oL);
    s.g = take ? og : s.g;
}

template <typename T, int DP, bool SMOOTH, bool FULL>
__global__ __launch_bounds__(64) void rc2_apply1(const WcArgs<T> a) {
    using O = Ops<T, DP>;
    __shared__ T tl_all[2 * O::TLN];
    __shared__ T rb_all[2 * 2 * DP];
    const int lane = threadIdx.x, ch = lane >> 5, i = lane & 31;
    T* tl = tl_all + ch * O::TLN;
    T* rb = rb_all + ch * 2 * DP;
    const int d = a.d;
    const long dd = (long)d * d;
    const long c_raw = (long)blockIdx.x * 2 + ch;
    const bool valid = c_raw < a.nchunk;
    const long c = valid ? c_raw : a.nchunk - 1;
    const long k0 = c * a.Lw, k1 = min(a.N, k0 + a.Lw);
    const bool row = i < d;
    const T hi_ = row ? a.H[row ? i : 0] : T(0);
    // state entering the chunk (wc_enter1)
    const T* en = a.enter1 + c * (d + dd);
    T m = row ? en[row ? i : 0] : T(0);
    T P[DP];
    O::template ld_row<FULL>(en + d, d, i, P);
    SmthAcc<T, DP> sacc;
    if (SMOOTH) {
#pragma unroll
        for (int j = 0; j < DP; ++j) { sacc.E[j] = (row && j == i) ? T(1) : T(0); sacc.L[j] = T(0); }
        sacc.g = T(0);
    }
    double quad = 0.0, mant = 1.0;
    long long expo = 0, count = 0;
    const bool ends_series = (k1 == a.N && a.seg_last != 0);
    const int trips = a.Lw + (SMOOTH ? 1 : 0);
    for (int s = 0; s < trips; ++s) {
        const long k = k0 + s;
        const bool regular = valid && k < k1;
        const bool halo = SMOOTH && valid && k == k1 && !ends_series;
        const bool first = (k == 0 && a.seg_first != 0);
        // F, Q of step k (the next segment's first step for the halo of a segment's last chunk)
        const long kc = k < a.N ? k : a.N - 1;
        const T* Fg = a.Fs + kc * dd;
        const T* Qg = a.Qs + kc * dd;
        if (SMOOTH && k == a.N && !a.seg_last) { Fg = a.halo_F; Qg = a.halo_Q; }
        T F[DP], Q[DP];
        O::template ld_row<FULL>(Fg, d, i, F);
        O::template ld_row_sym<FULL>(Qg, d, i, Q);
        const T y = a.ys[kc];
        // predict
        const T mp = O::mv(F, m);
        T FP[DP], Pp[DP];
        {
            T lo[DP], hi[DP];
            O::split_mat(P, lo, hi);
            O::zero(FP);
            O::mm0(FP, F, lo, hi);
            O::split_mat(F, lo, hi);
            O::copy(Pp, Q);
            O::mm1(Pp, FP, lo, hi);
        }
        O::symmetrise(Pp, tl, i);
        if (SMOOTH) {
            // element of step k-1: E = (Pp^-1 F P)^T, g = m - E mp, L = P - sym(E F P)
            const bool el = (regular || halo) && k > k0;
            T E[DP], L[DP], eg;
            {
                T M[DP], X[DP], diag = T(1);
                O::copy(M, Pp);
                O::copy(X, FP);
                O::solve(d, M, X, diag, rb, i);
                const T dinv = rcp_nr(diag);
#pragma unroll
                for (int j = 0; j < DP; ++j) X[j] *= dinv;
                O::transpose(X, E, tl, i);
            }
            O::template st_row<FULL>(a.Es + (k > 0 ? k - 1 : 0) * dd, d, i, el, E);
            eg = m - O::mv(E, mp);
            {
                T lo[DP], hi[DP], X[DP], Xt[DP];
                O::split_mat(FP, lo, hi);
                O::zero(X);
                O::mm0(X, E, lo, hi);
                O::transpose(X, Xt, tl, i);
#pragma unroll
                for (int j = 0; j < DP; ++j) L[j] = P[j] - T(0.5) * (X[j] + Xt[j]);
            }
            scombine<T, DP>(sacc, E, L, eg, el, tl, i);
        }
        // log-likelihood term from the predicted moments (also for the first step), measurement update
        const T u = O::mv(Pp, hi_);                     // (h: a distributed vector, element i in lane i)
        const T S = O::chain_sum(hi_ * u) + a.R;
        const T mu = O::chain_sum(hi_ * mp);
        const bool obs = !(y != y);
        if (obs && regular) {
            const double r = double(y) - double(mu);
            quad += r * r / double(S);
            int ex;
            mant = frexp(mant * double(S), &ex);
            expo += ex;
            count += 1;
        }
        T Pn[DP], mn;
        O::copy(Pn, Pp);
        T ub = u, Sb = S, mub = mu;
        mn = mp;
        if (blockIdx.x == 0 && s == 0 && a.seg_first) {
            // (wave-uniform) the first step of the series updates straight from the prior: chain 0 of this wave
            const T u0 = O::mv(P, hi_);
            const T S0 = O::chain_sum(hi_ * u0) + a.R;
            const T mu0 = O::chain_sum(hi_ * m);
            O::select(Pn, first, P);
            ub = first ? u0 : ub;
            Sb = first ? S0 : Sb;
            mub = first ? mu0 : mub;
            mn = first ? m : mn;
        }
        const T inv = obs ? rcp_nr(Sb) : T(0);
        O::rank1(Pn, ub, -ub * inv);
        mn += ub * (obs ? y - mub : T(0)) * inv;
        O::select(P, regular, Pn);
        m = regular ? mn : m;
        if (regular && row) a.fms[k * d + i] = m;
        O::template st_row<FULL>(a.fPs + kc * dd, d, i, regular, P);
    }
    if (SMOOTH) {
        // last element of the series: (0, m_N, P_N)
        T Z[DP];
        O::zero(Z);
        scombine<T, DP>(sacc, Z, P, m, valid && ends_series, tl, i);
        T* out = a.sagg1 + c * nsmth(d);
        O::template st_row<FULL>(out, d, i, valid, sacc.E);
        O::template st_row<FULL>(out + dd, d, i, valid, sacc.L);
        if (valid && row) out[2 * dd + i] = sacc.g;
    }
    if (valid && i == 0) {
        const double logdet = log(mant) + double(expo) * 0.6931471805599453;
        a.llpart[c] = -0.5 * (double(count) * 1.8378770664093453 + logdet + quad);
    }
}

// ====================================================================================================
// level 1 (smoother): RTS pass backwards over the chunk with the gains rc2_apply1 left in a.Es (wc_smooth1)
// ====================================================================================================
template <typename T, int DP, bool FULL>
__global__ __launch_bounds__(64) void rc2_smooth1(const WcArgs<T> a) {
    using O = Ops<T, DP>;
    __shared__ T tl_all[2 * O::TLN];
    const int lane = threadIdx.x, ch = lane >> 5, i = lane & 31;
    T* tl = tl_all + ch * O::TLN;
    const int d = a.d;
    const long dd = (long)d * d;
    const long c_raw = (long)blockIdx.x * 2 + ch;
    const bool valid = c_raw < a.nchunk;
    const long c = valid ? c_raw : a.nchunk - 1;
    const long k0 = c * a.Lw, k1 = min(a.N, k0 + a.Lw);
    const bool row = i < d;
    // smoothed state of the first step after the chunk (wc_senter1)
    const T* se = a.senter1 + c * (d + dd);
    T sm = row ? se[row ? i : 0] : T(0);
    T sP[DP];
    O::template ld_row<FULL>(se + d, d, i, sP);
    for (int s = 0; s < a.Lw; ++s) {
        const long kk = k1 - 1 - s;
        const bool live = valid && kk >= k0;
        const long k = kk >= k0 ? kk : k0;
        const bool terminal = (k == a.N - 1 && a.seg_last != 0);
        // (F, Q) of step k+1; for the terminal step of the series they are not used (any valid address)
        const T* Fg = a.Fs + (k + 1 < a.N ? k + 1 : k) * dd;
        const T* Qg = a.Qs + (k + 1 < a.N ? k + 1 : k) * dd;
        if (k + 1 == a.N && !a.seg_last) { Fg = a.halo_F; Qg = a.halo_Q; }
        T F[DP], X[DP], P[DP], E[DP];
        O::template ld_row<FULL>(Fg, d, i, F);
        O::template ld_row_sym<FULL>(Qg, d, i, X);                     // X: Q now, sP' - Pp below
        O::template ld_row<FULL>(a.fPs + k * dd, d, i, P);
        O::template ld_row<FULL>(a.Es + k * dd, d, i, E);
        const T m = row ? a.fms[k * d + (row ? i : 0)] : T(0);
        const T mp = O::mv(F, m);
        {
            T lo[DP], hi[DP], FP[DP];
            O::split_mat(P, lo, hi);
            O::zero(FP);
            O::mm0(FP, F, lo, hi);
            O::split_mat(F, lo, hi);
            O::mm1(X, FP, lo, hi);                      // Pp = F P F^T + Q
        }
#pragma unroll
        for (int j = 0; j < DP; ++j) X[j] = sP[j] - X[j];
        const T smn = m + O::mv(E, sm - mp);            // sm = m + E (sm' - mp)
        T sPn[DP];
        {
            T lo[DP], hi[DP], Y[DP];
            O::split_mat(X, lo, hi);
            O::zero(Y);
            O::mm0(Y, E, lo, hi);
            O::split_mat(E, lo, hi);
            O::copy(sPn, P);
            O::mm1(sPn, Y, lo, hi);                     // sP = P + E (sP' - Pp) E^T
        }
        O::symmetrise(sPn, tl, i);
        O::select(sPn, terminal, P);
        const T smt = terminal ? m : smn;
        O::select(sP, live, sPn);
        sm = live ? smt : sm;
        if (live && row) a.sms[k * d + i] = sm;
        O::template st_row<FULL>(a.sPs + k * dd, d, i, live, sP);
    }
}


// ====================================================================================================
// level 3: one Kogge-Stone level over the group totals, out[e] = in[e - stride] (x) in[e] (wc_ks_filter), two elements per
// wave.  filt_combine (pgps_math.h; the reference's filtering operator, pssgp/kalman/parallel.py:55-76) in row layout:
//   M = I + C1 J2;  M [G | Nm | w] = [A1 | C1 | b1 + C1 eta2]  (Gauss-Jordan with partial pivoting: the pivot of column c is
//   the largest entry among the rows that have not served -- a chain maximum and a ballot -- and no rows are exchanged: the
//   rows are put in order through LDS at the end);  A = A2 G,  b = A2 w + b2,  C = sym(A2 Nm A2^T + C2),
//   J = sym(G^T J2 A1 + J1),  eta = G^T (eta2 - J2 b1) + eta1.
// An element below the stride is copied: combined with the identity element, which is exact.  Operands are loaded when they
// are needed and results stored as soon as they are complete (in and out are different buffers): six matrices of an element
// pair do not fit the registers beside the elimination.
// ====================================================================================================
// maximum over the chain (every lane gets it): a butterfly of DPP moves inside the 16-lane rows -- lane ^ 1, lane ^ 2, then
// the mirrors of 8 and of 16 lanes, which act as ^ 4 and ^ 8 on what the earlier steps made uniform -- and the other row of
// the chain through the split.  (Five ds_bpermute round trips before: the pivot search of every elimination step.)
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double x) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, 0xf, 0xf, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xf, 0xf, true);
    return __builtin_bit_cast(double, (unsigned long long)lo | ((unsigned long long)hi << 32));
}
template <typename T>
__device__ __forceinline__ T chain_max(T v) {
    v = fmax(v, dpp_mov<0xB1>(v));      // quad_perm [1, 0, 3, 2]
    v = fmax(v, dpp_mov<0x4E>(v));      // quad_perm [2, 3, 0, 1]
    v = fmax(v, dpp_mov<0x141>(v));     // row_half_mirror
    v = fmax(v, dpp_mov<0x140>(v));     // row_mirror
    T lo, hi;
    split(v, lo, hi);
    return fmax(lo, hi);
}

template <typename T, int DP, bool FULL>
struct KsSolve {
    using O = Ops<T, DP>;
    static constexpr int NR = 2 * DP + 1;
    // M X = R in row layout with partial pivoting; on return row i of R holds the row of X whose index is `served`,
    // divided through already.  rb: DP + NR values of the chain's LDS.
    template <int C = 0>
    static __device__ __forceinline__ void run(int d, T* M, T* R, unsigned& used, T& diag, int& served, T* rb, int i, int ch) {
        if constexpr (C < DP) {
            if (C < d) {
                const bool cand = i < d && !((used >> i) & 1u);
                const T v = cand ? fabs(M[C]) : T(-1);
                const T mx = chain_max(v);
                const unsigned long long bal = __ballot(cand && v == mx);
                const unsigned half = ch ? (unsigned)(bal >> 32) : (unsigned)bal;
                const int p = half ? __builtin_ctz(half) : 0;
                used |= 1u << p;
                if (i == p) {
#pragma unroll
                    for (int j = C; j < DP; ++j) rb[j] = M[j];
#pragma unroll
                    for (int q = 0; q < NR; ++q) rb[DP + q] = R[q];
                    served = C;
                    diag = M[C];
                }
                wsync();
                const T inv = rcp_nr(rb[C]);
                const T fs = (i == p) ? T(0) : M[C] * inv;
#pragma unroll
                for (int j = C + 1; j < DP; ++j) M[j] = __builtin_fma(-fs, rb[j], M[j]);
#pragma unroll
                for (int q = 0; q < NR; ++q) R[q] = __builtin_fma(-fs, rb[DP + q], R[q]);
                wsync();
                run<C + 1>(d, M, R, used, diag, served, rb, i, ch);
            }
        }
    }
};

template <typename T, int DP, bool FULL>
__global__ __launch_bounds__(64) void rc2_ks_filter(const WcArgs<T> a) {
    using O = Ops<T, DP>;
    constexpr int NR = 2 * DP + 1;
    constexpr int PLD = NR + 1;                     // leading dimension of the row-ordering buffer
    __shared__ T tl_all[2 * O::TLN];
    __shared__ T rb_all[2 * (DP + NR)];
    __shared__ T pm_all[2 * DP * PLD];
    const int lane = threadIdx.x, ch = lane >> 5, i = lane & 31;
    T* tl = tl_all + ch * O::TLN;
    T* rb = rb_all + ch * (DP + NR);
    T* pm = pm_all + ch * DP * PLD;
    const int d = a.d;
    const long dd = (long)d * d;
    const int nf = nfilt(d);
    const long e_raw = (long)blockIdx.x * 2 + ch;
    const bool valid = e_raw < a.ks_n;
    const long e = valid ? e_raw : a.ks_n - 1;
    const bool comb = e >= a.ks_stride;             // below the stride: the element itself (combined with the identity)
    const T* g1 = a.ks_in + (comb ? e - a.ks_stride : e) * nf;
    const T* g2 = a.ks_in + e * nf;
    T* go = a.ks_out + e * nf;
    const bool row = i < d;
    const int ic = row ? i : 0;
    const T b1 = (comb && row) ? g1[3 * dd + ic] : T(0);
    const T eta1 = (comb && row) ? g1[3 * dd + d + ic] : T(0);
    const T b2 = row ? g2[3 * dd + ic] : T(0);
    const T eta2 = row ? g2[3 * dd + d + ic] : T(0);
    T G[DP], Nm[DP], w;
    {   // M = I + C1 J2, right-hand sides [A1 | C1 | b1 + C1 eta2], elimination, rows back in order
        T M[DP], R[NR];
        {
            T C1[DP], J2[DP], lo[DP], hi[DP];
            O::template ld_row<FULL>(g1 + dd, d, i, C1);
            O::template ld_row<FULL>(g2 + 2 * dd, d, i, J2);
#pragma unroll
            for (int j = 0; j < DP; ++j) C1[j] = comb ? C1[j] : T(0);
            O::split_mat(J2, lo, hi);
#pragma unroll
            for (int j = 0; j < DP; ++j) M[j] = (j == i) ? T(1) : T(0);
            O::mm0(M, C1, lo, hi);
            T A1[DP];
            O::template ld_row<FULL>(g1, d, i, A1);
#pragma unroll
            for (int j = 0; j < DP; ++j) {
                R[j] = comb ? A1[j] : ((j == i && row) ? T(1) : T(0));
                R[DP + j] = C1[j];
            }
            R[2 * DP] = b1 + O::mv(C1, eta2);
        }
        unsigned used = 0;
        T diag = T(1);
        int served = i;
        KsSolve<T, DP, FULL>::run(d, M, R, used, diag, served, rb, i, ch);
        const T dinv = rcp_nr(diag);
        if (i < DP) {
            const int r = (i < d) ? served : i;     // (rows beyond d never serve: they stay where they are)
#pragma unroll
            for (int q = 0; q < NR; ++q) pm[r * PLD + q] = R[q] * dinv;
        }
        wsync();
        const int ii = i < DP ? i : 0;
#pragma unroll
        for (int j = 0; j < DP; ++j) { G[j] = pm[ii * PLD + j]; Nm[j] = pm[ii * PLD + DP + j]; }
        w = pm[ii * PLD + 2 * DP];
        wsync();
    }
    {   // A = A2 G, b = A2 w + b2, C = sym(A2 Nm A2^T + C2)
        T A2[DP], lo[DP], hi[DP], Ao[DP], X[DP], Co[DP];
        O::template ld_row<FULL>(g2, d, i, A2);
        O::split_mat(G, lo, hi);
        O::zero(Ao);
        O::mm0(Ao, A2, lo, hi);
        O::template st_row<FULL>(go, d, i, valid, Ao);
        O::split_mat(Nm, lo, hi);
        O::zero(X);
        O::mm0(X, A2, lo, hi);
        const T bo = O::mv(A2, w) + b2;
        if (valid && row) go[3 * dd + i] = bo;
        O::template ld_row<FULL>(g2 + dd, d, i, Co);
        O::split_mat(A2, lo, hi);
        O::mm1(Co, X, lo, hi);
        O::symmetrise(Co, tl, i);
        O::template st_row<FULL>(go + dd, d, i, valid, Co);
    }
    {   // J = sym(G^T J2 A1 + J1), eta = G^T (eta2 - J2 b1) + eta1
        T J2[DP], A1[DP], lo[DP], hi[DP], M2[DP], Gt[DP], Jo[DP];
        O::template ld_row<FULL>(g2 + 2 * dd, d, i, J2);
        O::template ld_row<FULL>(g1, d, i, A1);
#pragma unroll
        for (int j = 0; j < DP; ++j) A1[j] = comb ? A1[j] : ((j == i && row) ? T(1) : T(0));
        const T z = eta2 - O::mv(J2, b1);
        O::split_mat(A1, lo, hi);
        O::zero(M2);
        O::mm0(M2, J2, lo, hi);
        O::transpose(G, Gt, tl, i);
        O::template ld_row<FULL>(g1 + 2 * dd, d, i, Jo);
#pragma unroll
        for (int j = 0; j < DP; ++j) Jo[j] = comb ? Jo[j] : T(0);
        O::split_mat(M2, lo, hi);
        O::mm0(Jo, Gt, lo, hi);
        O::symmetrise(Jo, tl, i);
        O::template st_row<FULL>(go + 2 * dd, d, i, valid, Jo);
        const T eo = O::mv(Gt, z) + eta1;
        if (valid && row) go[3 * dd + d + i] = eo;
    }
}

// one Kogge-Stone level of the smoother's suffix scan: out[e] = in[e] (x) in[e + stride] (wc_ks_smoother), two elements per
// wave; beyond the end: combined with the identity element (exact)
template <typename T, int DP, bool FULL>
__global__ __launch_bounds__(64) void rc2_ks_smoother(const WcArgs<T> a) {
    using O = Ops<T, DP>;
    __shared__ T tl_all[2 * O::TLN];
    const int lane = threadIdx.x, ch = lane >> 5, i = lane & 31;
    T* tl = tl_all + ch * O::TLN;
    const int d = a.d;
    const long dd = (long)d * d;
    const int ns = nsmth(d);
    const long e_raw = (long)blockIdx.x * 2 + ch;
    const bool valid = e_raw < a.ks_n;
    const long e = valid ? e_raw : a.ks_n - 1;
    const bool comb = e + a.ks_stride < a.ks_n;
    const T* ga = a.ks_in + e * ns;
    const T* gb = a.ks_in + (comb ? e + a.ks_stride : e) * ns;
    T* go = a.ks_out + e * ns;
    const bool row = i < d;
    const int ic = row ? i : 0;
    SmthAcc<T, DP> s;
    O::template ld_row<FULL>(ga, d, i, s.E);
    O::template ld_row<FULL>(ga + dd, d, i, s.L);
    s.g = row ? ga[2 * dd + ic] : T(0);
    T eE[DP], eL[DP];
    O::template ld_row<FULL>(gb, d, i, eE);
    O::template ld_row<FULL>(gb + dd, d, i, eL);
#pragma unroll
    for (int j = 0; j < DP; ++j) {
        eE[j] = comb ? eE[j] : ((j == i && row) ? T(1) : T(0));
        eL[j] = comb ? eL[j] : T(0);
    }
    const T eg = (comb && row) ? gb[2 * dd + ic] : T(0);
    scombine<T, DP>(s, eE, eL, eg, true, tl, i);
    O::template st_row<FULL>(go, d, i, valid, s.E);
    O::template st_row<FULL>(go + dd, d, i, valid, s.L);
    if (valid && row) go[2 * dd + i] = s.g;
}

}  // namespace rc2
}  // namespace pgps
