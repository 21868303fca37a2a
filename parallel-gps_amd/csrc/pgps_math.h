// pgps_math.h -- small-matrix algebra of the parallel Kalman filter / RTS smoother scan.
//
// Everything here is a register-resident, compile-time-D, fully unrolled device function:
// one lane owns whole d x d operands (d <= 6 or so).  No MFMA -- there is no dense
// contraction at these sizes; the kernels that use these functions are bound by HBM/L2
// traffic.  The header also compiles as plain C++ (g++) so the algebra can be
// checked on the CPU against the numpy oracle (tests/cpu_math/).
//
// Reference semantics (paths relative to /root/reference):
//   filtering elements   pssgp/kalman/parallel.py:13-72,83-97
//   filtering operator   pssgp/kalman/parallel.py:100-118
//   log-likelihood       pssgp/kalman/parallel.py:135-151
//   smoothing elements   pssgp/kalman/parallel.py:155-173
//   smoothing operator   pssgp/kalman/parallel.py:176-184
//
// Symmetric matrices (C, J, L, P) are stored as their upper triangle, row-major packed.
#pragma once

#if defined(__HIPCC__)
#define PGPS_HD __host__ __device__ __forceinline__
#else
#define PGPS_HD inline
#endif

#include <cmath>

namespace pgps {

template <int D>
struct Dim {
    static constexpr int MAT = D * D;
    static constexpr int SYM = D * (D + 1) / 2;
    static constexpr int NFILT = MAT + D + SYM + SYM + D;   // A, b, C, J, eta
    static constexpr int NSMTH = MAT + D + SYM;             // E, g, L
    static constexpr int NMP = D + SYM;                     // m, P
};

// packed index of (i, j) in an upper-triangular row-major store
template <int D>
PGPS_HD constexpr int symi(int i, int j) {
    return i <= j ? (i * D - (i * (i - 1)) / 2 + (j - i)) : (j * D - (j * (j - 1)) / 2 + (i - j));
}

template <typename T, int D>
struct FiltElem {           // x_out | x_in ~ N(A x_in + b, C);  p(y | x_in) ~ N_info(eta, J)
    T A[D * D];
    T b[D];
    T C[Dim<D>::SYM];
    T J[Dim<D>::SYM];
    T eta[D];
};

template <typename T, int D>
struct SmthElem {           // x_k | x_next ~ N(E x_next + g, L)
    T E[D * D];
    T g[D];
    T L[Dim<D>::SYM];
};

template <typename T, int D>
struct MeanCov {
    T m[D];
    T P[Dim<D>::SYM];
};

template <typename T>
PGPS_HD bool is_nan(T x) { return x != x; }

// 1 / x.  On the device, for plain float / double: the hardware reciprocal refined by Newton steps (v_rcp_f64 + two
// steps, v_rcp_f32 + one) instead of the IEEE division sequence (div_scale x 2, rcp, ~8 fma, div_fmas, div_fixup): the
// operands are innovation variances and 2 x 2 determinants, far from the range limits that sequence exists for, and the
// result agrees with the quotient to the last bit or the one before it.  Host builds (tests/cpu_math) and dual numbers
// divide.  -DPGPS_FAST_RCP=0 restores the divisions (A/B: profiles/r03_experiments.txt).
#ifndef PGPS_FAST_RCP
#define PGPS_FAST_RCP 1
#endif
template <typename T>
PGPS_HD T recip(T x) { return T(1) / x; }
#if defined(__HIP_DEVICE_COMPILE__) && PGPS_FAST_RCP
template <>
PGPS_HD double recip<double>(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    return r;
}
template <>
PGPS_HD float recip<float>(float x) {
    float r = __builtin_amdgcn_rcpf(x);
    r = __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
    return r;
}
#endif


// ------------------------------------------------------------------------------------
// Packed float32.  The float32 d >= 3 kernels keep whole operands in registers and run ONE wave per SIMD (512 registers);
// from one wave a v_fma_f32 issues every ~5.8 cycles and a v_pk_fma_f32 -- two multiply-adds -- every ~4.2
// (tools/micro/pk_f32.hip, profiles/r05_experiments.txt item 8): 2.8 x the multiply-adds per issue slot.  hipcc's SLP pass
// does not get there on the fully unrolled units (> 10 min per unit, Makefile), so the products are written on two-element
// vectors here, device float only; everything else (double, dual numbers, the host build of tests/cpu_math) takes the scalar
// loops.  Conventions: a pair is (column j, column j + 1) with j EVEN of a row -- rows of general matrices and of the packed
// triangles alike, so that a value belongs to one pair for its whole life and the register allocator can keep it there;
// products of a row with a column broadcast the row's entry (op_sel, free) against a pair of the other operand; products of
// two rows (X Y^T, matrix-vector) pair along the summation index and add the two halves at the end.  A product that is
// symmetric in exact arithmetic (F P F^T, A N A^T, G^T J A, E (F P)) is evaluated on its upper triangle only -- the scalar
// path averages it with its transpose, twice the multiply-adds for a rounding-level difference; the packed triangle it is
// stored in stays symmetric by construction either way.  -DPGPS_PK_F32=0 restores the scalar path (A/B).
// ------------------------------------------------------------------------------------
#ifndef PGPS_PK_F32
#define PGPS_PK_F32 14
#endif
#if defined(__HIP_DEVICE_COMPILE__) && PGPS_PK_F32
#define PGPS_PK_ON 1
#else
#define PGPS_PK_ON 0
#endif
// which callers take the packed path is a bit mask (PGPS_PK_F32): registers decide -- a pair lives in an aligned 64-bit
// register, and the d = 6 kernels already keep half of their values in the accumulation registers
enum PkSel { kPkTree = 0, kPkSolve = 1, kPkExtend = 2, kPkStep = 3, kPkOther = 4, kPkTreeS = 5 };
template <typename T, int D, int SEL>
struct UsePk { static constexpr bool on = false; };
#if PGPS_PK_ON
template <int D, int SEL>
struct UsePk<float, D, SEL> { static constexpr bool on = (D >= 2) && (((PGPS_PK_F32) >> SEL) & 1); };

namespace pk {
typedef float f2 __attribute__((ext_vector_type(2)));
PGPS_HD f2 mk(float a, float b) { return f2{a, b}; }
PGPS_HD f2 sp(float a) { return f2{a, a}; }
PGPS_HD f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
// pair (j, j + 1) of row i of a general / packed-symmetric matrix
template <int D> PGPS_HD f2 gen2(const float* X, int i, int j) { return mk(X[i * D + j], X[i * D + j + 1]); }
template <int D> PGPS_HD f2 sym2(const float* S, int i, int j) { return mk(S[symi<D>(i, j)], S[symi<D>(i, j + 1)]); }

// sum_k a[k] b[k] over contiguous runs (rows): pairs along k, halves added at the end
template <int N>
PGPS_HD float dot(const float* a, const float* b) {
    if constexpr (N == 1) {
        return a[0] * b[0];
    } else {
        f2 acc = mk(a[0], a[1]) * mk(b[0], b[1]);
#pragma unroll
        for (int k = 2; k + 1 < N; k += 2) acc = fma2(mk(a[k], a[k + 1]), mk(b[k], b[k + 1]), acc);
        float r = acc.x + acc.y;
        if constexpr (N & 1) r = __builtin_fmaf(a[N - 1], b[N - 1], r);
        return r;
    }
}

// out = X Y, both general: out(i, j..j+1) = sum_k X(i, k) * Y(k, j..j+1)
template <int D>
PGPS_HD void mat_mul(const float* X, const float* Y, float* out) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
#pragma unroll
        for (int j = 0; j + 1 < D; j += 2) {
            f2 acc = sp(X[i * D]) * gen2<D>(Y, 0, j);
#pragma unroll
            for (int k = 1; k < D; ++k) acc = fma2(sp(X[i * D + k]), gen2<D>(Y, k, j), acc);
            out[i * D + j] = acc.x;
            out[i * D + j + 1] = acc.y;
        }
        if constexpr (D & 1) {
            float acc = X[i * D] * Y[D - 1];
#pragma unroll
            for (int k = 1; k < D; ++k) acc = __builtin_fmaf(X[i * D + k], Y[k * D + D - 1], acc);
            out[i * D + D - 1] = acc;
        }
    }
}

// out = X S, S packed symmetric
template <int D>
PGPS_HD void mat_mul_sym(const float* X, const float* S, float* out) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
#pragma unroll
        for (int j = 0; j + 1 < D; j += 2) {
            f2 acc = sp(X[i * D]) * sym2<D>(S, 0, j);
#pragma unroll
            for (int k = 1; k < D; ++k) acc = fma2(sp(X[i * D + k]), sym2<D>(S, k, j), acc);
            out[i * D + j] = acc.x;
            out[i * D + j + 1] = acc.y;
        }
        if constexpr (D & 1) {
            float acc = X[i * D] * S[symi<D>(0, D - 1)];
#pragma unroll
            for (int k = 1; k < D; ++k) acc = __builtin_fmaf(X[i * D + k], S[symi<D>(k, D - 1)], acc);
            out[i * D + D - 1] = acc;
        }
    }
}

// out = S X, S packed symmetric
template <int D>
PGPS_HD void sym_mul_mat(const float* S, const float* X, float* out) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
#pragma unroll
        for (int j = 0; j + 1 < D; j += 2) {
            f2 acc = sp(S[symi<D>(i, 0)]) * gen2<D>(X, 0, j);
#pragma unroll
            for (int k = 1; k < D; ++k) acc = fma2(sp(S[symi<D>(i, k)]), gen2<D>(X, k, j), acc);
            out[i * D + j] = acc.x;
            out[i * D + j + 1] = acc.y;
        }
        if constexpr (D & 1) {
            float acc = S[symi<D>(i, 0)] * X[D - 1];
#pragma unroll
            for (int k = 1; k < D; ++k) acc = __builtin_fmaf(S[symi<D>(i, k)], X[k * D + D - 1], acc);
            out[i * D + D - 1] = acc;
        }
    }
}

// out(sym) = upper(X Y^T) + add(sym): rows against rows
template <int D>
PGPS_HD void mat_mul_t_sym(const float* X, const float* Y, const float* add, float* out) {
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = i; j < D; ++j) {
            const float acc = dot<D>(X + i * D, Y + j * D);
            out[symi<D>(i, j)] = add ? acc + add[symi<D>(i, j)] : acc;
        }
}

// out(sym) = upper(X^T Y) + add(sym): out(i, j..j+1) = sum_k X(k, i) * Y(k, j..j+1), the pairs that reach the triangle
template <int D>
PGPS_HD void mat_t_mul_sym(const float* X, const float* Y, const float* add, float* out) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
#pragma unroll
        for (int j = (i & ~1); j + 1 < D; j += 2) {
            f2 acc = sp(X[i]) * gen2<D>(Y, 0, j);
#pragma unroll
            for (int k = 1; k < D; ++k) acc = fma2(sp(X[k * D + i]), gen2<D>(Y, k, j), acc);
            if (j >= i) out[symi<D>(i, j)] = add ? acc.x + add[symi<D>(i, j)] : acc.x;
            out[symi<D>(i, j + 1)] = add ? acc.y + add[symi<D>(i, j + 1)] : acc.y;
        }
        if constexpr (D & 1) {
            float acc = X[i] * Y[D - 1];
#pragma unroll
            for (int k = 1; k < D; ++k) acc = __builtin_fmaf(X[k * D + i], Y[k * D + D - 1], acc);
            out[symi<D>(i, D - 1)] = add ? acc + add[symi<D>(i, D - 1)] : acc;
        }
    }
}

template <int D>
PGPS_HD void mat_vec(const float* X, const float* v, float* out) {
#pragma unroll
    for (int i = 0; i < D; ++i) out[i] = dot<D>(X + i * D, v);
}

// out = X^T v: out(j..j+1) = sum_k v(k) * X(k, j..j+1)
template <int D>
PGPS_HD void mat_t_vec(const float* X, const float* v, float* out) {
#pragma unroll
    for (int j = 0; j + 1 < D; j += 2) {
        f2 acc = sp(v[0]) * gen2<D>(X, 0, j);
#pragma unroll
        for (int k = 1; k < D; ++k) acc = fma2(sp(v[k]), gen2<D>(X, k, j), acc);
        out[j] = acc.x;
        out[j + 1] = acc.y;
    }
    if constexpr (D & 1) {
        float acc = v[0] * X[D - 1];
#pragma unroll
        for (int k = 1; k < D; ++k) acc = __builtin_fmaf(v[k], X[k * D + D - 1], acc);
        out[D - 1] = acc;
    }
}

template <int D>
PGPS_HD void sym_vec(const float* S, const float* v, float* out) {
#pragma unroll
    for (int j = 0; j + 1 < D; j += 2) {
        f2 acc = sp(v[0]) * sym2<D>(S, 0, j);
#pragma unroll
        for (int k = 1; k < D; ++k) acc = fma2(sp(v[k]), sym2<D>(S, k, j), acc);
        out[j] = acc.x;
        out[j + 1] = acc.y;
    }
    if constexpr (D & 1) {
        float acc = v[0] * S[symi<D>(0, D - 1)];
#pragma unroll
        for (int k = 1; k < D; ++k) acc = __builtin_fmaf(v[k], S[symi<D>(k, D - 1)], acc);
        out[D - 1] = acc;
    }
}

// dst(j) += f * src(j) for FROM <= j < N of two rows: the even-aligned pairs packed, the ragged ends scalar
template <int N, int FROM>
PGPS_HD void row_axpy(float* dst, const float* src, float f) {
    constexpr int A0 = (FROM + 1) & ~1;             // first even column >= FROM
    if constexpr (FROM < A0 && FROM < N) dst[FROM] = __builtin_fmaf(f, src[FROM], dst[FROM]);
#pragma unroll
    for (int j = A0; j + 1 < N; j += 2) {
        const f2 r = fma2(sp(f), mk(src[j], src[j + 1]), mk(dst[j], dst[j + 1]));
        dst[j] = r.x;
        dst[j + 1] = r.y;
    }
    if constexpr (A0 < N && ((N - A0) & 1)) dst[N - 1] = __builtin_fmaf(f, src[N - 1], dst[N - 1]);
}
template <int N, int FROM>
PGPS_HD void row_scale(float* dst, float f) {
    constexpr int A0 = (FROM + 1) & ~1;
    if constexpr (FROM < A0 && FROM < N) dst[FROM] *= f;
#pragma unroll
    for (int j = A0; j + 1 < N; j += 2) {
        const f2 r = sp(f) * mk(dst[j], dst[j + 1]);
        dst[j] = r.x;
        dst[j + 1] = r.y;
    }
    if constexpr (A0 < N && ((N - A0) & 1)) dst[N - 1] *= f;
}

// out(sym) = in(sym) + alpha u u^T: row i from its diagonal on, the even-aligned pairs packed
template <int D>
PGPS_HD void sym_rank1(const float* in, const float* u, float alpha, float* out) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
        const float t = alpha * u[i];
        if (i & 1) out[symi<D>(i, i)] = __builtin_fmaf(t, u[i], in[symi<D>(i, i)]);
#pragma unroll
        for (int j = (i + 1) & ~1; j + 1 < D; j += 2) {
            const f2 r = fma2(sp(t), mk(u[j], u[j + 1]), sym2<D>(in, i, j));
            out[symi<D>(i, j)] = r.x;
            out[symi<D>(i, j + 1)] = r.y;
        }
        if constexpr (D & 1) out[symi<D>(i, D - 1)] = __builtin_fmaf(t, u[D - 1], in[symi<D>(i, D - 1)]);
    }
}

// out = in + alpha a b^T (general D x D)
template <int D>
PGPS_HD void rank1(const float* in, const float* a, const float* b, float alpha, float* out) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
        const float t = alpha * a[i];
#pragma unroll
        for (int j = 0; j + 1 < D; j += 2) {
            const f2 r = fma2(sp(t), mk(b[j], b[j + 1]), gen2<D>(in, i, j));
            out[i * D + j] = r.x;
            out[i * D + j + 1] = r.y;
        }
        if constexpr (D & 1) out[i * D + D - 1] = __builtin_fmaf(t, b[D - 1], in[i * D + D - 1]);
    }
}
// out(sym) = add(sym) + upper(X Y), both general
template <int D>
PGPS_HD void mat_mul_upper(const float* X, const float* Y, const float* add, float* out) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
#pragma unroll
        for (int j = (i & ~1); j + 1 < D; j += 2) {
            f2 acc = mk(j >= i ? add[symi<D>(i, j)] : 0.0f, add[symi<D>(i, j + 1)]);
#pragma unroll
            for (int k = 0; k < D; ++k) acc = fma2(sp(X[i * D + k]), gen2<D>(Y, k, j), acc);
            if (j >= i) out[symi<D>(i, j)] = acc.x;
            out[symi<D>(i, j + 1)] = acc.y;
        }
        if constexpr (D & 1) {
            float acc = add[symi<D>(i, D - 1)];
#pragma unroll
            for (int k = 0; k < D; ++k) acc = __builtin_fmaf(X[i * D + k], Y[k * D + D - 1], acc);
            out[symi<D>(i, D - 1)] = acc;
        }
    }
}

// M = I + C J, both packed symmetric (the system of the filtering operator)
template <int D>
PGPS_HD void eye_plus_sym_sym(const float* C, const float* J, float* M) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
#pragma unroll
        for (int j = 0; j + 1 < D; j += 2) {
            f2 acc = mk(i == j ? 1.0f : 0.0f, i == j + 1 ? 1.0f : 0.0f);
#pragma unroll
            for (int k = 0; k < D; ++k) acc = fma2(sp(C[symi<D>(i, k)]), sym2<D>(J, k, j), acc);
            M[i * D + j] = acc.x;
            M[i * D + j + 1] = acc.y;
        }
        if constexpr (D & 1) {
            float acc = (i == D - 1) ? 1.0f : 0.0f;
#pragma unroll
            for (int k = 0; k < D; ++k) acc = __builtin_fmaf(C[symi<D>(i, k)], J[symi<D>(k, D - 1)], acc);
            M[i * D + D - 1] = acc;
        }
    }
}

// one column of the Gauss-Jordan elimination of gj_solve (below): pivot search by predicated row swaps as there, the row
// operations packed
template <int D, int NR, bool PIVOT, int C>
PGPS_HD void gj_col(float* M, float* B) {
    if constexpr (PIVOT) {
#pragma unroll
        for (int r = C + 1; r < D; ++r) {
            const bool sw = __builtin_fabsf(M[r * D + C]) > __builtin_fabsf(M[C * D + C]);
#pragma unroll
            for (int j = C; j < D; ++j) {
                const float u = M[C * D + j], v = M[r * D + j];
                M[C * D + j] = sw ? v : u;
                M[r * D + j] = sw ? u : v;
            }
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                const float u = B[C * NR + j], v = B[r * NR + j];
                B[C * NR + j] = sw ? v : u;
                B[r * NR + j] = sw ? u : v;
            }
        }
    }
    const float inv = recip(M[C * D + C]);
    row_scale<D, C + 1>(M + C * D, inv);
    row_scale<NR, 0>(B + C * NR, inv);
#pragma unroll
    for (int r = 0; r < D; ++r) {
        if (r == C) continue;
        const float f = -M[r * D + C];
        row_axpy<D, C + 1>(M + r * D, M + C * D, f);
        row_axpy<NR, 0>(B + r * NR, B + C * NR, f);
    }
    if constexpr (C + 1 < D) gj_col<D, NR, PIVOT, C + 1>(M, B);
}
}  // namespace pk
#endif      // PGPS_PK_ON

// ------------------------------------------------------------------------------------
// basic products
// ------------------------------------------------------------------------------------
// out = X * Y  (general D x D)
template <typename T, int D, int SEL = kPkOther>
PGPS_HD void mat_mul(const T* X, const T* Y, T* out) {
#if PGPS_PK_ON
    if constexpr (UsePk<T, D, SEL>::on) { pk::mat_mul<D>(X, Y, out); return; }
#endif
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) {
            T acc = T(0);
#pragma unroll
            for (int k = 0; k < D; ++k) acc += X[i * D + k] * Y[k * D + j];
            out[i * D + j] = acc;
        }
}

// out = X * S  with S symmetric-packed
template <typename T, int D, int SEL = kPkOther>
PGPS_HD void mat_mul_sym(const T* X, const T* S, T* out) {
#if PGPS_PK_ON
    if constexpr (UsePk<T, D, SEL>::on) { pk::mat_mul_sym<D>(X, S, out); return; }
#endif
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) {
            T acc = T(0);
#pragma unroll
            for (int k = 0; k < D; ++k) acc += X[i * D + k] * S[symi<D>(k, j)];
            out[i * D + j] = acc;
        }
}

// out = S * X  with S symmetric-packed
template <typename T, int D, int SEL = kPkOther>
PGPS_HD void sym_mul_mat(const T* S, const T* X, T* out) {
#if PGPS_PK_ON
    if constexpr (UsePk<T, D, SEL>::on) { pk::sym_mul_mat<D>(S, X, out); return; }
#endif
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) {
            T acc = T(0);
#pragma unroll
            for (int k = 0; k < D; ++k) acc += S[symi<D>(i, k)] * X[k * D + j];
            out[i * D + j] = acc;
        }
}

// out(sym) = sym_part(X * Y^T) + add(sym)      (add may be nullptr)
template <typename T, int D, int SEL = kPkOther>
PGPS_HD void mat_mul_t_sym(const T* X, const T* Y, const T* add, T* out) {
#if PGPS_PK_ON
    if constexpr (UsePk<T, D, SEL>::on) { pk::mat_mul_t_sym<D>(X, Y, add, out); return; }
#endif
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = i; j < D; ++j) {
            T acc = T(0);
            if (i == j) {
#pragma unroll
                for (int k = 0; k < D; ++k) acc += X[i * D + k] * Y[i * D + k];
            } else {
#pragma unroll
                for (int k = 0; k < D; ++k)
                    acc += X[i * D + k] * Y[j * D + k] + X[j * D + k] * Y[i * D + k];
                acc *= T(0.5);
            }
            out[symi<D>(i, j)] = add ? acc + add[symi<D>(i, j)] : acc;
        }
}

// out(sym) = sym_part(X^T * Y) + add(sym)
template <typename T, int D, int SEL = kPkOther>
PGPS_HD void mat_t_mul_sym(const T* X, const T* Y, const T* add, T* out) {
#if PGPS_PK_ON
    if constexpr (UsePk<T, D, SEL>::on) { pk::mat_t_mul_sym<D>(X, Y, add, out); return; }
#endif
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = i; j < D; ++j) {
            T acc = T(0);
            if (i == j) {
#pragma unroll
                for (int k = 0; k < D; ++k) acc += X[k * D + i] * Y[k * D + i];
            } else {
#pragma unroll
                for (int k = 0; k < D; ++k)
                    acc += X[k * D + i] * Y[k * D + j] + X[k * D + j] * Y[k * D + i];
                acc *= T(0.5);
            }
            out[symi<D>(i, j)] = add ? acc + add[symi<D>(i, j)] : acc;
        }
}

template <typename T, int D, int SEL = kPkOther>
PGPS_HD void mat_vec(const T* X, const T* v, T* out) {
#if PGPS_PK_ON
    if constexpr (UsePk<T, D, SEL>::on) { pk::mat_vec<D>(X, v, out); return; }
#endif
#pragma unroll
    for (int i = 0; i < D; ++i) {
        T acc = T(0);
#pragma unroll
        for (int k = 0; k < D; ++k) acc += X[i * D + k] * v[k];
        out[i] = acc;
    }
}

template <typename T, int D, int SEL = kPkOther>
PGPS_HD void mat_t_vec(const T* X, const T* v, T* out) {
#if PGPS_PK_ON
    if constexpr (UsePk<T, D, SEL>::on) { pk::mat_t_vec<D>(X, v, out); return; }
#endif
#pragma unroll
    for (int i = 0; i < D; ++i) {
        T acc = T(0);
#pragma unroll
        for (int k = 0; k < D; ++k) acc += X[k * D + i] * v[k];
        out[i] = acc;
    }
}

template <typename T, int D, int SEL = kPkOther>
PGPS_HD void sym_vec(const T* S, const T* v, T* out) {
#if PGPS_PK_ON
    if constexpr (UsePk<T, D, SEL>::on) { pk::sym_vec<D>(S, v, out); return; }
#endif
#pragma unroll
    for (int i = 0; i < D; ++i) {
        T acc = T(0);
#pragma unroll
        for (int k = 0; k < D; ++k) acc += S[symi<D>(i, k)] * v[k];
        out[i] = acc;
    }
}

// out(sym) = F * P(sym) * F^T + Q(sym);  FP (general) is also returned (= F P)
template <typename T, int D, int SEL = kPkOther>
PGPS_HD void predict_cov(const T* F, const T* P, const T* Q, T* FP, T* out) {
    mat_mul_sym<T, D, SEL>(F, P, FP);
    mat_mul_t_sym<T, D, SEL>(FP, F, Q, out);
}

// sum_i a[i] b[i] + init
template <typename T, int D, int SEL = kPkOther>
PGPS_HD T dot_add(const T* a, const T* b, T init) {
#if PGPS_PK_ON
    if constexpr (UsePk<T, D, SEL>::on) { return init + pk::dot<D>(a, b); }
#endif
    T acc = init;
#pragma unroll
    for (int i = 0; i < D; ++i) acc += a[i] * b[i];
    return acc;
}

// out(sym) = in(sym) -/+ u u^T inv   (the rank-one updates of the scalar-innovation steps; out may be in)
template <typename T, int D, bool MINUS, int SEL = kPkOther>
PGPS_HD void sym_outer(const T* in, const T* u, T inv, T* out) {
#if PGPS_PK_ON
    if constexpr (UsePk<T, D, SEL>::on) { pk::sym_rank1<D>(in, u, MINUS ? -inv : inv, out); return; }
#endif
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = i; j < D; ++j) {
            if (MINUS) out[symi<D>(i, j)] = in[symi<D>(i, j)] - u[i] * u[j] * inv;
            else out[symi<D>(i, j)] = in[symi<D>(i, j)] + u[i] * u[j] * inv;
        }
}

// ------------------------------------------------------------------------------------
// Solves.  gj_solve: M X = B for general M (partial pivoting by predicated row swaps, so
// every index stays compile-time and operands stay in registers).  M and B are destroyed;
// B holds X on return.  D = 1, 2 take closed forms.
// ------------------------------------------------------------------------------------
template <typename T, int D, int NR, bool PIVOT, int SEL = kPkOther>
PGPS_HD void gj_solve(T* M, T* B) {
    if constexpr (D == 1) {
        const T inv = recip(M[0]);
#pragma unroll
        for (int j = 0; j < NR; ++j) B[j] *= inv;
        return;
    } else if constexpr (D == 2) {
        const T a = M[0], b = M[1], c = M[2], d = M[3];
        const T inv = recip(a * d - b * c);
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const T x0 = B[j], x1 = B[NR + j];
            B[j] = (d * x0 - b * x1) * inv;
            B[NR + j] = (a * x1 - c * x0) * inv;
        }
        return;
    } else {
#if PGPS_PK_ON
    if constexpr (UsePk<T, D, SEL>::on) { pk::gj_col<D, NR, PIVOT, 0>(M, B); return; }
#endif
#pragma unroll
    for (int c = 0; c < D; ++c) {
        if (PIVOT) {
#pragma unroll
            for (int r = c + 1; r < D; ++r) {
                using std::fabs;
                const bool sw = fabs(M[r * D + c]) > fabs(M[c * D + c]);
#pragma unroll
                for (int j = c; j < D; ++j) {
                    const T u = M[c * D + j], v = M[r * D + j];
                    M[c * D + j] = sw ? v : u;
                    M[r * D + j] = sw ? u : v;
                }
#pragma unroll
                for (int j = 0; j < NR; ++j) {
                    const T u = B[c * NR + j], v = B[r * NR + j];
                    B[c * NR + j] = sw ? v : u;
                    B[r * NR + j] = sw ? u : v;
                }
            }
        }
        const T inv = recip(M[c * D + c]);
#pragma unroll
        for (int j = c + 1; j < D; ++j) M[c * D + j] *= inv;
#pragma unroll
        for (int j = 0; j < NR; ++j) B[c * NR + j] *= inv;
#pragma unroll
        for (int r = 0; r < D; ++r) {
            if (r == c) continue;
            const T f = M[r * D + c];
#pragma unroll
            for (int j = c + 1; j < D; ++j) M[r * D + j] -= f * M[c * D + j];
#pragma unroll
            for (int j = 0; j < NR; ++j) B[r * NR + j] -= f * B[c * NR + j];
        }
    }
    }
}

// ------------------------------------------------------------------------------------
// Filtering elements and operator
// ------------------------------------------------------------------------------------
template <typename T, int D>
PGPS_HD void filt_identity(FiltElem<T, D>& e) {
#pragma unroll
    for (int i = 0; i < D * D; ++i) e.A[i] = T(0);
#pragma unroll
    for (int i = 0; i < D; ++i) { e.A[i * D + i] = T(1); e.b[i] = T(0); e.eta[i] = T(0); }
#pragma unroll
    for (int i = 0; i < Dim<D>::SYM; ++i) { e.C[i] = T(0); e.J[i] = T(0); }
}

// First element of the series (parallel.py:13-43 with m0 = 0, parallel.py:125): the prior is
// updated WITHOUT a predict step.  J, eta of this element never reach an output (they only
// feed J, eta of prefixes, and a prefix is never a right operand), so they are left zero.
template <typename T, int D>
PGPS_HD void filt_first(FiltElem<T, D>& e, const T* P0 /*sym*/, T y, const T* h, T R) {
#pragma unroll
    for (int i = 0; i < D * D; ++i) e.A[i] = T(0);
#pragma unroll
    for (int i = 0; i < D; ++i) { e.b[i] = T(0); e.eta[i] = T(0); }
#pragma unroll
    for (int i = 0; i < Dim<D>::SYM; ++i) { e.C[i] = P0[i]; e.J[i] = T(0); }
    if (!is_nan(y)) {
        T u[D];
        sym_vec<T, D, kPkExtend>(P0, h, u);
        const T S = dot_add<T, D, kPkExtend>(h, u, R);
        const T inv = recip(S);
#pragma unroll
        for (int i = 0; i < D; ++i) e.b[i] = u[i] * (y * inv);
        sym_outer<T, D, true, kPkExtend>(e.C, u, inv, e.C);
    }
}

// e <- e (x) element(F, Q, y): extend an aggregate by one raw time step on its right.
// Written as "predict the conditional, then scalar-innovation update", which is the
// filtering operator (parallel.py:100-118) specialised to a raw right operand
// (parallel.py:46-72): J2 = (HF)^T (HF) / S is rank one, so (I + C1 J2)^-1 collapses to
// Sherman-Morrison and no d x d solve is needed.  Starting from the identity it reproduces
// the raw element itself.  NaN y = pure predict (parallel.py:46-53).
template <typename T, int D>
PGPS_HD void filt_extend(FiltElem<T, D>& e, const T* F, const T* Q /*sym*/, T y, const T* h, T R) {
    T Ap[D * D], bp[D], FC[D * D], Cp[Dim<D>::SYM];
    mat_mul<T, D, kPkExtend>(F, e.A, Ap);
    mat_vec<T, D, kPkExtend>(F, e.b, bp);
    predict_cov<T, D, kPkExtend>(F, e.C, Q, FC, Cp);
    if (is_nan(y)) {
#pragma unroll
        for (int i = 0; i < D * D; ++i) e.A[i] = Ap[i];
#pragma unroll
        for (int i = 0; i < D; ++i) e.b[i] = bp[i];
#pragma unroll
        for (int i = 0; i < Dim<D>::SYM; ++i) e.C[i] = Cp[i];
        return;
    }
    T u[D], v[D];
    sym_vec<T, D, kPkExtend>(Cp, h, u);            // Cp H^T
    mat_t_vec<T, D, kPkExtend>(Ap, h, v);          // (H Ap)^T
    const T S = dot_add<T, D, kPkExtend>(h, u, R), hb = dot_add<T, D, kPkExtend>(h, bp, T(0));
    const T inv = recip(S);
    const T res = y - hb;
#if PGPS_PK_ON
    if constexpr (UsePk<T, D, kPkExtend>::on) {
        pk::rank1<D>(Ap, u, v, -inv, e.A);
#pragma unroll
        for (int i = 0; i < D; ++i) {
            e.b[i] = bp[i] + (u[i] * inv) * res;
            e.eta[i] += v[i] * (res * inv);
        }
        sym_outer<T, D, true, kPkExtend>(Cp, u, inv, e.C);
        sym_outer<T, D, false, kPkExtend>(e.J, v, inv, e.J);
        return;
    }
#endif
#pragma unroll
    for (int i = 0; i < D; ++i) {
        const T Ki = u[i] * inv;
#pragma unroll
        for (int j = 0; j < D; ++j) e.A[i * D + j] = Ap[i * D + j] - Ki * v[j];
        e.b[i] = bp[i] + Ki * res;
        e.eta[i] += v[i] * (res * inv);
    }
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = i; j < D; ++j) {
            e.C[symi<D>(i, j)] = Cp[symi<D>(i, j)] - u[i] * u[j] * inv;
            e.J[symi<D>(i, j)] += v[i] * v[j] * inv;
        }
}

// the system of the filtering operator: M = I + C1 J2 and w = b1 + C1 eta2
template <typename T, int D, int SEL>
PGPS_HD void filt_system(const T* C1, const T* J2, const T* b1, const T* eta2, T* M, T* w) {
#if PGPS_PK_ON
    if constexpr (UsePk<T, D, SEL>::on) {
        pk::eye_plus_sym_sym<D>(C1, J2, M);
        T t[D];
        pk::sym_vec<D>(C1, eta2, t);
#pragma unroll
        for (int i = 0; i < D; ++i) w[i] = b1[i] + t[i];
        return;
    }
#endif
#pragma unroll
    for (int i = 0; i < D; ++i) {
        T wi = b1[i];
#pragma unroll
        for (int j = 0; j < D; ++j) {
            T acc = (i == j) ? T(1) : T(0);
#pragma unroll
            for (int k = 0; k < D; ++k) acc += C1[symi<D>(i, k)] * J2[symi<D>(k, j)];
            M[i * D + j] = acc;
            wi += C1[symi<D>(i, j)] * eta2[j];
        }
        w[i] = wi;
    }
}

// out = e1 (x) e2, the general filtering operator (parallel.py:100-118).
// One factorisation of M = I + C1 J2 serves both halves because (I + J2 C1) = M^T.
//   G = M^-1 A1, N = M^-1 C1, w = M^-1 (b1 + C1 eta2)
//   A = A2 G;  b = A2 w + b2;  C = sym(A2 N A2^T) + C2
//   eta = G^T (eta2 - J2 b1) + eta1;  J = sym(G^T J2 A1) + J1
template <typename T, int D>
PGPS_HD void filt_combine(const FiltElem<T, D>& e1, const FiltElem<T, D>& e2, FiltElem<T, D>& out) {
    constexpr int NR = 2 * D + 1;
    T M[D * D], B[D * NR], w[D];
    filt_system<T, D, kPkTree>(e1.C, e2.J, e1.b, e2.eta, M, w);
#pragma unroll
    for (int i = 0; i < D; ++i) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            B[i * NR + j] = e1.A[i * D + j];
            B[i * NR + D + j] = e1.C[symi<D>(i, j)];
        }
        B[i * NR + 2 * D] = w[i];
    }
    gj_solve<T, D, NR, true, kPkSolve>(M, B);
    T G[D * D], Nm[D * D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
#pragma unroll
        for (int j = 0; j < D; ++j) { G[i * D + j] = B[i * NR + j]; Nm[i * D + j] = B[i * NR + D + j]; }
        w[i] = B[i * NR + 2 * D];
    }
    T X[D * D];
    mat_mul<T, D, kPkTree>(e2.A, G, out.A);
    mat_vec<T, D, kPkTree>(e2.A, w, out.b);
#pragma unroll
    for (int i = 0; i < D; ++i) out.b[i] += e2.b[i];
    mat_mul<T, D, kPkTree>(e2.A, Nm, X);
    mat_mul_t_sym<T, D, kPkTree>(X, e2.A, e2.C, out.C);
    T z[D], Jb[D];
    sym_vec<T, D, kPkTree>(e2.J, e1.b, Jb);
#pragma unroll
    for (int i = 0; i < D; ++i) z[i] = e2.eta[i] - Jb[i];
    mat_t_vec<T, D, kPkTree>(G, z, out.eta);
#pragma unroll
    for (int i = 0; i < D; ++i) out.eta[i] += e1.eta[i];
    sym_mul_mat<T, D, kPkTree>(e2.J, e1.A, X);
    mat_t_mul_sym<T, D, kPkTree>(G, X, e1.J, out.J);
}

// (m, P) <- (0, m, P, ., .) (x) e2: push a filtered state through an aggregate.  This is
// filt_combine with A1 = 0 (every prefix that contains the first element has A = 0).
template <typename T, int D>
PGPS_HD void filt_apply(MeanCov<T, D>& s, const FiltElem<T, D>& e2) {
    constexpr int NR = D + 1;
    T M[D * D], B[D * NR], w[D];
    filt_system<T, D, kPkTree>(s.P, e2.J, s.m, e2.eta, M, w);
#pragma unroll
    for (int i = 0; i < D; ++i) {
#pragma unroll
        for (int j = 0; j < D; ++j) B[i * NR + j] = s.P[symi<D>(i, j)];
        B[i * NR + D] = w[i];
    }
    gj_solve<T, D, NR, true, kPkSolve>(M, B);
    T Nm[D * D], X[D * D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
#pragma unroll
        for (int j = 0; j < D; ++j) Nm[i * D + j] = B[i * NR + j];
        w[i] = B[i * NR + D];
    }
    mat_vec<T, D, kPkTree>(e2.A, w, s.m);
#pragma unroll
    for (int i = 0; i < D; ++i) s.m[i] += e2.b[i];
    mat_mul<T, D, kPkTree>(e2.A, Nm, X);
    mat_mul_t_sym<T, D, kPkTree>(X, e2.A, e2.C, s.P);
}

// ------------------------------------------------------------------------------------
// One Kalman step from a carried state, with the log-likelihood term of
// parallel.py:135-151.  `first` = first step of the whole series: the update uses the prior
// P0 directly (parallel.py:24-30) while the likelihood term uses F0 P0 F0^T + Q0
// (parallel.py:136-141).  Outputs mp, Pp (the predicted moments) for the smoother element.
// ------------------------------------------------------------------------------------
// residual / innovation variance widened to the accumulator's type (fp64 for float and double operands;
// dual numbers keep their derivatives: overloads in pgps_dual.h)
template <typename T> PGPS_HD double ll_diff(T y, T mu) { return double(y) - double(mu); }
template <typename T> PGPS_HD double ll_wide(T S) { return double(S); }

// Log-likelihood accumulator.  sum_k log s2_k is kept as a product of mantissas plus an exponent
// count (frexp is two cheap instructions on the GPU; an fp64 log is ~80): one log at the end.
struct LogLik {
    double quad = 0.0;      // sum (y - mu)^2 / s2
    double mant = 1.0;      // product of the mantissas of s2, renormalised to [0.5, 1) every step
    long long expo = 0;     // sum of the binary exponents of s2
    long long count = 0;    // number of observed steps

    PGPS_HD void add(double r, double S) {
        quad += r * r * recip(S);
        int e;
        mant = std::frexp(mant * S, &e);
        expo += e;
        count += 1;
    }
    PGPS_HD double value() const {
        const double logdet = std::log(mant) + double(expo) * 0.6931471805599453;
        return -0.5 * (double(count) * 1.8378770664093453 + logdet + quad);
    }
};

template <typename T, int D, typename LL>
PGPS_HD void kf_step(MeanCov<T, D>& s, const T* F, const T* Q /*sym*/, T y, const T* h, T R,
                     bool first, LL& ll, T* mp, T* Pp, T* FP) {
    mat_vec<T, D, kPkStep>(F, s.m, mp);
    predict_cov<T, D, kPkStep>(F, s.P, Q, FP, Pp);
    const bool obs = !is_nan(y);
    T u[D];
    sym_vec<T, D, kPkStep>(Pp, h, u);
    const T S = dot_add<T, D, kPkStep>(h, u, R), mu = dot_add<T, D, kPkStep>(h, mp, T(0));
    if (obs) {
        ll.add(ll_diff(y, mu), ll_wide(S));
    }
    if (first) {
        // update straight from the prior (s holds m0 = 0, P0)
        T u0[D];
        sym_vec<T, D, kPkStep>(s.P, h, u0);
        const T S0 = dot_add<T, D, kPkStep>(h, u0, R), mu0 = dot_add<T, D, kPkStep>(h, s.m, T(0));
        if (obs) {
            const T inv = recip(S0);
            const T res = y - mu0;
#pragma unroll
            for (int i = 0; i < D; ++i) s.m[i] += u0[i] * (res * inv);
            sym_outer<T, D, true, kPkStep>(s.P, u0, inv, s.P);
        }
        return;
    }
    if (obs) {
        const T inv = recip(S);
        const T res = y - mu;
#pragma unroll
        for (int i = 0; i < D; ++i) s.m[i] = mp[i] + u[i] * (res * inv);
        sym_outer<T, D, true, kPkStep>(Pp, u, inv, s.P);
    } else {
#pragma unroll
        for (int i = 0; i < D; ++i) s.m[i] = mp[i];
#pragma unroll
        for (int i = 0; i < Dim<D>::SYM; ++i) s.P[i] = Pp[i];
    }
}

// ------------------------------------------------------------------------------------
// Smoothing
// ------------------------------------------------------------------------------------
template <typename T, int D>
PGPS_HD void smth_identity(SmthElem<T, D>& e) {
#pragma unroll
    for (int i = 0; i < D * D; ++i) e.E[i] = T(0);
#pragma unroll
    for (int i = 0; i < D; ++i) { e.E[i * D + i] = T(1); e.g[i] = T(0); }
#pragma unroll
    for (int i = 0; i < Dim<D>::SYM; ++i) e.L[i] = T(0);
}

// Smoother gain E = P F^T Pp^-1 (parallel.py:160-162) from FP = F P and Pp (sym, SPD).
template <typename T, int D>
PGPS_HD void smth_gain(const T* FP, const T* Pp, T* E) {
    T M[D * D], B[D * D];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) { M[i * D + j] = Pp[symi<D>(i, j)]; B[i * D + j] = FP[i * D + j]; }
    gj_solve<T, D, D, false, kPkStep>(M, B);     // B = Pp^-1 F P = E^T
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) E[i * D + j] = B[j * D + i];
}

// Element (E, g, L) of step k from its filtered (m, P) and the NEXT step's predict
// (mp = F m, Pp = F P F^T + Q, FP = F P): g = m - E mp, L = sym(P - E (F P))
// (parallel.py:159-166; E Pp E^T = E F P because Pp E^T = F P).
template <typename T, int D>
PGPS_HD void smth_element(const MeanCov<T, D>& s, const T* mp, const T* Pp, const T* FP, SmthElem<T, D>& e) {
    smth_gain<T, D>(FP, Pp, e.E);
    T Em[D];
    mat_vec<T, D, kPkStep>(e.E, mp, Em);
#pragma unroll
    for (int i = 0; i < D; ++i) e.g[i] = s.m[i] - Em[i];
#if PGPS_PK_ON
    if constexpr (UsePk<T, D, kPkStep>::on) {
        T nE[D * D];
#pragma unroll
        for (int i = 0; i < D * D; ++i) nE[i] = -e.E[i];
        pk::mat_mul_upper<D>(nE, FP, s.P, e.L);
        return;
    }
#endif
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = i; j < D; ++j) {
            T acc = T(0);
#pragma unroll
            for (int k = 0; k < D; ++k)
                acc += e.E[i * D + k] * FP[k * D + j] + e.E[j * D + k] * FP[k * D + i];
            e.L[symi<D>(i, j)] = s.P[symi<D>(i, j)] - T(0.5) * acc;
        }
}

// Last element of the series: (0, m_N, P_N) (parallel.py:155-156)
template <typename T, int D>
PGPS_HD void smth_last(const MeanCov<T, D>& s, SmthElem<T, D>& e) {
#pragma unroll
    for (int i = 0; i < D * D; ++i) e.E[i] = T(0);
#pragma unroll
    for (int i = 0; i < D; ++i) e.g[i] = s.m[i];
#pragma unroll
    for (int i = 0; i < Dim<D>::SYM; ++i) e.L[i] = s.P[i];
}

// out = earlier (x) later in TIME order (the reference scans the reversed series, so its
// elem2 is `earlier` and elem1 is `later`, parallel.py:176-184):
//   E = Ea Eb;  g = Ea gb + ga;  L = Ea Lb Ea^T + La
template <typename T, int D>
PGPS_HD void smth_combine(const SmthElem<T, D>& a, const SmthElem<T, D>& b, SmthElem<T, D>& out) {
    T X[D * D], gv[D];
    mat_mul<T, D, kPkTreeS>(a.E, b.E, out.E);
    mat_vec<T, D, kPkTreeS>(a.E, b.g, gv);
#pragma unroll
    for (int i = 0; i < D; ++i) out.g[i] = gv[i] + a.g[i];
    mat_mul_sym<T, D, kPkTreeS>(a.E, b.L, X);
    mat_mul_t_sym<T, D, kPkTreeS>(X, a.E, a.L, out.L);
}

// (sm, sP) at the step after an aggregate -> (sm, sP) at the aggregate's first step
template <typename T, int D>
PGPS_HD void smth_apply(const SmthElem<T, D>& a, MeanCov<T, D>& s) {
    T X[D * D], gv[D], Ls[Dim<D>::SYM];
    mat_vec<T, D, kPkTreeS>(a.E, s.m, gv);
#pragma unroll
    for (int i = 0; i < D; ++i) s.m[i] = gv[i] + a.g[i];
    mat_mul_sym<T, D, kPkTreeS>(a.E, s.P, X);
    mat_mul_t_sym<T, D, kPkTreeS>(X, a.E, a.L, Ls);
#pragma unroll
    for (int i = 0; i < Dim<D>::SYM; ++i) s.P[i] = Ls[i];
}

// One RTS step (parallel.py:159-166 + 176-184 collapsed to Rauch-Tung-Striebel form):
//   sm_k = m_k + E (sm_{k+1} - mp);  sP_k = P_k + E (sP_{k+1} - Pp) E^T
// `s` holds (sm_{k+1}, sP_{k+1}) on entry and (sm_k, sP_k) on exit.
template <typename T, int D>
PGPS_HD void rts_step(const MeanCov<T, D>& f, const T* mp, const T* Pp, const T* FP, MeanCov<T, D>& s) {
    T E[D * D], dm[D], dP[Dim<D>::SYM], X[D * D], Em[D];
    smth_gain<T, D>(FP, Pp, E);
#pragma unroll
    for (int i = 0; i < D; ++i) dm[i] = s.m[i] - mp[i];
#pragma unroll
    for (int i = 0; i < Dim<D>::SYM; ++i) dP[i] = s.P[i] - Pp[i];
    mat_vec<T, D, kPkStep>(E, dm, Em);
#pragma unroll
    for (int i = 0; i < D; ++i) s.m[i] = f.m[i] + Em[i];
    mat_mul_sym<T, D, kPkStep>(E, dP, X);
    mat_mul_t_sym<T, D, kPkStep>(X, E, f.P, s.P);
}

// ------------------------------------------------------------------------------------
// The smoothing element in INNOVATION FORM.  With one scalar observation per step the filter's update is rank one:
// m_{k+1} - mp = u res / S,  P_{k+1} - Pp = -u u^T / S  (u = Pp H^T).  Write the smoothed moments relative to the filtered
// ones, d_k = sm_k - m_k,  D_k = sP_k - P_k; the RTS recursion (parallel.py:159-166 + 176-184) becomes
//     d_k = E_k d_{k+1} + (E_k u) res / S,      D_k = E_k D_{k+1} E_k^T - (E_k u)(E_k u)^T / S
// i.e. in these coordinates step k's element is (E_k, v res / S, -v v^T / S) with v = E_k u: its L is RANK ONE.  The
// composition law is the smoothing operator's own (it is the same affine map, read in shifted coordinates), so scans and
// trees do not change; what changes is the lane-serial fold: extending a total by a raw element costs one matrix product
// (E_a E), two matrix-vector products and a rank-one update instead of three matrix products, and the element needs no
// L = P - E (F P) product at all -- the smoother's counterpart of filt_extend's Sherman-Morrison step.  Per step at d = 6:
// 1080 -> ~530 multiply-adds on the smoothing side.  The series' last element maps everything to (0, 0): E = 0, g = 0, L = 0.
// Whoever applies a total in this form gets (d, D) and adds the filtered moments of that step (k_smoother_apply, `dform`).
// ------------------------------------------------------------------------------------
// kf_step that also hands out the update it made: u = Pp H^T, inv = 1 / S (0 for a missing observation), res = y - H mp
template <typename T, int D, typename LL>
PGPS_HD void kf_step_u(MeanCov<T, D>& s, const T* F, const T* Q /*sym*/, T y, const T* h, T R, bool first, LL& ll, T* mp,
                       T* Pp, T* FP, T* u, T& inv_o, T& res_o) {
    mat_vec<T, D, kPkStep>(F, s.m, mp);
    predict_cov<T, D, kPkStep>(F, s.P, Q, FP, Pp);
    const bool obs = !is_nan(y);
    sym_vec<T, D, kPkStep>(Pp, h, u);
    const T S = dot_add<T, D, kPkStep>(h, u, R), mu = dot_add<T, D, kPkStep>(h, mp, T(0));
    if (obs) ll.add(ll_diff(y, mu), ll_wide(S));
    inv_o = T(0);
    res_o = T(0);
    if (first) {
        // update straight from the prior (s holds m0 = 0, P0); no element is built from this step's predict
        T u0[D];
        sym_vec<T, D, kPkStep>(s.P, h, u0);
        const T S0 = dot_add<T, D, kPkStep>(h, u0, R), mu0 = dot_add<T, D, kPkStep>(h, s.m, T(0));
        if (obs) {
            const T inv = recip(S0);
            const T res = y - mu0;
#pragma unroll
            for (int i = 0; i < D; ++i) s.m[i] += u0[i] * (res * inv);
            sym_outer<T, D, true, kPkStep>(s.P, u0, inv, s.P);
        }
        return;
    }
    if (obs) {
        const T inv = recip(S);
        const T res = y - mu;
        inv_o = inv;
        res_o = res;
#pragma unroll
        for (int i = 0; i < D; ++i) s.m[i] = mp[i] + u[i] * (res * inv);
        sym_outer<T, D, true, kPkStep>(Pp, u, inv, s.P);
    } else {
#pragma unroll
        for (int i = 0; i < D; ++i) s.m[i] = mp[i];
#pragma unroll
        for (int i = 0; i < Dim<D>::SYM; ++i) s.P[i] = Pp[i];
    }
}

// a <- a (x) element, the element of a step in innovation form: gain E, and the NEXT step's update (u, inv = 1/S or 0, res)
template <typename T, int D>
PGPS_HD void smth_extend_u(SmthElem<T, D>& a, const T* E, const T* u, T inv, T res) {
    T v[D], w[D], EE[D * D];
    mat_vec<T, D, kPkStep>(E, u, v);
    mat_vec<T, D, kPkStep>(a.E, v, w);               // (with the total's gain BEFORE this step)
    const T c = res * inv;
#pragma unroll
    for (int i = 0; i < D; ++i) a.g[i] += w[i] * c;
    sym_outer<T, D, true, kPkStep>(a.L, w, inv, a.L);
    mat_mul<T, D, kPkStep>(a.E, E, EE);
#pragma unroll
    for (int i = 0; i < D * D; ++i) a.E[i] = EE[i];
}
// ... and the series' last element: everything after it is forgotten, nothing is added
template <typename T, int D>
PGPS_HD void smth_extend_last_u(SmthElem<T, D>& a) {
#pragma unroll
    for (int i = 0; i < D * D; ++i) a.E[i] = T(0);
}

}  // namespace pgps
