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
// basic products
// ------------------------------------------------------------------------------------
// out = X * Y  (general D x D)
template <typename T, int D>
PGPS_HD void mat_mul(const T* X, const T* Y, T* out) {
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
template <typename T, int D>
PGPS_HD void mat_mul_sym(const T* X, const T* S, T* out) {
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
template <typename T, int D>
PGPS_HD void sym_mul_mat(const T* S, const T* X, T* out) {
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
template <typename T, int D>
PGPS_HD void mat_mul_t_sym(const T* X, const T* Y, const T* add, T* out) {
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
template <typename T, int D>
PGPS_HD void mat_t_mul_sym(const T* X, const T* Y, const T* add, T* out) {
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

template <typename T, int D>
PGPS_HD void mat_vec(const T* X, const T* v, T* out) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
        T acc = T(0);
#pragma unroll
        for (int k = 0; k < D; ++k) acc += X[i * D + k] * v[k];
        out[i] = acc;
    }
}

template <typename T, int D>
PGPS_HD void mat_t_vec(const T* X, const T* v, T* out) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
        T acc = T(0);
#pragma unroll
        for (int k = 0; k < D; ++k) acc += X[k * D + i] * v[k];
        out[i] = acc;
    }
}

template <typename T, int D>
PGPS_HD void sym_vec(const T* S, const T* v, T* out) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
        T acc = T(0);
#pragma unroll
        for (int k = 0; k < D; ++k) acc += S[symi<D>(i, k)] * v[k];
        out[i] = acc;
    }
}

// out(sym) = F * P(sym) * F^T + Q(sym);  FP (general) is also returned (= F P)
template <typename T, int D>
PGPS_HD void predict_cov(const T* F, const T* P, const T* Q, T* FP, T* out) {
    mat_mul_sym<T, D>(F, P, FP);
    mat_mul_t_sym<T, D>(FP, F, Q, out);
}

// ------------------------------------------------------------------------------------
// Solves.  gj_solve: M X = B for general M (partial pivoting by predicated row swaps, so
// every index stays compile-time and operands stay in registers).  M and B are destroyed;
// B holds X on return.  D = 1, 2 take closed forms.
// ------------------------------------------------------------------------------------
template <typename T, int D, int NR, bool PIVOT>
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
        sym_vec<T, D>(P0, h, u);
        T S = R;
#pragma unroll
        for (int i = 0; i < D; ++i) S += h[i] * u[i];
        const T inv = recip(S);
#pragma unroll
        for (int i = 0; i < D; ++i) e.b[i] = u[i] * (y * inv);
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = i; j < D; ++j) e.C[symi<D>(i, j)] -= u[i] * u[j] * inv;
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
    mat_mul<T, D>(F, e.A, Ap);
    mat_vec<T, D>(F, e.b, bp);
    predict_cov<T, D>(F, e.C, Q, FC, Cp);
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
    sym_vec<T, D>(Cp, h, u);            // Cp H^T
    mat_t_vec<T, D>(Ap, h, v);          // (H Ap)^T
    T S = R, hb = T(0);
#pragma unroll
    for (int i = 0; i < D; ++i) { S += h[i] * u[i]; hb += h[i] * bp[i]; }
    const T inv = recip(S);
    const T res = y - hb;
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

// out = e1 (x) e2, the general filtering operator (parallel.py:100-118).
// One factorisation of M = I + C1 J2 serves both halves because (I + J2 C1) = M^T.
//   G = M^-1 A1, N = M^-1 C1, w = M^-1 (b1 + C1 eta2)
//   A = A2 G;  b = A2 w + b2;  C = sym(A2 N A2^T) + C2
//   eta = G^T (eta2 - J2 b1) + eta1;  J = sym(G^T J2 A1) + J1
template <typename T, int D>
PGPS_HD void filt_combine(const FiltElem<T, D>& e1, const FiltElem<T, D>& e2, FiltElem<T, D>& out) {
    constexpr int NR = 2 * D + 1;
    T M[D * D], B[D * NR];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        T wi = e1.b[i];
#pragma unroll
        for (int j = 0; j < D; ++j) {
            T acc = (i == j) ? T(1) : T(0);
#pragma unroll
            for (int k = 0; k < D; ++k) acc += e1.C[symi<D>(i, k)] * e2.J[symi<D>(k, j)];
            M[i * D + j] = acc;
            B[i * NR + j] = e1.A[i * D + j];
            B[i * NR + D + j] = e1.C[symi<D>(i, j)];
            wi += e1.C[symi<D>(i, j)] * e2.eta[j];
        }
        B[i * NR + 2 * D] = wi;
    }
    gj_solve<T, D, NR, true>(M, B);
    T G[D * D], Nm[D * D], w[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
#pragma unroll
        for (int j = 0; j < D; ++j) { G[i * D + j] = B[i * NR + j]; Nm[i * D + j] = B[i * NR + D + j]; }
        w[i] = B[i * NR + 2 * D];
    }
    T X[D * D];
    mat_mul<T, D>(e2.A, G, out.A);
    mat_vec<T, D>(e2.A, w, out.b);
#pragma unroll
    for (int i = 0; i < D; ++i) out.b[i] += e2.b[i];
    mat_mul<T, D>(e2.A, Nm, X);
    mat_mul_t_sym<T, D>(X, e2.A, e2.C, out.C);
    T z[D], Jb[D];
    sym_vec<T, D>(e2.J, e1.b, Jb);
#pragma unroll
    for (int i = 0; i < D; ++i) z[i] = e2.eta[i] - Jb[i];
    mat_t_vec<T, D>(G, z, out.eta);
#pragma unroll
    for (int i = 0; i < D; ++i) out.eta[i] += e1.eta[i];
    sym_mul_mat<T, D>(e2.J, e1.A, X);
    mat_t_mul_sym<T, D>(G, X, e1.J, out.J);
}

// (m, P) <- (0, m, P, ., .) (x) e2: push a filtered state through an aggregate.  This is
// filt_combine with A1 = 0 (every prefix that contains the first element has A = 0).
template <typename T, int D>
PGPS_HD void filt_apply(MeanCov<T, D>& s, const FiltElem<T, D>& e2) {
    constexpr int NR = D + 1;
    T M[D * D], B[D * NR];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        T wi = s.m[i];
#pragma unroll
        for (int j = 0; j < D; ++j) {
            T acc = (i == j) ? T(1) : T(0);
#pragma unroll
            for (int k = 0; k < D; ++k) acc += s.P[symi<D>(i, k)] * e2.J[symi<D>(k, j)];
            M[i * D + j] = acc;
            B[i * NR + j] = s.P[symi<D>(i, j)];
            wi += s.P[symi<D>(i, j)] * e2.eta[j];
        }
        B[i * NR + D] = wi;
    }
    gj_solve<T, D, NR, true>(M, B);
    T Nm[D * D], w[D], X[D * D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
#pragma unroll
        for (int j = 0; j < D; ++j) Nm[i * D + j] = B[i * NR + j];
        w[i] = B[i * NR + D];
    }
    mat_vec<T, D>(e2.A, w, s.m);
#pragma unroll
    for (int i = 0; i < D; ++i) s.m[i] += e2.b[i];
    mat_mul<T, D>(e2.A, Nm, X);
    mat_mul_t_sym<T, D>(X, e2.A, e2.C, s.P);
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
    mat_vec<T, D>(F, s.m, mp);
    predict_cov<T, D>(F, s.P, Q, FP, Pp);
    const bool obs = !is_nan(y);
    T u[D];
    sym_vec<T, D>(Pp, h, u);
    T S = R, mu = T(0);
#pragma unroll
    for (int i = 0; i < D; ++i) { S += h[i] * u[i]; mu += h[i] * mp[i]; }
    if (obs) {
        ll.add(ll_diff(y, mu), ll_wide(S));
    }
    if (first) {
        // update straight from the prior (s holds m0 = 0, P0)
        T u0[D];
        sym_vec<T, D>(s.P, h, u0);
        T S0 = R, mu0 = T(0);
#pragma unroll
        for (int i = 0; i < D; ++i) { S0 += h[i] * u0[i]; mu0 += h[i] * s.m[i]; }
        if (obs) {
            const T inv = recip(S0);
            const T res = y - mu0;
#pragma unroll
            for (int i = 0; i < D; ++i) s.m[i] += u0[i] * (res * inv);
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = i; j < D; ++j) s.P[symi<D>(i, j)] -= u0[i] * u0[j] * inv;
        }
        return;
    }
    if (obs) {
        const T inv = recip(S);
        const T res = y - mu;
#pragma unroll
        for (int i = 0; i < D; ++i) s.m[i] = mp[i] + u[i] * (res * inv);
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = i; j < D; ++j) s.P[symi<D>(i, j)] = Pp[symi<D>(i, j)] - u[i] * u[j] * inv;
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
    gj_solve<T, D, D, false>(M, B);     // B = Pp^-1 F P = E^T
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
    mat_vec<T, D>(e.E, mp, Em);
#pragma unroll
    for (int i = 0; i < D; ++i) e.g[i] = s.m[i] - Em[i];
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
    mat_mul<T, D>(a.E, b.E, out.E);
    mat_vec<T, D>(a.E, b.g, gv);
#pragma unroll
    for (int i = 0; i < D; ++i) out.g[i] = gv[i] + a.g[i];
    mat_mul_sym<T, D>(a.E, b.L, X);
    mat_mul_t_sym<T, D>(X, a.E, a.L, out.L);
}

// (sm, sP) at the step after an aggregate -> (sm, sP) at the aggregate's first step
template <typename T, int D>
PGPS_HD void smth_apply(const SmthElem<T, D>& a, MeanCov<T, D>& s) {
    T X[D * D], gv[D], Ls[Dim<D>::SYM];
    mat_vec<T, D>(a.E, s.m, gv);
#pragma unroll
    for (int i = 0; i < D; ++i) s.m[i] = gv[i] + a.g[i];
    mat_mul_sym<T, D>(a.E, s.P, X);
    mat_mul_t_sym<T, D>(X, a.E, a.L, Ls);
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
    mat_vec<T, D>(E, dm, Em);
#pragma unroll
    for (int i = 0; i < D; ++i) s.m[i] = f.m[i] + Em[i];
    mat_mul_sym<T, D>(E, dP, X);
    mat_mul_t_sym<T, D>(X, E, f.P, s.P);
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
    mat_vec<T, D>(F, s.m, mp);
    predict_cov<T, D>(F, s.P, Q, FP, Pp);
    const bool obs = !is_nan(y);
    sym_vec<T, D>(Pp, h, u);
    T S = R, mu = T(0);
#pragma unroll
    for (int i = 0; i < D; ++i) { S += h[i] * u[i]; mu += h[i] * mp[i]; }
    if (obs) ll.add(ll_diff(y, mu), ll_wide(S));
    inv_o = T(0);
    res_o = T(0);
    if (first) {
        // update straight from the prior (s holds m0 = 0, P0); no element is built from this step's predict
        T u0[D];
        sym_vec<T, D>(s.P, h, u0);
        T S0 = R, mu0 = T(0);
#pragma unroll
        for (int i = 0; i < D; ++i) { S0 += h[i] * u0[i]; mu0 += h[i] * s.m[i]; }
        if (obs) {
            const T inv = recip(S0);
            const T res = y - mu0;
#pragma unroll
            for (int i = 0; i < D; ++i) s.m[i] += u0[i] * (res * inv);
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = i; j < D; ++j) s.P[symi<D>(i, j)] -= u0[i] * u0[j] * inv;
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
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = i; j < D; ++j) s.P[symi<D>(i, j)] = Pp[symi<D>(i, j)] - u[i] * u[j] * inv;
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
    mat_vec<T, D>(E, u, v);
    mat_vec<T, D>(a.E, v, w);               // (with the total's gain BEFORE this step)
    const T c = res * inv;
#pragma unroll
    for (int i = 0; i < D; ++i) a.g[i] += w[i] * c;
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = i; j < D; ++j) a.L[symi<D>(i, j)] -= w[i] * w[j] * inv;
    mat_mul<T, D>(a.E, E, EE);
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
