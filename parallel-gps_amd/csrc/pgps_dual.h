// pgps_dual.h -- forward-mode dual numbers: a value and NP partial derivatives.
//
// Instantiating the scan algebra of pgps_math.h on Dual<NP> differentiates the whole parallel filter
// -- elements, the associative operator, the log-likelihood -- with respect to NP hyper-parameters in
// one pass: differentiation commutes with the scan (the operator on dual elements is still
// associative), so no adjoint scan is needed for the gradient of the log-likelihood that the
// reference obtains from TensorFlow autodiff (tests/test_gp_vs_kfs.py:53-78).
#pragma once

#include <cmath>

#include "pgps_math.h"

namespace pgps {

template <int NP>
struct Dual {
    double v;
    double d[NP];

    PGPS_HD Dual() {}
    PGPS_HD Dual(double x) : v(x) {
#pragma unroll
        for (int i = 0; i < NP; ++i) d[i] = 0.0;
    }
    PGPS_HD explicit operator double() const { return v; }

    PGPS_HD Dual operator-() const {
        Dual r;
        r.v = -v;
#pragma unroll
        for (int i = 0; i < NP; ++i) r.d[i] = -d[i];
        return r;
    }
    PGPS_HD Dual& operator+=(const Dual& o) {
        v += o.v;
#pragma unroll
        for (int i = 0; i < NP; ++i) d[i] += o.d[i];
        return *this;
    }
    PGPS_HD Dual& operator-=(const Dual& o) {
        v -= o.v;
#pragma unroll
        for (int i = 0; i < NP; ++i) d[i] -= o.d[i];
        return *this;
    }
    PGPS_HD Dual& operator*=(const Dual& o) {
#pragma unroll
        for (int i = 0; i < NP; ++i) d[i] = d[i] * o.v + v * o.d[i];
        v *= o.v;
        return *this;
    }
};

template <int NP>
PGPS_HD Dual<NP> operator+(Dual<NP> a, const Dual<NP>& b) { a += b; return a; }
template <int NP>
PGPS_HD Dual<NP> operator-(Dual<NP> a, const Dual<NP>& b) { a -= b; return a; }
template <int NP>
PGPS_HD Dual<NP> operator*(Dual<NP> a, const Dual<NP>& b) { a *= b; return a; }
template <int NP>
PGPS_HD Dual<NP> operator/(const Dual<NP>& a, const Dual<NP>& b) {
    Dual<NP> r;
    const double inv = 1.0 / b.v;
    r.v = a.v * inv;
#pragma unroll
    for (int i = 0; i < NP; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv;
    return r;
}
template <int NP> PGPS_HD bool operator!=(const Dual<NP>& a, const Dual<NP>& b) { return a.v != b.v; }
template <int NP> PGPS_HD bool operator==(const Dual<NP>& a, const Dual<NP>& b) { return a.v == b.v; }
template <int NP> PGPS_HD bool operator>(const Dual<NP>& a, const Dual<NP>& b) { return a.v > b.v; }
template <int NP> PGPS_HD bool operator<(const Dual<NP>& a, const Dual<NP>& b) { return a.v < b.v; }

template <int NP>
PGPS_HD Dual<NP> fabs(const Dual<NP>& a) { return a.v < 0 ? -a : a; }
template <int NP>
PGPS_HD Dual<NP> exp(const Dual<NP>& a) {
    Dual<NP> r;
    r.v = ::exp(a.v);
#pragma unroll
    for (int i = 0; i < NP; ++i) r.d[i] = r.v * a.d[i];
    return r;
}

template <int NP> PGPS_HD Dual<NP> ll_diff(const Dual<NP>& y, const Dual<NP>& mu) { return y - mu; }
template <int NP> PGPS_HD Dual<NP> ll_wide(const Dual<NP>& S) { return S; }

// log-likelihood accumulator on duals: value as in LogLik (mantissa product), derivative of
// sum log s2 as sum ds2 / s2
template <int NP>
struct LogLikDual {
    Dual<NP> quad = Dual<NP>(0.0);
    double mant = 1.0;
    long long expo = 0;
    long long count = 0;
    double dlog[NP] = {};

    PGPS_HD void add(const Dual<NP>& r, const Dual<NP>& S) {
        quad += r * r / S;
        int e;
        mant = std::frexp(mant * S.v, &e);
        expo += e;
        count += 1;
        const double inv = 1.0 / S.v;
#pragma unroll
        for (int i = 0; i < NP; ++i) dlog[i] += S.d[i] * inv;
    }
    PGPS_HD Dual<NP> value() const {
        Dual<NP> r;
        r.v = -0.5 * (double(count) * 1.8378770664093453 + std::log(mant) + double(expo) * 0.6931471805599453 + quad.v);
#pragma unroll
        for (int i = 0; i < NP; ++i) r.d[i] = -0.5 * (dlog[i] + quad.d[i]);
        return r;
    }
};

// F(dt) = exp(-lam dt) (I + dt N + dt^2 N^2/2)  and  Q(dt) = Pinf - F Pinf F^T  on duals (d <= 3), the
// closed-form discretisation of pgps_fused.hip.h with derivatives carried along.
template <int NP, int D>
PGPS_HD void lti_step_dual(const Dual<NP>& lam, const Dual<NP>* N1, const Dual<NP>* N2, const Dual<NP>* Pinf, double dt,
                           Dual<NP>* F, Dual<NP>* Q /*sym*/) {
    using T = Dual<NP>;
    constexpr int MAT = D * D;
    const T e = exp(-(lam * T(dt)));
    T X[MAT];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) {
            T v = T(i == j ? 1.0 : 0.0);
            if (D >= 2) v += T(dt) * N1[i * D + j];
            if (D >= 3) v += T(dt * dt) * N2[i * D + j];
            F[i * D + j] = e * v;
        }
    mat_mul<T, D>(F, Pinf, X);
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = i; j < D; ++j) {
            T acc = T(0.0), acct = T(0.0);
#pragma unroll
            for (int l = 0; l < D; ++l) { acc += X[i * D + l] * F[j * D + l]; acct += X[j * D + l] * F[i * D + l]; }
            Q[symi<D>(i, j)] = T(0.5) * (Pinf[i * D + j] + Pinf[j * D + i]) - T(0.5) * (acc + acct);
        }
}

}  // namespace pgps
