// pgps_seq_host.cpp -- the reference's SEQUENTIAL Kalman filter / RTS smoother
// (pssgp/kalman/sequential.py:11-73; StateSpaceGP(parallel=False), pssgp/model.py:76-79) as
// host C++.  The reference runs this mode on the CPU too (`--device=/cpu:0`,
// experiments/toy_models/speed_and_stability.sh:8).  It is an explicit mode of the API, not a
// fallback: nothing on the parallel=True path ever reaches this file.
#include <cmath>
#include <vector>

#include "../../include/pgps.h"

namespace {

template <typename T>
struct Seq {
    int d;
    std::vector<T> FP, tmp;
    explicit Seq(int d_) : d(d_), FP((size_t)d_ * d_), tmp((size_t)d_ * d_) {}

    // Pp = sym(F P F^T + Q)
    void predict(const T* F, const T* P, const T* Q, T* Pp) {
        for (int i = 0; i < d; ++i)
            for (int j = 0; j < d; ++j) {
                T acc = 0;
                for (int k = 0; k < d; ++k) acc += F[i * d + k] * P[k * d + j];
                FP[i * d + j] = acc;
            }
        for (int i = 0; i < d; ++i)
            for (int j = 0; j < d; ++j) {
                T acc = Q[i * d + j];
                for (int k = 0; k < d; ++k) acc += FP[i * d + k] * F[j * d + k];
                tmp[i * d + j] = acc;
            }
        for (int i = 0; i < d; ++i)
            for (int j = 0; j < d; ++j) Pp[i * d + j] = T(0.5) * (tmp[i * d + j] + tmp[j * d + i]);
    }
};

// sequential.py:11-47
template <typename T>
int seq_kf(long N, int d, const T* P0, const T* Fs, const T* Qs, const T* H, T R, const T* ys, T* fms, T* fPs,
           double* ll, T* mps, T* Pps) {
    if (N < 1 || d < 1 || !P0 || !Fs || !Qs || !H || !ys || !fms || !fPs) return PGPS_E_INVALID;
    const size_t dd = (size_t)d * d;
    Seq<T> w(d);
    std::vector<T> m(d, T(0)), P(P0, P0 + dd), mp(d), Pp(dd), u(d);
    double ell = 0.0;
    for (long k = 0; k < N; ++k) {
        const T* F = Fs + k * dd;
        const T* Q = Qs + k * dd;
        for (int i = 0; i < d; ++i) {
            T acc = 0;
            for (int j = 0; j < d; ++j) acc += F[i * d + j] * m[j];
            mp[i] = acc;
        }
        w.predict(F, P.data(), Q, Pp.data());
        const T y = ys[k];
        if (y == y) {
            T S = R, yp = 0;
            for (int i = 0; i < d; ++i) {
                T acc = 0;
                for (int j = 0; j < d; ++j) acc += Pp[i * d + j] * H[j];
                u[i] = acc;
            }
            for (int i = 0; i < d; ++i) { S += H[i] * u[i]; yp += H[i] * mp[i]; }
            const double r = double(y) - double(yp);
            ell += -0.5 * (1.8378770664093453 + std::log(double(S)) + r * r / double(S));
            for (int i = 0; i < d; ++i) m[i] = mp[i] + u[i] / S * (y - yp);
            for (int i = 0; i < d; ++i)
                for (int j = 0; j < d; ++j) P[i * d + j] = Pp[i * d + j] - u[i] * u[j] / S;
        } else {
            m = mp;
            P = Pp;
        }
        for (int i = 0; i < d; ++i)
            for (int j = i + 1; j < d; ++j) {
                const T s = T(0.5) * (P[i * d + j] + P[j * d + i]);
                P[i * d + j] = s;
                P[j * d + i] = s;
            }
        for (int i = 0; i < d; ++i) fms[k * d + i] = m[i];
        for (size_t i = 0; i < dd; ++i) fPs[k * dd + i] = P[i];
        if (mps) for (int i = 0; i < d; ++i) mps[k * d + i] = mp[i];
        if (Pps) for (size_t i = 0; i < dd; ++i) Pps[k * dd + i] = Pp[i];
    }
    if (ll) *ll = ell;
    return std::isfinite(ell) ? PGPS_OK : PGPS_E_NUMERIC;
}

// Cholesky solve  X = A^-1 B  (A SPD d x d, B d x d), in place in B.  Returns false if A is not PD.
template <typename T>
bool chol_solve(int d, std::vector<T>& A, T* B) {
    for (int j = 0; j < d; ++j) {
        T s = A[j * d + j];
        for (int k = 0; k < j; ++k) s -= A[j * d + k] * A[j * d + k];
        if (!(s > 0)) return false;
        const T l = std::sqrt(s);
        A[j * d + j] = l;
        for (int i = j + 1; i < d; ++i) {
            T t = A[i * d + j];
            for (int k = 0; k < j; ++k) t -= A[i * d + k] * A[j * d + k];
            A[i * d + j] = t / l;
        }
    }
    for (int c = 0; c < d; ++c) {
        for (int i = 0; i < d; ++i) {
            T t = B[i * d + c];
            for (int k = 0; k < i; ++k) t -= A[i * d + k] * B[k * d + c];
            B[i * d + c] = t / A[i * d + i];
        }
        for (int i = d - 1; i >= 0; --i) {
            T t = B[i * d + c];
            for (int k = i + 1; k < d; ++k) t -= A[k * d + i] * B[k * d + c];
            B[i * d + c] = t / A[i * d + i];
        }
    }
    return true;
}

// sequential.py:50-68
template <typename T>
int seq_ks(long N, int d, const T* Fs, const T* ms, const T* Ps, const T* mps, const T* Pps, T* sms, T* sPs) {
    if (N < 1 || d < 1 || !Fs || !ms || !Ps || !mps || !Pps || !sms || !sPs) return PGPS_E_INVALID;
    const size_t dd = (size_t)d * d;
    std::vector<T> A(dd), Ct(dd), D(dd), X(dd), sm(ms + (N - 1) * d, ms + N * d),
        sP(Ps + (N - 1) * dd, Ps + N * dd);
    for (int i = 0; i < d; ++i) sms[(N - 1) * d + i] = sm[i];
    for (size_t i = 0; i < dd; ++i) sPs[(N - 1) * dd + i] = sP[i];
    for (long k = N - 2; k >= 0; --k) {
        const T* F = Fs + (k + 1) * dd;
        const T* P = Ps + k * dd;
        const T* Pp = Pps + (k + 1) * dd;
        const T* mp = mps + (k + 1) * d;
        for (size_t i = 0; i < dd; ++i) A[i] = Pp[i];
        for (int i = 0; i < d; ++i)
            for (int j = 0; j < d; ++j) {
                T acc = 0;
                for (int l = 0; l < d; ++l) acc += F[i * d + l] * P[l * d + j];
                Ct[i * d + j] = acc;
            }
        if (!chol_solve(d, A, Ct.data())) return PGPS_E_NUMERIC;      // Ct = Pp^-1 F P
        for (int i = 0; i < d; ++i) {
            T acc = ms[k * d + i];
            for (int l = 0; l < d; ++l) acc += Ct[l * d + i] * (sm[l] - mp[l]);
            X[i] = acc;
        }
        for (size_t i = 0; i < dd; ++i) D[i] = sP[i] - Pp[i];
        for (int i = 0; i < d; ++i) sm[i] = X[i];
        for (int i = 0; i < d; ++i)            // X = Ct^T D
            for (int j = 0; j < d; ++j) {
                T acc = 0;
                for (int l = 0; l < d; ++l) acc += Ct[l * d + i] * D[l * d + j];
                X[i * d + j] = acc;
            }
        for (int i = 0; i < d; ++i)
            for (int j = 0; j < d; ++j) {
                T acc = P[i * d + j];
                for (int l = 0; l < d; ++l) acc += X[i * d + l] * Ct[l * d + j];
                A[i * d + j] = acc;
            }
        for (int i = 0; i < d; ++i)
            for (int j = 0; j < d; ++j) sP[i * d + j] = T(0.5) * (A[i * d + j] + A[j * d + i]);
        for (int i = 0; i < d; ++i) sms[k * d + i] = sm[i];
        for (size_t i = 0; i < dd; ++i) sPs[k * dd + i] = sP[i];
    }
    return PGPS_OK;
}

}  // namespace

extern "C" int pgps_seq_kf_f64(long N, int d, const double* P0, const double* Fs, const double* Qs, const double* H,
                               double R, const double* ys, double* fms, double* fPs, double* ll, double* mps,
                               double* Pps) {
    return seq_kf<double>(N, d, P0, Fs, Qs, H, R, ys, fms, fPs, ll, mps, Pps);
}
extern "C" int pgps_seq_kf_f32(long N, int d, const float* P0, const float* Fs, const float* Qs, const float* H,
                               float R, const float* ys, float* fms, float* fPs, double* ll, float* mps, float* Pps) {
    return seq_kf<float>(N, d, P0, Fs, Qs, H, R, ys, fms, fPs, ll, mps, Pps);
}
extern "C" int pgps_seq_ks_f64(long N, int d, const double* Fs, const double* ms, const double* Ps, const double* mps,
                               const double* Pps, double* sms, double* sPs) {
    return seq_ks<double>(N, d, Fs, ms, Ps, mps, Pps, sms, sPs);
}
extern "C" int pgps_seq_ks_f32(long N, int d, const float* Fs, const float* ms, const float* Ps, const float* mps,
                               const float* Pps, float* sms, float* sPs) {
    return seq_ks<float>(N, d, Fs, ms, Ps, mps, Pps, sms, sPs);
}

// The sweep of balance_ss (pssgp/kernels/math_utils.py:10-29, a numba loop in the reference): n_iter passes over the
// states, each visit equalising the off-diagonal column and row 2-norms of the progressively rescaled matrix; returns
// the accumulated diagonal scaling.  It is the inner loop of every get_sde() of a composite kernel, i.e. of every
// hyper-parameter setting an optimiser or sampler visits.  0/0 (an isolated state) gives NaN, as in the reference.
extern "C" int pgps_host_balance_f64(int d, const double* F, int n_iter, double* scale) {
    if (d < 1 || !F || !scale || n_iter < 0) return PGPS_E_INVALID;
    std::vector<double> W((size_t)d * d);
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) W[(size_t)i * d + j] = (i == j) ? 0.0 : F[(size_t)i * d + j];
    for (int i = 0; i < d; ++i) scale[i] = 1.0;
    for (int it = 0; it < n_iter; ++it)
        for (int i = 0; i < d; ++i) {
            double c = 0.0, r = 0.0;
            for (int k = 0; k < d; ++k) {
                c += W[(size_t)k * d + i] * W[(size_t)k * d + i];
                r += W[(size_t)i * d + k] * W[(size_t)i * d + k];
            }
            const double f = std::pow(r / c, 0.25);
            scale[i] *= f;
            for (int k = 0; k < d; ++k) {
                W[(size_t)k * d + i] *= f;
                W[(size_t)i * d + k] /= f;
            }
        }
    return PGPS_OK;
}
