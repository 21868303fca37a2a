/* CPU ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * Plain-C restatement of the reference's sequential Kalman filter and RTS smoother,
 * pssgp/kalman/sequential.py:11-73, in the same arithmetic order:
 *   kf  (11-47): predict mp = F m, Pp = F P F^T + Q, symmetrise (19-21); if y is not NaN:
 *                S = H Pp H^T + R, Cholesky of the 1x1 S, log N(y; H mp, S) added to ell,
 *                Kt = S^-1 H Pp, m = mp + Kt^T (y - H mp), P = Pp - Kt^T S Kt (23-34);
 *                symmetrise P (39).
 *   ks  (50-68): backwards, Ct = Pp^-1 F P by Cholesky (57-58), sm = m + Ct^T (sm - mp),
 *                sP = P + Ct^T (sP - Pp) Ct, symmetrise (59-61).
 * Used (a) as the large-N checker for the HIP scan (tests/, N up to 2^20 in about a second)
 * and (b) as bench.py's `cpu_baseline` ("port", 1 core).  It is pinned to the numpy oracle
 * (oracle/np_oracle.py, itself pinned to the dense GP) in tests/test_oracle.py.
 *
 * Build: make -C oracle   ->  oracle/_build/liboracle_seq.so
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ODMAX 32
#define LOG2PI 1.8378770664093453

#define DEFINE_ORACLE(SUF, T, SQRT)                                                                       \
    static void matmul_##SUF(int d, const T* A, const T* B, T* C, int tb) {                               \
        for (int i = 0; i < d; ++i)                                                                        \
            for (int j = 0; j < d; ++j) {                                                                  \
                T s = 0;                                                                                   \
                for (int k = 0; k < d; ++k) s += A[i * d + k] * (tb ? B[j * d + k] : B[k * d + j]);        \
                C[i * d + j] = s;                                                                          \
            }                                                                                              \
    }                                                                                                      \
    static void symm_##SUF(int d, T* A) {                                                                  \
        for (int i = 0; i < d; ++i)                                                                        \
            for (int j = i + 1; j < d; ++j) {                                                              \
                T s = (T)0.5 * (A[i * d + j] + A[j * d + i]);                                              \
                A[i * d + j] = s;                                                                          \
                A[j * d + i] = s;                                                                          \
            }                                                                                              \
    }                                                                                                      \
    /* lower Cholesky factor in place; returns 0 on success */                                            \
    static int chol_##SUF(int d, T* A) {                                                                   \
        for (int j = 0; j < d; ++j) {                                                                      \
            T s = A[j * d + j];                                                                            \
            for (int k = 0; k < j; ++k) s -= A[j * d + k] * A[j * d + k];                                  \
            if (!(s > 0)) return 1;                                                                        \
            A[j * d + j] = SQRT(s);                                                                        \
            for (int i = j + 1; i < d; ++i) {                                                              \
                T t = A[i * d + j];                                                                        \
                for (int k = 0; k < j; ++k) t -= A[i * d + k] * A[j * d + k];                              \
                A[i * d + j] = t / A[j * d + j];                                                           \
            }                                                                                              \
        }                                                                                                  \
        return 0;                                                                                          \
    }                                                                                                      \
    static void chol_solve_##SUF(int d, const T* L, T* B) {                                                \
        for (int c = 0; c < d; ++c) {                                                                      \
            for (int i = 0; i < d; ++i) {                                                                  \
                T t = B[i * d + c];                                                                        \
                for (int k = 0; k < i; ++k) t -= L[i * d + k] * B[k * d + c];                              \
                B[i * d + c] = t / L[i * d + i];                                                           \
            }                                                                                              \
            for (int i = d - 1; i >= 0; --i) {                                                             \
                T t = B[i * d + c];                                                                        \
                for (int k = i + 1; k < d; ++k) t -= L[k * d + i] * B[k * d + c];                          \
                B[i * d + c] = t / L[i * d + i];                                                           \
            }                                                                                              \
        }                                                                                                  \
    }                                                                                                      \
    /* sequential.py:11-47.  mps / Pps must be given (the smoother needs them). */                        \
    int oracle_kf_##SUF(long N, int d, const T* P0, const T* Fs, const T* Qs, const T* H, T R,            \
                        const T* ys, T* fms, T* fPs, T* mps, T* Pps, double* ll) {                         \
        if (d < 1 || d > ODMAX || N < 1) return -1;                                                        \
        const int dd = d * d;                                                                              \
        T m[ODMAX], P[ODMAX * ODMAX], mp[ODMAX], Pp[ODMAX * ODMAX], FP[ODMAX * ODMAX], HP[ODMAX];          \
        memset(m, 0, sizeof(m));                                                                           \
        memcpy(P, P0, sizeof(T) * dd);                                                                     \
        double ell = 0.0;                                                                                  \
        for (long k = 0; k < N; ++k) {                                                                     \
            const T* F = Fs + k * dd;                                                                      \
            const T* Q = Qs + k * dd;                                                                      \
            for (int i = 0; i < d; ++i) {                                                                  \
                T s = 0;                                                                                   \
                for (int j = 0; j < d; ++j) s += F[i * d + j] * m[j];                                      \
                mp[i] = s;                                                                                 \
            }                                                                                              \
            matmul_##SUF(d, F, P, FP, 0);                                                                  \
            matmul_##SUF(d, FP, F, Pp, 1);                                                                 \
            for (int i = 0; i < dd; ++i) Pp[i] += Q[i];                                                    \
            symm_##SUF(d, Pp);                                                                             \
            T y = ys[k];                                                                                   \
            if (y == y) {                                                                                  \
                T S = R, yp = 0;                                                                           \
                for (int j = 0; j < d; ++j) {                                                              \
                    T s = 0;                                                                               \
                    for (int i = 0; i < d; ++i) s += H[i] * Pp[i * d + j];                                 \
                    HP[j] = s;                                                                             \
                }                                                                                          \
                for (int i = 0; i < d; ++i) { S += HP[i] * H[i]; yp += H[i] * mp[i]; }                     \
                T chol = SQRT(S);                                                                          \
                double z = ((double)y - (double)yp) / (double)chol;                                        \
                ell += -0.5 * z * z - log((double)chol) - 0.5 * LOG2PI;                                    \
                for (int i = 0; i < d; ++i) {                                                              \
                    T Kt = HP[i] / chol / chol;                                                            \
                    m[i] = mp[i] + Kt * (y - yp);                                                          \
                }                                                                                          \
                for (int i = 0; i < d; ++i)                                                                \
                    for (int j = 0; j < d; ++j)                                                            \
                        P[i * d + j] = Pp[i * d + j] - (HP[i] / chol / chol) * S * (HP[j] / chol / chol);  \
            } else {                                                                                       \
                memcpy(m, mp, sizeof(T) * d);                                                              \
                memcpy(P, Pp, sizeof(T) * dd);                                                             \
            }                                                                                              \
            symm_##SUF(d, P);                                                                              \
            memcpy(fms + k * d, m, sizeof(T) * d);                                                         \
            memcpy(fPs + k * dd, P, sizeof(T) * dd);                                                       \
            memcpy(mps + k * d, mp, sizeof(T) * d);                                                        \
            memcpy(Pps + k * dd, Pp, sizeof(T) * dd);                                                      \
        }                                                                                                  \
        if (ll) *ll = ell;                                                                                 \
        return 0;                                                                                          \
    }                                                                                                      \
    /* sequential.py:50-68 */                                                                             \
    int oracle_ks_##SUF(long N, int d, const T* Fs, const T* fms, const T* fPs, const T* mps,             \
                        const T* Pps, T* sms, T* sPs) {                                                    \
        if (d < 1 || d > ODMAX || N < 1) return -1;                                                        \
        const int dd = d * d;                                                                              \
        T sm[ODMAX], sP[ODMAX * ODMAX], L[ODMAX * ODMAX], Ct[ODMAX * ODMAX], D[ODMAX * ODMAX],             \
            X[ODMAX * ODMAX], nm[ODMAX];                                                                   \
        memcpy(sm, fms + (N - 1) * d, sizeof(T) * d);                                                      \
        memcpy(sP, fPs + (N - 1) * dd, sizeof(T) * dd);                                                    \
        memcpy(sms + (N - 1) * d, sm, sizeof(T) * d);                                                      \
        memcpy(sPs + (N - 1) * dd, sP, sizeof(T) * dd);                                                    \
        for (long k = N - 2; k >= 0; --k) {                                                                \
            const T* F = Fs + (k + 1) * dd;                                                                \
            const T* P = fPs + k * dd;                                                                     \
            const T* Pp = Pps + (k + 1) * dd;                                                              \
            const T* mp = mps + (k + 1) * d;                                                               \
            memcpy(L, Pp, sizeof(T) * dd);                                                                 \
            if (chol_##SUF(d, L)) return -2;                                                               \
            matmul_##SUF(d, F, P, Ct, 0);                                                                  \
            chol_solve_##SUF(d, L, Ct);                                                                    \
            for (int i = 0; i < d; ++i) {                                                                  \
                T s = fms[k * d + i];                                                                      \
                for (int l = 0; l < d; ++l) s += Ct[l * d + i] * (sm[l] - mp[l]);                          \
                nm[i] = s;                                                                                 \
            }                                                                                              \
            for (int i = 0; i < dd; ++i) D[i] = sP[i] - Pp[i];                                             \
            for (int i = 0; i < d; ++i)                                                                    \
                for (int j = 0; j < d; ++j) {                                                              \
                    T s = 0;                                                                               \
                    for (int l = 0; l < d; ++l) s += Ct[l * d + i] * D[l * d + j];                         \
                    X[i * d + j] = s;                                                                      \
                }                                                                                          \
            matmul_##SUF(d, X, Ct, sP, 0);                                                                 \
            for (int i = 0; i < dd; ++i) sP[i] += P[i];                                                    \
            symm_##SUF(d, sP);                                                                             \
            memcpy(sm, nm, sizeof(T) * d);                                                                 \
            memcpy(sms + k * d, sm, sizeof(T) * d);                                                        \
            memcpy(sPs + k * dd, sP, sizeof(T) * dd);                                                      \
        }                                                                                                  \
        return 0;                                                                                          \
    }                                                                                                      \
    /* kfs (sequential.py:71-73) with scratch for the predicted moments */                                \
    int oracle_kfs_##SUF(long N, int d, const T* P0, const T* Fs, const T* Qs, const T* H, T R,           \
                         const T* ys, T* fms, T* fPs, T* sms, T* sPs, double* ll) {                        \
        T* mps = (T*)malloc(sizeof(T) * (size_t)N * d);                                                    \
        T* Pps = (T*)malloc(sizeof(T) * (size_t)N * d * d);                                                \
        if (!mps || !Pps) { free(mps); free(Pps); return -3; }                                             \
        int rc = oracle_kf_##SUF(N, d, P0, Fs, Qs, H, R, ys, fms, fPs, mps, Pps, ll);                      \
        if (!rc) rc = oracle_ks_##SUF(N, d, Fs, fms, fPs, mps, Pps, sms, sPs);                             \
        free(mps);                                                                                         \
        free(Pps);                                                                                         \
        return rc;                                                                                         \
    }

DEFINE_ORACLE(f64, double, sqrt)
DEFINE_ORACLE(f32, float, sqrtf)
