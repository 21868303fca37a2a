/* CPU ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * The same filter + smoother + log-likelihood as kalman_seq.c on ALL host cores: a chunked scan with OpenMP, the
 * "all cores, for context" CPU baseline of SURVEY.md 8(d).  It restates the algebra of the reference's parallel path
 * (pssgp/kalman/parallel.py) at chunk granularity:
 *   filter   pass 1 (parallel)  each chunk's total as ONE filtering element (A, b, C, J, eta): generic elements
 *                               (parallel.py:56-72, NaN variant 46-53) folded left to right with the filtering
 *                               operator (100-118);
 *            pass 2 (serial)    the state entering every chunk: the total applied to (m, P) -- the operator with the
 *                               left element (0, m, P, 0, 0), which is what every prefix that contains step 0 is;
 *            pass 3 (parallel)  the sequential filter of kalman_seq.c inside the chunk from that state (same
 *                               arithmetic order as sequential.py:11-47), log-likelihood terms summed per chunk;
 *   smoother pass 1 (parallel)  each chunk's total as ONE smoothing element (E, g, L) (parallel.py:155-166) folded with
 *                               the smoothing operator (176-184) from the right;
 *            pass 2 (serial)    the smoothed state of the first step after every chunk;
 *            pass 3 (parallel)  sequential RTS steps inside the chunk (sequential.py:50-68).
 * About 3x the arithmetic of the sequential path, spread over the cores.  fp64 only, d <= 16.
 * Pinned to kalman_seq.c (and through it to the numpy oracle) in tests/test_oracle.py; timed by bench.py as
 * `cpu_baseline_all_cores`.
 *
 * Build: make -C oracle   ->  oracle/_build/liboracle_par.so   (gcc -fopenmp)
 */
#include <math.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>

#define PDMAX 16
#define PDD (PDMAX * PDMAX)
#define LOG2PI 1.8378770664093453

typedef struct { double A[PDD], b[PDMAX], C[PDD], J[PDD], eta[PDMAX]; } felem;
typedef struct { double E[PDD], g[PDMAX], L[PDD]; } selem;

static void mm(int d, const double* A, const double* B, double* C, int tb) {
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) {
            double s = 0;
            for (int k = 0; k < d; ++k) s += A[i * d + k] * (tb ? B[j * d + k] : B[k * d + j]);
            C[i * d + j] = s;
        }
}
static void mv(int d, const double* A, const double* x, double* y) {
    for (int i = 0; i < d; ++i) {
        double s = 0;
        for (int k = 0; k < d; ++k) s += A[i * d + k] * x[k];
        y[i] = s;
    }
}
static void symm(int d, double* A) {
    for (int i = 0; i < d; ++i)
        for (int j = i + 1; j < d; ++j) {
            double s = 0.5 * (A[i * d + j] + A[j * d + i]);
            A[i * d + j] = s;
            A[j * d + i] = s;
        }
}
/* X <- M^-1 X for an n-column right-hand side, Gaussian elimination with partial pivoting (M is destroyed) */
static int solve(int d, double* M, double* X, int n) {
    for (int c = 0; c < d; ++c) {
        int p = c;
        for (int r = c + 1; r < d; ++r)
            if (fabs(M[r * d + c]) > fabs(M[p * d + c])) p = r;
        if (M[p * d + c] == 0.0) return 1;
        if (p != c) {
            for (int j = 0; j < d; ++j) { double t = M[c * d + j]; M[c * d + j] = M[p * d + j]; M[p * d + j] = t; }
            for (int j = 0; j < n; ++j) { double t = X[c * n + j]; X[c * n + j] = X[p * n + j]; X[p * n + j] = t; }
        }
        const double inv = 1.0 / M[c * d + c];
        for (int r = 0; r < d; ++r) {
            if (r == c) continue;
            const double f = M[r * d + c] * inv;
            if (f == 0.0) continue;
            for (int j = c; j < d; ++j) M[r * d + j] -= f * M[c * d + j];
            for (int j = 0; j < n; ++j) X[r * n + j] -= f * X[c * n + j];
        }
    }
    for (int r = 0; r < d; ++r) {
        const double inv = 1.0 / M[r * d + r];
        for (int j = 0; j < n; ++j) X[r * n + j] *= inv;
    }
    return 0;
}

/* parallel.py:56-72 (observed) / 46-53 (missing) */
static void generic_element(int d, const double* F, const double* Q, const double* H, double R, double y, felem* e) {
    const int dd = d * d;
    if (!(y == y)) {
        memcpy(e->A, F, sizeof(double) * dd);
        memcpy(e->C, Q, sizeof(double) * dd);
        memset(e->b, 0, sizeof(double) * d);
        memset(e->J, 0, sizeof(double) * dd);
        memset(e->eta, 0, sizeof(double) * d);
        return;
    }
    double HQ[PDMAX], HF[PDMAX], S = R;
    for (int j = 0; j < d; ++j) {
        double s = 0, t = 0;
        for (int i = 0; i < d; ++i) { s += H[i] * Q[i * d + j]; t += H[i] * F[i * d + j]; }
        HQ[j] = s;
        HF[j] = t;
    }
    for (int i = 0; i < d; ++i) S += HQ[i] * H[i];
    for (int i = 0; i < d; ++i) {
        const double K = HQ[i] / S;                         /* Q symmetric: (Q H^T)_i = (H Q)_i */
        e->b[i] = K * y;
        e->eta[i] = HF[i] * y / S;
        for (int j = 0; j < d; ++j) {
            e->A[i * d + j] = F[i * d + j] - K * HF[j];
            e->C[i * d + j] = Q[i * d + j] - K * HQ[j];
            e->J[i * d + j] = HF[i] * HF[j] / S;
        }
    }
}

/* parallel.py:100-118: out = e1 (x) e2 (e1 earlier).  One factorisation: (I + J2 C1)^-T = (I + C1 J2)^-1 transposed. */
static int f_combine(int d, const felem* e1, const felem* e2, felem* o) {
    const int dd = d * d;
    double M[PDD], X[PDD * 2 + PDMAX], T1[PDD], T2[PDD], v[PDMAX], w[PDMAX];
    /* M = I + C1 J2;  solve M [A1 | C1 | b1 + C1 eta2] */
    mm(d, e1->C, e2->J, M, 0);
    for (int i = 0; i < d; ++i) M[i * d + i] += 1.0;
    mv(d, e1->C, e2->eta, v);
    const int n = 2 * d + 1;
    for (int i = 0; i < d; ++i) {
        for (int j = 0; j < d; ++j) { X[i * n + j] = e1->A[i * d + j]; X[i * n + d + j] = e1->C[i * d + j]; }
        X[i * n + 2 * d] = e1->b[i] + v[i];
    }
    if (solve(d, M, X, n)) return 1;
    double G[PDD], Nm[PDD], u[PDMAX];
    for (int i = 0; i < d; ++i) {
        for (int j = 0; j < d; ++j) { G[i * d + j] = X[i * n + j]; Nm[i * d + j] = X[i * n + d + j]; }
        u[i] = X[i * n + 2 * d];
    }
    /* A = A2 G;  b = A2 u + b2;  C = A2 Nm A2^T + C2 */
    mm(d, e2->A, G, T1, 0);
    mv(d, e2->A, u, w);
    mm(d, e2->A, Nm, T2, 0);
    double Cn[PDD];
    mm(d, T2, e2->A, Cn, 1);
    /* eta = A1^T (I + J2 C1)^-1 (eta2 - J2 b1) + eta1;  J = A1^T (I + J2 C1)^-1 J2 A1 + J1,
       with (I + J2 C1)^-1 J2 = J2 (I + C1 J2)^-1  =>  J = A1^T J2 G + J1,  eta = A1^T (eta2 - J2 u') ... written out:
       (I + J2 C1)^-1 (eta2 - J2 b1) = eta2 - J2 (I + C1 J2)^-1 (b1 + C1 eta2) = eta2 - J2 u */
    double J2u[PDMAX], r[PDMAX], J2G[PDD], Jn[PDD];
    mv(d, e2->J, u, J2u);
    for (int i = 0; i < d; ++i) r[i] = e2->eta[i] - J2u[i];
    mm(d, e2->J, G, J2G, 0);
    for (int i = 0; i < d; ++i) {
        double s = e1->eta[i];
        for (int k = 0; k < d; ++k) s += e1->A[k * d + i] * r[k];
        o->eta[i] = s;
        for (int j = 0; j < d; ++j) {
            double t = e1->J[i * d + j];
            for (int k = 0; k < d; ++k) t += e1->A[k * d + i] * J2G[k * d + j];
            Jn[i * d + j] = t;
        }
    }
    for (int i = 0; i < dd; ++i) { o->A[i] = T1[i]; o->C[i] = Cn[i] + e2->C[i]; o->J[i] = Jn[i]; }
    for (int i = 0; i < d; ++i) o->b[i] = w[i] + e2->b[i];
    symm(d, o->C);
    symm(d, o->J);
    return 0;
}

/* (m, P) pushed through an element: the operator with the left element (0, m, P, 0, 0) */
static int f_apply(int d, const felem* e, double* m, double* P) {
    double M[PDD], X[PDD + PDMAX], v[PDMAX], T2[PDD], Pn[PDD], mn[PDMAX];
    mm(d, P, e->J, M, 0);
    for (int i = 0; i < d; ++i) M[i * d + i] += 1.0;
    mv(d, P, e->eta, v);
    const int n = d + 1;
    for (int i = 0; i < d; ++i) {
        for (int j = 0; j < d; ++j) X[i * n + j] = P[i * d + j];
        X[i * n + d] = m[i] + v[i];
    }
    if (solve(d, M, X, n)) return 1;
    double Nm[PDD], u[PDMAX];
    for (int i = 0; i < d; ++i) {
        for (int j = 0; j < d; ++j) Nm[i * d + j] = X[i * n + j];
        u[i] = X[i * n + d];
    }
    mv(d, e->A, u, mn);
    mm(d, e->A, Nm, T2, 0);
    mm(d, T2, e->A, Pn, 1);
    for (int i = 0; i < d; ++i) m[i] = mn[i] + e->b[i];
    for (int i = 0; i < d * d; ++i) P[i] = Pn[i] + e->C[i];
    symm(d, P);
    return 0;
}

static int chol(int d, double* A) {
    for (int j = 0; j < d; ++j) {
        double s = A[j * d + j];
        for (int k = 0; k < j; ++k) s -= A[j * d + k] * A[j * d + k];
        if (!(s > 0)) return 1;
        A[j * d + j] = sqrt(s);
        for (int i = j + 1; i < d; ++i) {
            double t = A[i * d + j];
            for (int k = 0; k < j; ++k) t -= A[i * d + k] * A[j * d + k];
            A[i * d + j] = t / A[j * d + j];
        }
    }
    return 0;
}
static void chol_solve(int d, const double* L, double* B) {
    for (int c = 0; c < d; ++c) {
        for (int i = 0; i < d; ++i) {
            double t = B[i * d + c];
            for (int k = 0; k < i; ++k) t -= L[i * d + k] * B[k * d + c];
            B[i * d + c] = t / L[i * d + i];
        }
        for (int i = d - 1; i >= 0; --i) {
            double t = B[i * d + c];
            for (int k = i + 1; k < d; ++k) t -= L[k * d + i] * B[k * d + c];
            B[i * d + c] = t / L[i * d + i];
        }
    }
}

/* sequential.py:11-47 over steps [k0, k1) from (m, P); same arithmetic order as kalman_seq.c */
static double kf_range(long k0, long k1, int d, const double* Fs, const double* Qs, const double* H, double R,
                       const double* ys, double* m, double* P, double* fms, double* fPs, double* mps, double* Pps) {
    const int dd = d * d;
    double mp[PDMAX], Pp[PDD], FP[PDD], HP[PDMAX], ell = 0.0;
    for (long k = k0; k < k1; ++k) {
        const double *F = Fs + k * dd, *Q = Qs + k * dd;
        mv(d, F, m, mp);
        mm(d, F, P, FP, 0);
        mm(d, FP, F, Pp, 1);
        for (int i = 0; i < dd; ++i) Pp[i] += Q[i];
        symm(d, Pp);
        const double y = ys[k];
        if (y == y) {
            double S = R, yp = 0;
            for (int j = 0; j < d; ++j) {
                double s = 0;
                for (int i = 0; i < d; ++i) s += H[i] * Pp[i * d + j];
                HP[j] = s;
            }
            for (int i = 0; i < d; ++i) { S += HP[i] * H[i]; yp += H[i] * mp[i]; }
            const double c = sqrt(S), z = (y - yp) / c;
            ell += -0.5 * z * z - log(c) - 0.5 * LOG2PI;
            for (int i = 0; i < d; ++i) m[i] = mp[i] + HP[i] / c / c * (y - yp);
            for (int i = 0; i < d; ++i)
                for (int j = 0; j < d; ++j) P[i * d + j] = Pp[i * d + j] - (HP[i] / c / c) * S * (HP[j] / c / c);
        } else {
            memcpy(m, mp, sizeof(double) * d);
            memcpy(P, Pp, sizeof(double) * dd);
        }
        symm(d, P);
        memcpy(fms + k * d, m, sizeof(double) * d);
        memcpy(fPs + k * dd, P, sizeof(double) * dd);
        memcpy(mps + k * d, mp, sizeof(double) * d);
        memcpy(Pps + k * dd, Pp, sizeof(double) * dd);
    }
    return ell;
}

/* the RTS gain of step k (< N-1): Ct = Pp_{k+1}^-1 F_{k+1} P_k  (gain = Ct^T), sequential.py:57-58 */
static int rts_gain(long k, int d, const double* Fs, const double* fPs, const double* Pps, double* Ct) {
    const int dd = d * d;
    double L[PDD];
    memcpy(L, Pps + (k + 1) * dd, sizeof(double) * dd);
    if (chol(d, L)) return 1;
    mm(d, Fs + (k + 1) * dd, fPs + k * dd, Ct, 0);
    chol_solve(d, L, Ct);
    return 0;
}

int oracle_par_kfs_f64(long N, int d, const double* P0, const double* Fs, const double* Qs, const double* H, double R,
                       const double* ys, double* fms, double* fPs, double* sms, double* sPs, double* ll,
                       int nthreads) {
    if (d < 1 || d > PDMAX || N < 1) return -1;
    if (nthreads < 1) nthreads = omp_get_max_threads();
    const int dd = d * d;
    long nchunk = (long)nthreads * 4;
    if (nchunk > N) nchunk = N;
    const long per = (N + nchunk - 1) / nchunk;
    nchunk = (N + per - 1) / per;
    felem* fagg = (felem*)malloc(sizeof(felem) * (size_t)nchunk);
    selem* sagg = (selem*)malloc(sizeof(selem) * (size_t)nchunk);
    double* m_in = (double*)malloc(sizeof(double) * (size_t)nchunk * d);
    double* P_in = (double*)malloc(sizeof(double) * (size_t)nchunk * dd);
    double* ells = (double*)calloc((size_t)nchunk, sizeof(double));
    double* mps = (double*)malloc(sizeof(double) * (size_t)N * d);
    double* Pps = (double*)malloc(sizeof(double) * (size_t)N * dd);
    int bad = 0;
    if (!fagg || !sagg || !m_in || !P_in || !ells || !mps || !Pps) { bad = -3; goto done; }

    /* ---- filter, pass 1: chunk totals (the last chunk's is never used) ---- */
#pragma omp parallel for schedule(static) num_threads(nthreads) reduction(| : bad)
    for (long c = 0; c < nchunk - 1; ++c) {
        const long k0 = c * per, k1 = (k0 + per < N) ? k0 + per : N;
        felem cur, nxt, acc;
        generic_element(d, Fs + k0 * dd, Qs + k0 * dd, H, R, ys[k0], &acc);
        for (long k = k0 + 1; k < k1; ++k) {
            generic_element(d, Fs + k * dd, Qs + k * dd, H, R, ys[k], &cur);
            bad |= f_combine(d, &acc, &cur, &nxt);
            acc = nxt;
        }
        fagg[c] = acc;
    }
    if (bad) { bad = -2; goto done; }
    /* ---- pass 2: state entering each chunk ---- */
    memset(m_in, 0, sizeof(double) * d);
    memcpy(P_in, P0, sizeof(double) * dd);
    for (long c = 1; c < nchunk; ++c) {
        memcpy(m_in + c * d, m_in + (c - 1) * d, sizeof(double) * d);
        memcpy(P_in + c * dd, P_in + (c - 1) * dd, sizeof(double) * dd);
        if (f_apply(d, &fagg[c - 1], m_in + c * d, P_in + c * dd)) { bad = -2; goto done; }
    }
    /* ---- pass 3: sequential filter inside the chunks ---- */
#pragma omp parallel for schedule(static) num_threads(nthreads)
    for (long c = 0; c < nchunk; ++c) {
        const long k0 = c * per, k1 = (k0 + per < N) ? k0 + per : N;
        double m[PDMAX], P[PDD];
        memcpy(m, m_in + c * d, sizeof(double) * d);
        memcpy(P, P_in + c * dd, sizeof(double) * dd);
        ells[c] = kf_range(k0, k1, d, Fs, Qs, H, R, ys, m, P, fms, fPs, mps, Pps);
    }
    if (ll) {
        double t = 0.0;
        for (long c = 0; c < nchunk; ++c) t += ells[c];
        *ll = t;
    }

    /* ---- smoother, pass 1: chunk totals from the right (the first chunk's is never used) ---- */
#pragma omp parallel for schedule(static) num_threads(nthreads) reduction(| : bad)
    for (long c = 1; c < nchunk; ++c) {
        const long k0 = c * per, k1 = (k0 + per < N) ? k0 + per : N;
        selem acc;
        double Ct[PDD], T1[PDD], T2[PDD], v[PDMAX];
        for (long k = k1 - 1; k >= k0; --k) {
            if (k == N - 1) {                               /* parallel.py:155-156 */
                memset(acc.E, 0, sizeof(double) * dd);
                memcpy(acc.g, fms + k * d, sizeof(double) * d);
                memcpy(acc.L, fPs + k * dd, sizeof(double) * dd);
                continue;
            }
            bad |= rts_gain(k, d, Fs, fPs, Pps, Ct);
            /* element of step k: E = Ct^T, g = m - E mp', L = P - E Pp' E^T (parallel.py:159-166) */
            double E[PDD], g[PDMAX], L[PDD];
            for (int i = 0; i < d; ++i)
                for (int j = 0; j < d; ++j) E[i * d + j] = Ct[j * d + i];
            mv(d, E, mps + (k + 1) * d, v);
            for (int i = 0; i < d; ++i) g[i] = fms[k * d + i] - v[i];
            mm(d, E, Pps + (k + 1) * dd, T1, 0);
            mm(d, T1, E, T2, 1);
            for (int i = 0; i < dd; ++i) L[i] = fPs[k * dd + i] - T2[i];
            symm(d, L);
            if (k == k1 - 1) {
                memcpy(acc.E, E, sizeof(E)); memcpy(acc.g, g, sizeof(g)); memcpy(acc.L, L, sizeof(L));
            } else {                                        /* parallel.py:176-184: step k applied after the total so far */
                mv(d, E, acc.g, v);
                mm(d, E, acc.L, T1, 0);
                mm(d, T1, E, T2, 1);
                mm(d, E, acc.E, T1, 0);
                for (int i = 0; i < d; ++i) acc.g[i] = v[i] + g[i];
                for (int i = 0; i < dd; ++i) { acc.L[i] = T2[i] + L[i]; acc.E[i] = T1[i]; }
                symm(d, acc.L);
            }
        }
        sagg[c] = acc;
    }
    if (bad) { bad = -2; goto done; }
    /* ---- pass 2: smoothed state of the first step after each chunk (m_in / P_in reused) ---- */
    for (long c = nchunk - 2; c >= 0; --c) {
        const selem* a = &sagg[c + 1];
        double* sm = m_in + c * d;
        double* sP = P_in + c * dd;
        if (c == nchunk - 2) {                              /* E of the last chunk's total is 0 */
            memcpy(sm, a->g, sizeof(double) * d);
            memcpy(sP, a->L, sizeof(double) * dd);
        } else {
            double T1[PDD], T2[PDD], v[PDMAX];
            mv(d, a->E, m_in + (c + 1) * d, v);
            mm(d, a->E, P_in + (c + 1) * dd, T1, 0);
            mm(d, T1, a->E, T2, 1);
            for (int i = 0; i < d; ++i) sm[i] = v[i] + a->g[i];
            for (int i = 0; i < dd; ++i) sP[i] = T2[i] + a->L[i];
            symm(d, sP);
        }
    }
    /* ---- pass 3: sequential RTS inside the chunks (sequential.py:50-68) ---- */
#pragma omp parallel for schedule(static) num_threads(nthreads) reduction(| : bad)
    for (long c = 0; c < nchunk; ++c) {
        const long k0 = c * per, k1 = (k0 + per < N) ? k0 + per : N;
        double sm[PDMAX], sP[PDD], Ct[PDD], D[PDD], X[PDD], nm[PDMAX];
        long k = k1 - 1;
        if (k1 == N) {
            memcpy(sm, fms + (N - 1) * d, sizeof(double) * d);
            memcpy(sP, fPs + (N - 1) * dd, sizeof(double) * dd);
            memcpy(sms + (N - 1) * d, sm, sizeof(double) * d);
            memcpy(sPs + (N - 1) * dd, sP, sizeof(double) * dd);
            --k;
        } else {
            memcpy(sm, m_in + c * d, sizeof(double) * d);
            memcpy(sP, P_in + c * dd, sizeof(double) * dd);
        }
        for (; k >= k0; --k) {
            bad |= rts_gain(k, d, Fs, fPs, Pps, Ct);
            const double *Pp = Pps + (k + 1) * dd, *mp = mps + (k + 1) * d;
            for (int i = 0; i < d; ++i) {
                double s = fms[k * d + i];
                for (int l = 0; l < d; ++l) s += Ct[l * d + i] * (sm[l] - mp[l]);
                nm[i] = s;
            }
            for (int i = 0; i < dd; ++i) D[i] = sP[i] - Pp[i];
            for (int i = 0; i < d; ++i)
                for (int j = 0; j < d; ++j) {
                    double s = 0;
                    for (int l = 0; l < d; ++l) s += Ct[l * d + i] * D[l * d + j];
                    X[i * d + j] = s;
                }
            mm(d, X, Ct, sP, 0);
            for (int i = 0; i < dd; ++i) sP[i] += fPs[k * dd + i];
            symm(d, sP);
            memcpy(sm, nm, sizeof(double) * d);
            memcpy(sms + k * d, sm, sizeof(double) * d);
            memcpy(sPs + k * dd, sP, sizeof(double) * dd);
        }
    }
    if (bad) bad = -2;
done:
    free(fagg); free(sagg); free(m_in); free(P_in); free(ells); free(mps); free(Pps);
    return bad;
}

int oracle_par_max_threads(void) { return omp_get_max_threads(); }
