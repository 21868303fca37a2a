// CPU check of the device algebra in parallel-gps_amd/csrc/pgps_math.h (TEST TOOL, not a
// product path): runs the same three-phase chunked scan the HIP kernels run --
//   (1) per-"lane" chunk aggregates by filt_extend, (2) Kogge-Stone scans over groups of W
//   lanes with filt_combine + a fold across groups, (3) carry-in by filt_apply and a lane-serial
//   Kalman pass that also builds the smoothing aggregates, then the mirrored suffix scan and
//   RTS pass -- as plain loops on the host, so the math can be compared with the numpy oracle
//   without a GPU.  Built by tests/test_cpu_math.py with g++.
#include <cstring>
#include <vector>

#include "pgps_math.h"
#include "pgps_dual.h"

using namespace pgps;

template <typename T, int D>
static void load_sym(const T* full, T* sym) {
    for (int i = 0; i < D; ++i)
        for (int j = i; j < D; ++j) sym[symi<D>(i, j)] = T(0.5) * (full[i * D + j] + full[j * D + i]);
}
template <typename T, int D>
static void store_sym(const T* sym, T* full) {
    for (int i = 0; i < D; ++i)
        for (int j = 0; j < D; ++j) full[i * D + j] = sym[symi<D>(i, j)];
}

// DFORM: the smoothing totals in innovation form (pgps_math.h kf_step_u / smth_extend_u), as the whole-series pkfs kernels of
// the lane-chunk family keep them: the carry of a chunk is (sm - m, sP - P) and the filtered moments are added back
template <typename T, int D, bool DFORM = false>
static int run(long N, int Lc, int W, const T* P0f, const T* Fs, const T* Qs, const T* h, T R,
               const T* ys, T* fms, T* fPs, T* sms, T* sPs, double* ll_out) {
    constexpr int MAT = D * D, SYM = Dim<D>::SYM;
    const long nth = (N + Lc - 1) / Lc;
    T P0[SYM];
    load_sym<T, D>(P0f, P0);
    std::vector<FiltElem<T, D>> agg(nth), incl(nth), excl(nth);
    // phase 1
    for (long t = 0; t < nth; ++t) {
        filt_identity(agg[t]);
        for (long k = t * Lc; k < std::min<long>(N, (t + 1) * Lc); ++k) {
            T Q[SYM];
            load_sym<T, D>(Qs + k * MAT, Q);
            if (k == 0) filt_first(agg[t], P0, ys[0], h, R);
            else filt_extend(agg[t], Fs + k * MAT, Q, ys[k], h, R);
        }
    }
    // phase 2: Kogge-Stone per group of W, then fold across groups
    incl = agg;
    for (long g0 = 0; g0 < nth; g0 += W) {
        const long gn = std::min<long>(W, nth - g0);
        for (int s = 1; s < W; s <<= 1) {
            std::vector<FiltElem<T, D>> nxt(incl.begin() + g0, incl.begin() + g0 + gn);
            for (long l = s; l < gn; ++l) filt_combine(incl[g0 + l - s], incl[g0 + l], nxt[l]);
            std::copy(nxt.begin(), nxt.end(), incl.begin() + g0);
        }
    }
    FiltElem<T, D> gpre;
    filt_identity(gpre);
    for (long g0 = 0; g0 < nth; g0 += W) {
        const long gn = std::min<long>(W, nth - g0);
        for (long l = 0; l < gn; ++l) {
            FiltElem<T, D> loc;
            if (l == 0) filt_identity(loc); else loc = incl[g0 + l - 1];
            filt_combine(gpre, loc, excl[g0 + l]);
        }
        FiltElem<T, D> tmp;
        filt_combine(gpre, incl[g0 + gn - 1], tmp);
        gpre = tmp;
    }
    // phase 3: carry-in + lane-serial Kalman pass, smoothing aggregates
    std::vector<SmthElem<T, D>> sagg(nth), sincl(nth), sexcl(nth);
    LogLik ll;
    for (long t = 0; t < nth; ++t) {
        MeanCov<T, D> s;
        for (int i = 0; i < D; ++i) s.m[i] = T(0);
        for (int i = 0; i < SYM; ++i) s.P[i] = P0[i];
        filt_apply(s, excl[t]);
        smth_identity(sagg[t]);
        MeanCov<T, D> prev = s;
        const long k0 = t * Lc, k1 = std::min<long>(N, (t + 1) * Lc);
        for (long k = k0; k <= k1; ++k) {
            if (k == N) {               // element of the last step of the series
                if (DFORM) { smth_extend_last_u(sagg[t]); break; }
                SmthElem<T, D> e, tmp;
                smth_last(prev, e);
                smth_combine(sagg[t], e, tmp);
                sagg[t] = tmp;
                break;
            }
            T Q[SYM], mp[D], Pp[SYM], FP[MAT];
            load_sym<T, D>(Qs + k * MAT, Q);
            LogLik dummy;
            MeanCov<T, D> cur = prev;
            if (DFORM) {
                // (the halo step k == k1 lends its TRUE observation to the last element of the chunk)
                T u[D], inv, res;
                kf_step_u(cur, Fs + k * MAT, Q, ys[k], h, R, k == 0, k < k1 ? ll : dummy, mp, Pp, FP, u, inv, res);
                if (k > k0) {
                    T E[MAT];
                    smth_gain<T, D>(FP, Pp, E);
                    smth_extend_u(sagg[t], E, u, inv, res);
                }
            } else {
            kf_step(cur, Fs + k * MAT, Q, k < k1 ? ys[k] : T(0), h, R, k == 0, k < k1 ? ll : dummy, mp, Pp, FP);
            if (k > k0) {               // element of step k-1 from this step's predict
                SmthElem<T, D> e, tmp;
                smth_element(prev, mp, Pp, FP, e);
                smth_combine(sagg[t], e, tmp);
                sagg[t] = tmp;
            }
            }
            if (k == k1) break;         // halo step: only its predict was needed
            for (int i = 0; i < D; ++i) fms[k * D + i] = cur.m[i];
            store_sym<T, D>(cur.P, fPs + k * MAT);
            prev = cur;
        }
    }
    *ll_out = ll.value();
    // phase 4: suffix scan of smoothing aggregates
    sincl = sagg;
    for (long g0 = 0; g0 < nth; g0 += W) {
        const long gn = std::min<long>(W, nth - g0);
        for (int s = 1; s < W; s <<= 1) {
            std::vector<SmthElem<T, D>> nxt(sincl.begin() + g0, sincl.begin() + g0 + gn);
            for (long l = 0; l + s < gn; ++l) smth_combine(sincl[g0 + l], sincl[g0 + l + s], nxt[l]);
            std::copy(nxt.begin(), nxt.end(), sincl.begin() + g0);
        }
    }
    SmthElem<T, D> gsuf;
    smth_identity(gsuf);
    const long ngroups = (nth + W - 1) / W;
    for (long g = ngroups - 1; g >= 0; --g) {
        const long g0 = g * W, gn = std::min<long>(W, nth - g0);
        for (long l = 0; l < gn; ++l) {
            SmthElem<T, D> loc;
            if (l == gn - 1) smth_identity(loc); else loc = sincl[g0 + l + 1];
            smth_combine(loc, gsuf, sexcl[g0 + l]);
        }
        SmthElem<T, D> tmp;
        smth_combine(sincl[g0], gsuf, tmp);
        gsuf = tmp;
    }
    // phase 5: RTS pass per chunk
    for (long t = 0; t < nth; ++t) {
        MeanCov<T, D> s;
        for (int i = 0; i < D; ++i) s.m[i] = T(0);
        for (int i = 0; i < SYM; ++i) s.P[i] = T(0);
        smth_apply(sexcl[t], s);        // state at the first step after this chunk
        const long k0 = t * Lc, k1 = std::min<long>(N, (t + 1) * Lc);
        if (DFORM && k1 < N) {          // innovation form: (sm - m, sP - P) of step k1 -- add its filtered moments
            T Pn[SYM];
            load_sym<T, D>(fPs + k1 * MAT, Pn);
            for (int i = 0; i < D; ++i) s.m[i] += fms[k1 * D + i];
            for (int i = 0; i < SYM; ++i) s.P[i] += Pn[i];
        }
        for (long k = k1 - 1; k >= k0; --k) {
            MeanCov<T, D> f;
            for (int i = 0; i < D; ++i) f.m[i] = fms[k * D + i];
            load_sym<T, D>(fPs + k * MAT, f.P);
            if (k == N - 1) {
                s = f;
            } else {
                T Q[SYM], mp[D], Pp[SYM], FP[MAT];
                load_sym<T, D>(Qs + (k + 1) * MAT, Q);
                mat_vec<T, D>(Fs + (k + 1) * MAT, f.m, mp);
                predict_cov<T, D>(Fs + (k + 1) * MAT, f.P, Q, FP, Pp);
                rts_step(f, mp, Pp, FP, s);
            }
            for (int i = 0; i < D; ++i) sms[k * D + i] = s.m[i];
            store_sym<T, D>(s.P, sPs + k * MAT);
        }
    }
    return 0;
}

#define CASE(D_) case D_: return run<T, D_>(N, Lc, W, P0, Fs, Qs, h, R, ys, fms, fPs, sms, sPs, ll);
template <typename T>
static int dispatch(int d, long N, int Lc, int W, const T* P0, const T* Fs, const T* Qs, const T* h, T R,
                    const T* ys, T* fms, T* fPs, T* sms, T* sPs, double* ll) {
    switch (d) {
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6)
        default: return -1;
    }
}

extern "C" int emul_pkfs_f64(int d, long N, int Lc, int W, const double* P0, const double* Fs, const double* Qs,
                             const double* h, double R, const double* ys, double* fms, double* fPs,
                             double* sms, double* sPs, double* ll) {
    return dispatch<double>(d, N, Lc, W, P0, Fs, Qs, h, R, ys, fms, fPs, sms, sPs, ll);
}
extern "C" int emul_pkfs_f32(int d, long N, int Lc, int W, const float* P0, const float* Fs, const float* Qs,
                             const float* h, float R, const float* ys, float* fms, float* fPs,
                             float* sms, float* sPs, double* ll) {
    return dispatch<float>(d, N, Lc, W, P0, Fs, Qs, h, R, ys, fms, fPs, sms, sPs, ll);
}
// the same scan with the smoothing totals in innovation form
#undef CASE
#define CASE(D_) case D_: return run<T, D_, true>(N, Lc, W, P0, Fs, Qs, h, R, ys, fms, fPs, sms, sPs, ll);
template <typename T>
static int dispatch_dform(int d, long N, int Lc, int W, const T* P0, const T* Fs, const T* Qs, const T* h, T R,
                          const T* ys, T* fms, T* fPs, T* sms, T* sPs, double* ll) {
    switch (d) {
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6)
        default: return -1;
    }
}
extern "C" int emul_pkfs_dform_f64(int d, long N, int Lc, int W, const double* P0, const double* Fs, const double* Qs,
                                   const double* h, double R, const double* ys, double* fms, double* fPs,
                                   double* sms, double* sPs, double* ll) {
    return dispatch_dform<double>(d, N, Lc, W, P0, Fs, Qs, h, R, ys, fms, fPs, sms, sPs, ll);
}
extern "C" int emul_pkfs_dform_f32(int d, long N, int Lc, int W, const float* P0, const float* Fs, const float* Qs,
                                   const float* h, float R, const float* ys, float* fms, float* fPs,
                                   float* sms, float* sPs, double* ll) {
    return dispatch_dform<float>(d, N, Lc, W, P0, Fs, Qs, h, R, ys, fms, fPs, sms, sPs, ll);
}

// ---------------------------------------------------------------------------------------------
// log-likelihood gradient on dual numbers (pgps_dual.h), chunked exactly as k_grad_reduce /
// k_grad_apply do it: chunk aggregates, a serial fold standing in for the block scan, carry-in,
// lane-serial Kalman pass with the dual log-likelihood accumulator.
// model: (1 + np) blocks of [lam | N1 (d*d) | Pinf (d*d) | H (d) | R]
// ---------------------------------------------------------------------------------------------
template <int D>
static int run_grad(long N, int Lc, int np, const double* model, const double* ts, double t0, const double* ys,
                    double* out) {
    constexpr int NP = 3, MAT = D * D, SYM = Dim<D>::SYM;
    using T = Dual<NP>;
    const int stride = 1 + 2 * MAT + D + 1;
    auto get = [&](int off) {
        T x(model[off]);
        for (int p = 0; p < np; ++p) x.d[p] = model[(p + 1) * stride + off];
        return x;
    };
    T lam = get(0), N1[MAT], N2[MAT], Pinf[MAT], h[D], R = get(1 + 2 * MAT + D), P0[SYM];
    for (int i = 0; i < MAT; ++i) { N1[i] = get(1 + i); Pinf[i] = get(1 + MAT + i); }
    for (int i = 0; i < D; ++i) h[i] = get(1 + 2 * MAT + i);
    for (int i = 0; i < D; ++i)
        for (int j = 0; j < D; ++j) {
            T acc(0.0);
            for (int l = 0; l < D; ++l) acc += N1[i * D + l] * N1[l * D + j];
            N2[i * D + j] = T(0.5) * acc;
        }
    for (int i = 0; i < D; ++i)
        for (int j = i; j < D; ++j) P0[symi<D>(i, j)] = T(0.5) * (Pinf[i * D + j] + Pinf[j * D + i]);
    const long nth = (N + Lc - 1) / Lc;
    std::vector<FiltElem<T, D>> agg(nth);
    for (long t = 0; t < nth; ++t) {
        filt_identity(agg[t]);
        for (long k = t * Lc; k < std::min<long>(N, (t + 1) * Lc); ++k) {
            if (k == 0) {
                filt_first(agg[t], P0, T(ys[0]), h, R);
            } else {
                T F[MAT], Q[SYM];
                lti_step_dual<NP, D>(lam, N1, N2, Pinf, ts[k] - ts[k - 1], F, Q);
                filt_extend(agg[t], F, Q, T(ys[k]), h, R);
            }
        }
    }
    FiltElem<T, D> pre;
    filt_identity(pre);
    T total(0.0);
    for (long t = 0; t < nth; ++t) {
        MeanCov<T, D> s;
        for (int i = 0; i < D; ++i) s.m[i] = T(0.0);
        for (int i = 0; i < SYM; ++i) s.P[i] = P0[i];
        filt_apply(s, pre);
        LogLikDual<NP> ll;
        for (long k = t * Lc; k < std::min<long>(N, (t + 1) * Lc); ++k) {
            T F[MAT], Q[SYM], mp[D], Pp[SYM], FP[MAT];
            lti_step_dual<NP, D>(lam, N1, N2, Pinf, ts[k] - (k ? ts[k - 1] : t0), F, Q);
            kf_step(s, F, Q, T(ys[k]), h, R, k == 0, ll, mp, Pp, FP);
        }
        total += ll.value();
        FiltElem<T, D> nxt;
        filt_combine(pre, agg[t], nxt);
        pre = nxt;
    }
    out[0] = total.v;
    for (int p = 0; p < np; ++p) out[1 + p] = total.d[p];
    return 0;
}

extern "C" int emul_ll_grad(int d, long N, int Lc, int np, const double* model, const double* ts, double t0,
                            const double* ys, double* out) {
    if (np < 1 || np > 3) return -1;
    if (d == 1) return run_grad<1>(N, Lc, np, model, ts, t0, ys, out);
    if (d == 2) return run_grad<2>(N, Lc, np, model, ts, t0, ys, out);
    if (d == 3) return run_grad<3>(N, Lc, np, model, ts, t0, ys, out);
    return -1;
}
