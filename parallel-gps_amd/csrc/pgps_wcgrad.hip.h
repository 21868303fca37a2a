// pgps_wcgrad.hip.h -- the two level-1 kernels of the adjoint log-likelihood gradient (pgps_gradlti.h) in the
// wave-cooperative family: one wavefront per chunk, operands in the wave's LDS pool padded to DP, register tiles on the
// lane grid (pgps_wc.hip, which includes this file after its primitives).  Any state dimension up to 32; the automatic
// choice for d = 17..32 (the reference's CO2 kernel, pssgp/experiments/co2/mcmc.py:42-65, has d = 18), the cross-check of
// the row-cooperative kernels below that.  fp64.
//
// The states entering the chunks come from the scan's own entry kernels: wc_enter1 leaves the filtered (m, P) entering
// chunk c in enter1[c]; with the ADJOINT totals in the smoothing scan, wc_senter1 leaves (a, W) behind chunk c in
// senter1[c] (the scan's terminal state is zero: nothing lies behind the last step).
#pragma once

#include "pgps_gradlti.h"

namespace pgps {
namespace wc {

template <int DP>
constexpr size_t wg_lds_doubles() { return (size_t)9 * Geo<DP>::MSZ + 2 * ((Geo<DP>::NSL + 1) & ~1) + 16 * DP + 64; }

// forward: Kalman pass of the chunk (filtered moments stored), adjoint elements folded into the chunk's total
//   E = F^T - v K^T,  g = v r / s,  L = -v v^T / (2 s);   total <- total (x) element under the smoothing operator
template <int DP>
__global__ __launch_bounds__(64) void wg_apply1(const WcArgs<double> a) {
    using T = double;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MSZ = Geo<DP>::MSZ, NSL = Geo<DP>::NSL, TS = Geo<DP>::TS, LD = Geo<DP>::LD;
    const int d = a.d, dd = d * d, ns = nsmth(d), dk = Geo<DP>::dk(d);
    Pool<T> pool(reinterpret_cast<T*>(smem));
    T* m = pool.take(DP); T* P = pool.take(MSZ); T* F = pool.take(MSZ); T* FP = pool.take(MSZ); T* Pp = pool.take(MSZ);
    T* X = pool.take(MSZ); T* sacc = pool.take(NSL);
    T* h = pool.take(DP); T* mp = pool.take(DP); T* u = pool.take(DP); T* v = pool.take(DP); T* w = pool.take(DP);
    T* Kv = pool.take(DP);
    const long c = blockIdx.x;
    if (c >= a.nchunk) return;
    const long k0 = c * a.Lw, k1 = min(a.N, k0 + a.Lw);
    const int lane = lane_id();
    const int r0 = lrow<DP>() * TS, c0 = lcol<DP>() * TS;
    const bool act = lactive<DP>();
    vec_g2l<T, DP>(d, a.H, h);
    {
        const T* cg = a.enter1 + c * (d + dd);
        vec_g2l<T, DP>(d, cg, m);
        mat_g2l<T, DP>(d, cg + d, P);
    }
    smth_set_identity<T, DP>(d, sacc);
    Smth<T, DP> s(sacc);
    StepTiles<T, DP> st;
    st.fetch(d, a.Fs + k0 * dd, a.Qs + k0 * dd);
    sync();
    double quad = 0.0, mant = 1.0;
    long long expo = 0, count = 0;
    for (long k = k0; k < k1; ++k) {
        st.park_f(F);
        Tile<T, DP> pp;
#pragma unroll
        for (int ti = 0; ti < TS; ++ti)
#pragma unroll
            for (int tj = 0; tj < TS; ++tj) pp.v[ti][tj] = st.q[ti][tj];
        if (k + 1 < k1) st.fetch(d, a.Fs + (k + 1) * dd, a.Qs + (k + 1) * dd);
        sync();
        // predict
        Tile<T, DP> fp;
        fp.zero();
        mv<T, DP, false>(dk, F, m, mp);
        mv<T, DP, true>(dk, F, h, v);                   // v = (H F)^T
        mm_acc<T, DP, 1>(dk, F, P, fp);                 // P symmetric: read as P^T
        fp.st(FP);
        sync();
        mm_acc<T, DP, 1>(dk, FP, F, pp);
        pp.st(Pp);
        sync();
        {
            Tile<T, DP> t;
            t.ld_t(Pp);
            pp.average(t);
        }
        sync();
        pp.st(Pp);
        sync();
        const T y = a.ys[k];
        const bool obs = !(y != y);
        mv<T, DP, false>(dk, Pp, h, u);
        mv<T, DP, false>(dk, s.E, v, w);                // w = Ec v
        sync();
        const T S = dot<T, DP>(h, u) + a.R;
        const T mu = dot<T, DP>(h, mp);
        if (obs) {
            const double r = double(y) - double(mu);
            quad += r * r / double(S);
            int ex;
            mant = frexp(mant * double(S), &ex);
            expo += ex;
            count += 1;
        }
        const T inv = obs ? T(1) / S : T(0);
        const T res = obs ? y - mu : T(0);
        if (lane < DP) Kv[lane] = u[lane] * inv;
        sync();
        // fold the adjoint element: E' = Ec F^T - w K^T, g' = g + w r / s, L' = L - w w^T / (2 s)
        {
            Tile<T, DP> e2, lt;
            e2.zero();
            mm_acc<T, DP, 1>(dk, s.E, F, e2);
            lt.ld(s.L);
            if (act) {
#pragma unroll
                for (int ti = 0; ti < TS; ++ti)
#pragma unroll
                    for (int tj = 0; tj < TS; ++tj) {
                        e2.v[ti][tj] -= w[r0 + ti] * Kv[c0 + tj];
                        lt.v[ti][tj] -= T(0.5) * inv * w[r0 + ti] * w[c0 + tj];
                    }
            }
            sync();                                     // every lane has read Ec
            e2.st(s.E);
            lt.st(s.L);
            if (lane < DP) s.g[lane] += w[lane] * (res * inv);
        }
        // update
        {
            Tile<T, DP> pt = pp;
            if (act) {
#pragma unroll
                for (int ti = 0; ti < TS; ++ti)
#pragma unroll
                    for (int tj = 0; tj < TS; ++tj) pt.v[ti][tj] -= u[r0 + ti] * u[c0 + tj] * inv;
            }
            pt.st(P);
            if (lane < DP) m[lane] = mp[lane] + u[lane] * (res * inv);
            sync();
            vec_l2g<T, DP>(d, m, a.fms + k * d);
            pt.st_g(d, a.fPs + k * dd);
        }
    }
    (void)X; (void)LD;
    sync();
    smth_l2g<T, DP>(d, sacc, a.sagg1 + c * ns);
    if (lane == 0) {
        const double logdet = log(mant) + double(expo) * 0.6931471805599453;
        a.llpart[c] = -0.5 * (double(count) * 1.8378770664093453 + logdet + quad);
    }
}

// backward: from (a, W) behind the chunk down its steps, the predict of every step recomputed from the stored filtered
// moments of the step before; chunk partials of the model's adjoints (see rc_gback1 / oracle/np_grad.py)
template <int DP>
__global__ __launch_bounds__(64) void wg_back1(const WcArgs<double> a, const GradLtiArgs g) {
    using T = double;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MSZ = Geo<DP>::MSZ, TS = Geo<DP>::TS;
    const int d = a.d, dd = d * d, dk = Geo<DP>::dk(d);
    Pool<T> pool(reinterpret_cast<T*>(smem));
    T* av = pool.take(DP); T* W = pool.take(MSZ); T* F = pool.take(MSZ); T* P = pool.take(MSZ); T* FP = pool.take(MSZ);
    T* Pp = pool.take(MSZ); T* B = pool.take(MSZ); T* Ppb = pool.take(MSZ); T* X = pool.take(MSZ); T* Tm = pool.take(MSZ);
    T* h = pool.take(DP); T* m = pool.take(DP); T* mp = pool.take(DP); T* u = pool.take(DP); T* Kv = pool.take(DP);
    T* BK = pool.take(DP); T* ub = pool.take(DP); T* mpb = pool.take(DP); T* pu = pool.take(DP); T* an = pool.take(DP);
    const long c = blockIdx.x;
    if (c >= a.nchunk) return;
    const long k0 = c * a.Lw, k1 = min(a.N, k0 + a.Lw);
    const int lane = lane_id();
    const int r0 = lrow<DP>() * TS, c0 = lcol<DP>() * TS;
    const bool act = lactive<DP>();
    vec_g2l<T, DP>(d, a.H, h);
    {
        const T* cg = a.senter1 + c * (d + dd);
        vec_g2l<T, DP>(d, cg, av);
        mat_g2l<T, DP>(d, cg + d, W);
    }
    Tile<T, DP> pinf, abar;
    pinf.ld_g(d, a.P0);
    {
        Tile<T, DP> t;                                  // symmetric part of the stationary covariance
        pinf.st(X);
        sync();
        t.ld_t(X);
        pinf.average(t);
        sync();
    }
    abar.zero();
    T Ub = T(0), Hb = T(0), Rb = T(0);
    StepTiles<T, DP> st;
    st.fetch(d, a.Fs + (k1 - 1) * dd, a.Qs + (k1 - 1) * dd);
    sync();
    for (long k = k1 - 1; k >= k0; --k) {
        st.park_f(F);
        Tile<T, DP> pp;
#pragma unroll
        for (int ti = 0; ti < TS; ++ti)
#pragma unroll
            for (int tj = 0; tj < TS; ++tj) pp.v[ti][tj] = st.q[ti][tj];
        if (k > k0) st.fetch(d, a.Fs + (k - 1) * dd, a.Qs + (k - 1) * dd);
        // the filtered moments of the step before (the prior before the first step)
        if (k > 0) {
            Tile<T, DP> pt;
            pt.ld_g(d, a.fPs + (k - 1) * dd);
            pt.st(P);
            vec_g2l<T, DP>(d, a.fms + (k - 1) * d, m);
        } else {
            pinf.st(P);
            if (lane < DP) m[lane] = T(0);
        }
        const T dt = g.ts[k] - (k > 0 ? g.ts[k - 1] : g.t0);
        sync();
        // predict
        Tile<T, DP> fp;
        fp.zero();
        mv<T, DP, false>(dk, F, m, mp);
        mm_acc<T, DP, 1>(dk, F, P, fp);
        fp.st(FP);
        sync();
        mm_acc<T, DP, 1>(dk, FP, F, pp);
        pp.st(Pp);
        sync();
        {
            Tile<T, DP> t;
            t.ld_t(Pp);
            pp.average(t);
        }
        sync();
        pp.st(Pp);
        // B = W + a a^T / 2
        Tile<T, DP> bt;
        bt.ld(W);
        if (act) {
#pragma unroll
            for (int ti = 0; ti < TS; ++ti)
#pragma unroll
                for (int tj = 0; tj < TS; ++tj) bt.v[ti][tj] += T(0.5) * av[r0 + ti] * av[c0 + tj];
        }
        bt.st(B);
        sync();
        const T y = a.ys[k];
        const bool obs = !(y != y);
        mv<T, DP, false>(dk, Pp, h, u);
        sync();
        const T S = dot<T, DP>(h, u) + a.R;
        const T mu = dot<T, DP>(h, mp);
        const T inv = obs ? T(1) / S : T(0);
        const T res = obs ? y - mu : T(0);
        if (lane < DP) Kv[lane] = u[lane] * inv;
        sync();
        mv<T, DP, false>(dk, B, Kv, BK);
        sync();
        const T kap = dot<T, DP>(Kv, BK), aK = dot<T, DP>(av, Kv);
        const T ri = res * inv;
        const T sbar = obs ? (-aK * ri + kap - T(0.5) * inv + T(0.5) * ri * ri) : T(0);
        const T rbar = aK - ri;
        if (lane < DP) {
            ub[lane] = av[lane] * ri - T(2) * BK[lane] + sbar * h[lane];
            mpb[lane] = av[lane] - rbar * h[lane];
        }
        sync();
        // Ppbar = B + sym(ubar H);  X = 2 dt (Pp - Pinf)
        {
            Tile<T, DP> xt;
            if (act) {
#pragma unroll
                for (int ti = 0; ti < TS; ++ti)
#pragma unroll
                    for (int tj = 0; tj < TS; ++tj) {
                        bt.v[ti][tj] += T(0.5) * (ub[r0 + ti] * h[c0 + tj] + h[r0 + ti] * ub[c0 + tj]);
                        xt.v[ti][tj] = (T(2) * dt) * (pp.v[ti][tj] - pinf.v[ti][tj]);
                    }
            } else {
                xt.zero();
            }
            bt.st(Ppb);
            xt.st(X);
        }
        mv<T, DP, false>(dk, Pp, ub, pu);               // Pp ubar
        mv<T, DP, true>(dk, F, mpb, an);                // a' = F^T mpbar
        sync();
        mm_acc<T, DP, 0>(dk, Ppb, X, abar);
        if (act) {
#pragma unroll
            for (int ti = 0; ti < TS; ++ti)
#pragma unroll
                for (int tj = 0; tj < TS; ++tj) abar.v[ti][tj] += dt * mpb[r0 + ti] * mp[c0 + tj];
        }
        if (lane < DP) {
            Ub += ub[lane];
            Hb += sbar * u[lane] + pu[lane] - rbar * mp[lane];
        }
        Rb += sbar;
        // W' = F^T Ppbar F - a' a'^T / 2
        {
            Tile<T, DP> t;
            t.zero();
            mm_acc<T, DP, 0>(dk, Ppb, F, t);
            t.st(Tm);
            sync();
            Tile<T, DP> wn;
            wn.zero();
            mm_acc<T, DP, 2>(dk, F, Tm, wn);
            wn.st(W);
            sync();
            Tile<T, DP> wt;
            wt.ld_t(W);
            wn.average(wt);
            if (act) {
#pragma unroll
                for (int ti = 0; ti < TS; ++ti)
#pragma unroll
                    for (int tj = 0; tj < TS; ++tj) wn.v[ti][tj] -= T(0.5) * an[r0 + ti] * an[c0 + tj];
            }
            sync();
            wn.st(W);
            if (lane < DP) av[lane] = an[lane];
            sync();
        }
    }
    double* rec = g.gpart + c * (long)grad_lti_nstat(d);
    abar.st_g(d, rec);
    if (lane < d) {
        rec[dd + lane] = Ub;
        rec[dd + d + lane] = Hb;
    }
    if (lane == 0) rec[dd + 2 * d] = Rb;
}

}  // namespace wc
}  // namespace pgps
