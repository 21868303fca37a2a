// pgps_grad.hip -- host side of the log-likelihood gradient (pgps_gp_ll_grad_*): see pgps_grad.hip.h.
#include "pgps_grad.hip.h"

namespace pgps {

constexpr int kNP = 3;          // hyper-parameters differentiated per pass for d <= 2 (d = 3: one per pass)
constexpr long kGradOneLaunch = 2048;   // series up to this length: one workgroup per direction, one launch

template <int NP, int D>
static int launch_grad_d(pgps_ctx* ctx, GradModel<NP> m) {
    using T = Dual<NP>;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    geometry(ctx, m.N, &m.Lc, &m.nblocks);
    m.nlanes = (long)m.nblocks * kBlock;
    const size_t nb = (size_t)m.nblocks, nl = (size_t)m.nlanes;
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    size_t off = 0;
    const size_t o_spine = off; off = up(off + nb * Dim<D>::NFILT * sizeof(T));
    const size_t o_lpre = off;  off = up(off + nl * Dim<D>::NFILT * sizeof(T));
    const size_t o_ll = off;    off = up(off + nb * sizeof(T));
    int rc = ensure(ctx, ctx->ws, off);
    if (rc) return rc;
    char* base = (char*)ctx->ws.p;
    m.spine = (T*)(base + o_spine);
    m.lpre = (T*)(base + o_lpre);
    m.llpart = (T*)(base + o_ll);
    const dim3 grid(m.nblocks), block(kBlock);
    timed_launch(ctx, PGPS_K_FILTER_REDUCE, k_grad_reduce<NP, D>, grid, block, 0, m);
    timed_launch(ctx, PGPS_K_FILTER_APPLY, k_grad_apply<NP, D>, grid, block, 0, m);
    timed_launch(ctx, PGPS_K_LL_FINALIZE, k_grad_finalize<NP>, dim3(1), block, 0, (const T*)m.llpart, m.nblocks, m.out);
    HIPCHK(ctx, hipGetLastError());
    return PGPS_OK;
}

// model blocks -> duals with derivative directions [p0, p0 + NP)
template <int NP>
static GradModel<NP> make_model(int d, int np, int p0, const double* model, long N, const double* ts, double t0,
                                const double* ys, double* out) {
    using T = Dual<NP>;
    GradModel<NP> m{};
    const int stride = 1 + 2 * d * d + d + 1;
    auto get = [&](int off) {
        T x(model[off]);
        for (int p = 0; p < NP && p0 + p < np; ++p) x.d[p] = model[(p0 + p + 1) * stride + off];
        return x;
    };
    for (int i = 0; i < 9; ++i) { m.N1[i] = T(0.0); m.N2[i] = T(0.0); m.Pinf[i] = T(0.0); }
    for (int i = 0; i < 3; ++i) m.H[i] = T(0.0);
    m.lam = get(0);
    for (int i = 0; i < d * d; ++i) { m.N1[i] = get(1 + i); m.Pinf[i] = get(1 + d * d + i); }
    for (int i = 0; i < d; ++i) m.H[i] = get(1 + 2 * d * d + i);
    m.R = get(1 + 2 * d * d + d);
    for (int i = 0; i < d; ++i)                 // N2 = N1 N1 / 2 (product rule through the dual arithmetic)
        for (int j = 0; j < d; ++j) {
            T acc(0.0);
            for (int l = 0; l < d; ++l) acc += m.N1[i * d + l] * m.N1[l * d + j];
            m.N2[i * d + j] = T(0.5) * acc;
        }
    m.ts = ts; m.ys = ys; m.t_prev = t0; m.N = N; m.out = out;
    return m;
}

// One direction per model, the np models side by side in the same three launches (grid.y): where a series is short, one
// workgroup's scan tree on duals is the whole cost -- its latency, not its length -- and a Dual<1> tree is less than half
// of a Dual<3> one.  Pass p writes (ll, d ll / d theta_p) into context scratch; a last tiny kernel compacts them.
template <int D>
static int launch_grad_pack(pgps_ctx* ctx, long N, int np, const double* model, const double* ts, double t0, const double* ys,
                            double* out_dev) {
    using T = Dual<1>;
    GradPack<1> pack{};
    HIPCHK(ctx, hipSetDevice(ctx->device));
    long nlanes = 0;
    int Lc = 0, nblocks = 0;
    geometry(ctx, N, &Lc, &nblocks);
    // up to kGradOneLaunch steps: one workgroup per direction, everything in one launch (k_grad_one_pack)
    if (ctx->chunk <= 0 && N <= kGradOneLaunch && nblocks > 1) { Lc = (int)((N + kBlock - 1) / kBlock); nblocks = 1; }
    nlanes = (long)nblocks * kBlock;
    const size_t nb = (size_t)nblocks, nl = (size_t)nlanes;
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t s_spine = up(nb * Dim<D>::NFILT * sizeof(T)), s_lpre = up(nl * Dim<D>::NFILT * sizeof(T)), s_ll = up(nb * sizeof(T));
    const size_t per = s_spine + s_lpre + s_ll;
    int rc = ensure(ctx, ctx->ws, per * (size_t)np + 256);
    if (rc) return rc;
    char* base = (char*)ctx->ws.p;
    double* scratch = (double*)(base + per * (size_t)np);
    for (int p = 0; p < 3; ++p) pack.m[p] = make_model<1>(D, np, p < np ? p : 0, model, N, ts, t0, ys, scratch + 2 * (p < np ? p : 0));
    for (int p = 0; p < np; ++p) {
        GradModel<1>& m = pack.m[p];
        m.Lc = Lc; m.nblocks = nblocks; m.nlanes = nlanes;
        m.spine = (T*)(base + per * p);
        m.lpre = (T*)(base + per * p + s_spine);
        m.llpart = (T*)(base + per * p + s_spine + s_lpre);
    }
    const dim3 grid(nblocks, np), block(kBlock);
    if (nblocks == 1) {
        timed_launch(ctx, PGPS_K_FILTER_APPLY, k_grad_one_pack<D>, grid, block, 0, pack, out_dev);
        HIPCHK(ctx, hipGetLastError());
        return PGPS_OK;
    }
    timed_launch(ctx, PGPS_K_FILTER_REDUCE, k_grad_reduce_pack<1, D>, grid, block, 0, pack);
    timed_launch(ctx, PGPS_K_FILTER_APPLY, k_grad_apply_pack<1, D>, grid, block, 0, pack);
    timed_launch(ctx, PGPS_K_LL_FINALIZE, k_grad_finalize_pack<1>, dim3(1, np), block, 0, pack);
    k_grad_compact<<<1, 64, 0, ctx->stream>>>(scratch, np, out_dev);
    HIPCHK(ctx, hipGetLastError());
    return PGPS_OK;
}

// Series up to this many steps take one direction per model at d <= 2 as well (PGPS_GRAD_PACK_MAX; d = 3 always does).
// Matern-3/2, us per call, all directions in one dual / one per model: 2^12 152 / 80, 2^15 158 / 75, 2^17 146 / 104,
// 2^18 165 / 119, 2^20 207 / 211 (Matern-1/2 at 2^20: 80 / 100) -- tools/grad_pack_probe.py
#ifndef PGPS_GRAD_PACK_MAX
#define PGPS_GRAD_PACK_MAX (1L << 18)
#endif

// model: (1 + np) consecutive blocks, block 0 = values, block p = d/dtheta_p, each
//   [lam | N1 (d*d) | Pinf (d*d) | H (d) | R]
// out_dev: 1 + np doubles
int launch_grad(pgps_ctx* ctx, long N, int d, int np, const double* model, const double* ts, double t0,
                const double* ys, double* out_dev) {
    if (np < 1 || np > kNP) return PGPS_E_INVALID;
    if (d < 1 || d > 3) return PGPS_E_UNSUPPORTED_DIM;
    const long pack_max = ctx->grad_pack >= 0 ? ctx->grad_pack : PGPS_GRAD_PACK_MAX;
    if (d == 3) return launch_grad_pack<3>(ctx, N, np, model, ts, t0, ys, out_dev);
    if (N <= pack_max) {
        if (d == 1) return launch_grad_pack<1>(ctx, N, np, model, ts, t0, ys, out_dev);
        return launch_grad_pack<2>(ctx, N, np, model, ts, t0, ys, out_dev);
    }
    if (d == 1) return launch_grad_d<kNP, 1>(ctx, make_model<kNP>(d, np, 0, model, N, ts, t0, ys, out_dev));
    return launch_grad_d<kNP, 2>(ctx, make_model<kNP>(d, np, 0, model, N, ts, t0, ys, out_dev));
}

}  // namespace pgps
