// pgps_grad.hip -- host side of the log-likelihood gradient (pgps_gp_ll_grad_*): see pgps_grad.hip.h.
#include "pgps_grad.hip.h"

namespace pgps {

constexpr int kNP = 3;          // hyper-parameters differentiated per pass for d <= 2 (d = 3: one per pass)

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

// model: (1 + np) consecutive blocks, block 0 = values, block p = d/dtheta_p, each
//   [lam | N1 (d*d) | Pinf (d*d) | H (d) | R]
// out_dev: 1 + np doubles (d <= 2) -- for d = 3, 2 * np doubles of scratch follow: pass p writes
// (ll, d ll / d theta_p) at out_dev[2 p] and a last tiny kernel compacts them.
int launch_grad(pgps_ctx* ctx, long N, int d, int np, const double* model, const double* ts, double t0,
                const double* ys, double* out_dev) {
    if (np < 1 || np > kNP) return PGPS_E_INVALID;
    if (d < 1 || d > 3) return PGPS_E_UNSUPPORTED_DIM;
    if (d == 1) return launch_grad_d<kNP, 1>(ctx, make_model<kNP>(d, np, 0, model, N, ts, t0, ys, out_dev));
    if (d == 2) return launch_grad_d<kNP, 2>(ctx, make_model<kNP>(d, np, 0, model, N, ts, t0, ys, out_dev));
    // d = 3: one direction per model, the np models side by side in the same three launches (grid.y): at the
    // reference's series lengths one workgroup's scan tree on duals is the whole cost (64 us), not its length
    using T = Dual<1>;
    double* scratch = out_dev + 1 + np;
    GradPack<1> pack{};
    HIPCHK(ctx, hipSetDevice(ctx->device));
    for (int p = 0; p < 3; ++p) pack.m[p] = make_model<1>(d, np, p < np ? p : 0, model, N, ts, t0, ys, scratch + 2 * (p < np ? p : 0));
    GradModel<1>& m0 = pack.m[0];
    geometry(ctx, m0.N, &m0.Lc, &m0.nblocks);
    m0.nlanes = (long)m0.nblocks * kBlock;
    const size_t nb = (size_t)m0.nblocks, nl = (size_t)m0.nlanes;
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t s_spine = up(nb * Dim<3>::NFILT * sizeof(T)), s_lpre = up(nl * Dim<3>::NFILT * sizeof(T)), s_ll = up(nb * sizeof(T));
    const size_t per = s_spine + s_lpre + s_ll;
    int rc = ensure(ctx, ctx->ws, per * (size_t)np);
    if (rc) return rc;
    char* base = (char*)ctx->ws.p;
    for (int p = 0; p < np; ++p) {
        GradModel<1>& m = pack.m[p];
        m.Lc = m0.Lc; m.nblocks = m0.nblocks; m.nlanes = m0.nlanes;
        m.spine = (T*)(base + per * p);
        m.lpre = (T*)(base + per * p + s_spine);
        m.llpart = (T*)(base + per * p + s_spine + s_lpre);
    }
    const dim3 grid(m0.nblocks, np), block(kBlock);
    timed_launch(ctx, PGPS_K_FILTER_REDUCE, k_grad_reduce_pack<1, 3>, grid, block, 0, pack);
    timed_launch(ctx, PGPS_K_FILTER_APPLY, k_grad_apply_pack<1, 3>, grid, block, 0, pack);
    timed_launch(ctx, PGPS_K_LL_FINALIZE, k_grad_finalize_pack<1>, dim3(1, np), block, 0, pack);
    k_grad_compact<<<1, 64, 0, ctx->stream>>>(scratch, np, out_dev);
    HIPCHK(ctx, hipGetLastError());
    return PGPS_OK;
}

}  // namespace pgps
