// pgps_grad.hip -- host side of the log-likelihood gradient (pgps_gp_ll_grad_*): see pgps_grad.hip.h.
#include "pgps_grad.hip.h"

namespace pgps {

constexpr int kNP = 3;          // hyper-parameters differentiated at once (variance, lengthscale, noise)

template <int D>
static int launch_grad_d(pgps_ctx* ctx, GradModel<kNP> m) {
    using T = Dual<kNP>;
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
    timed_launch(ctx, PGPS_K_FILTER_REDUCE, k_grad_reduce<kNP, D>, grid, block, 0, m);
    timed_launch(ctx, PGPS_K_FILTER_APPLY, k_grad_apply<kNP, D>, grid, block, 0, m);
    timed_launch(ctx, PGPS_K_LL_FINALIZE, k_grad_finalize<kNP>, dim3(1), block, 0, (const T*)m.llpart, m.nblocks, m.out);
    HIPCHK(ctx, hipGetLastError());
    return PGPS_OK;
}

// model: (1 + kNP) consecutive blocks, block 0 = values, block p = d/dtheta_p, each
//   [lam | N1 (d*d) | Pinf (d*d) | H (d) | R]
int launch_grad(pgps_ctx* ctx, long N, int d, int np, const double* model, const double* ts, double t0,
                const double* ys, double* out_dev) {
    if (np < 1 || np > kNP) return PGPS_E_INVALID;
    if (d < 1 || d > 2) return PGPS_E_UNSUPPORTED_DIM;
    GradModel<kNP> m{};
    const int stride = 1 + 2 * d * d + d + 1;
    auto get = [&](int off) {
        Dual<kNP> x(model[off]);
        for (int p = 0; p < np; ++p) x.d[p] = model[(p + 1) * stride + off];
        return x;
    };
    m.lam = get(0);
    for (int i = 0; i < 4; ++i) { m.N1[i] = Dual<kNP>(0.0); m.Pinf[i] = Dual<kNP>(0.0); }
    for (int i = 0; i < 2; ++i) m.H[i] = Dual<kNP>(0.0);
    for (int i = 0; i < d * d; ++i) { m.N1[i] = get(1 + i); m.Pinf[i] = get(1 + d * d + i); }
    for (int i = 0; i < d; ++i) m.H[i] = get(1 + 2 * d * d + i);
    m.R = get(1 + 2 * d * d + d);
    m.ts = ts; m.ys = ys; m.t_prev = t0; m.N = N; m.out = out_dev;
    return d == 1 ? launch_grad_d<1>(ctx, m) : launch_grad_d<2>(ctx, m);
}

}  // namespace pgps
