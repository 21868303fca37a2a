// pgps_res_inst.hip -- the resident filter + smoother launch (pgps_resident.hip.h) instantiated for one (dtype, d):
// array form (Fs, Qs given: pgps_pkfs_dev_*) and fused form (Matern model + time stamps: pgps_gp_dev_*).
#include "pgps_resident.hip.h"

#ifndef PGPS_RES_T
#error "compile with -DPGPS_RES_T=<float|double> -DPGPS_RES_D=<d>"
#endif

namespace pgps {

template <typename T, int D>
int launch_resident(pgps_ctx* ctx, ResArgs<T> ra, bool fused, bool smooth) {
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ScanArgs<T>& a = ra.s;
    // steps per lane: 16 (4096 per workgroup) -- or 8 where the whole series then still fits the chip: twice the workgroups
    // for series of up to 2048 steps per CU, which at 16 would leave half of the CUs idle (2^19 steps: 39 -> 30 us)
    int lc = kResLc;
    if (ctx->resident > 0 && (ctx->chunk == 8 || ctx->chunk == 16)) lc = ctx->chunk;
    else if (a.N <= (long)kBlock * 8 * ctx->n_cu) lc = 8;
    a.Lc = lc;
    a.nblocks = (int)((a.N + (long)kBlock * lc - 1) / ((long)kBlock * lc));
    if (a.nblocks < 1 || a.nblocks > ctx->n_cu) return PGPS_E_INVALID;      // every workgroup must be resident
    a.nlanes = (long)a.nblocks * kBlock;
    a.seg_first = 1;
    a.seg_last = 1;
    a.shortcut = ctx->shortcut != 0 ? 1 : 0;        // (a workgroup spans 2048 or 4096 steps; the test is on the data either way)
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t nb = (size_t)a.nblocks;
    size_t off = 0;
    const size_t o_spine = off;  off = up(off + nb * Dim<D>::NFILT * sizeof(T));
    const size_t o_sspine = off; off = up(off + nb * Dim<D>::NSMTH * sizeof(T));
    const size_t o_ll = off;     off = up(off + nb * sizeof(double));
    int rc = ensure(ctx, ctx->ws, off);
    if (rc) return rc;
    ctx->ws_epoch++;
    char* base = (char*)ctx->ws.p;
    a.spine = (T*)(base + o_spine);
    a.sspine = (T*)(base + o_sspine);
    a.llpart = (double*)(base + o_ll);
    a.status = ctx->status_word;
    const unsigned e = ctx->res_epoch++;
    ra.bar = ctx->status_word + kResBarWord + (e & 1u) * 256;
    ra.bar_next = ctx->status_word + kResBarWord + ((e + 1u) & 1u) * 256;
    ra.flags1 = ctx->status_word + kResFlagWord;
    ra.flags2 = ctx->status_word + kResFlagWord + 256;
    ra.epoch = (int)(e % 0x7ffffffeu) + 1;           // compared for equality: a stale flag of any earlier launch never matches
    static_assert(kResFlagWord * 4 + 2 * 256 * 4 <= kStatusBytes, "hand-off flags outside the status buffer");
    ra.stamps = nullptr;
    if (ctx->resident == 2) {
        rc = ensure(ctx, ctx->res_stamps, nb * 16 * sizeof(long long));
        if (rc) return rc;
        ra.stamps = (long long*)ctx->res_stamps.p;
        ctx->res_stamp_blocks = a.nblocks;
    }
    const dim3 grid(a.nblocks), block(kBlock);
    auto go = [&](auto lcv, auto fusedv, auto smoothv) {
        timed_launch(ctx, PGPS_K_RESIDENT, k_pkfs_resident<T, D, decltype(lcv)::value, decltype(fusedv)::value, decltype(smoothv)::value>,
                     grid, block, 0, ra);
    };
    auto pick = [&](auto lcv) {
        if (fused) { if (smooth) go(lcv, std::true_type{}, std::true_type{}); else go(lcv, std::true_type{}, std::false_type{}); }
        else { if (smooth) go(lcv, std::false_type{}, std::true_type{}); else go(lcv, std::false_type{}, std::false_type{}); }
    };
    if (lc == 8) pick(std::integral_constant<int, 8>{}); else pick(std::integral_constant<int, kResLc>{});
    HIPCHK(ctx, hipGetLastError());
    return PGPS_OK;
}

template int launch_resident<PGPS_RES_T, PGPS_RES_D>(pgps_ctx*, ResArgs<PGPS_RES_T>, bool, bool);

}  // namespace pgps
