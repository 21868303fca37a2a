// pgps_rc2_inst.hip -- one translation unit per state dimension (their own padding: none from d = 18 on) of the two-rows level-1 kernels
// (-DPGPS_RC2_DP=18..32): both scalar types, launched from pgps_wc.hip through launch_rc2_level1.
#include "pgps_internal.h"
#include "pgps_rc2.hip.h"

#ifndef PGPS_RC2_DP
#error "compile with -DPGPS_RC2_DP=<18..32>"
#endif

namespace pgps {

#define PGPS_RC2_CAT2(a, b) a##b
#define PGPS_RC2_CAT(a, b) PGPS_RC2_CAT2(a, b)
#define PGPS_RC2_LAUNCH PGPS_RC2_CAT(launch_rc2_, PGPS_RC2_DP)

// which: 0 reduce1, 1 apply1 (filter only), 2 apply1 (with the smoothing total), 3 smooth1, 4 / 5 one level of the filter / smoother scan
template <typename T>
static int launch_one(pgps_ctx* ctx, int which, const wc::WcArgs<T>& a) {
    constexpr int DP = PGPS_RC2_DP;
    const dim3 blk(64), grid((unsigned)((a.nchunk + 1) / 2));
    const bool full = a.d == DP;        // no padding: the loads and stores without their column conditions
    // (the padded flavour exists for d = 17 on 18 alone: every other dimension has a unit of its own)
    constexpr bool PADDED = DP == 18;
    if (!full && !PADDED) return PGPS_E_UNSUPPORTED_DIM;
#define PGPS_RC2_GO(SLOT, ...)                                                                  \
    do {                                                                                        \
        if (full) timed_launch(ctx, SLOT, rc2::__VA_ARGS__, true>, grid, blk, 0u, a);           \
        else if constexpr (PADDED) timed_launch(ctx, SLOT, rc2::__VA_ARGS__, false>, grid, blk, 0u, a); \
    } while (0)
    switch (which) {
        case 0: PGPS_RC2_GO(PGPS_K_FILTER_REDUCE, rc2_reduce1<T, DP); break;
        case 1: PGPS_RC2_GO(PGPS_K_FILTER_APPLY, rc2_apply1<T, DP, false); break;
        case 2: PGPS_RC2_GO(PGPS_K_FILTER_APPLY, rc2_apply1<T, DP, true); break;
        case 3: PGPS_RC2_GO(PGPS_K_SMOOTHER_APPLY, rc2_smooth1<T, DP); break;
        case 4: {       // one Kogge-Stone level over the group totals (a.ks_*)
            const dim3 gk((unsigned)((a.ks_n + 1) / 2));
            if (full) timed_launch(ctx, PGPS_K_FILTER_REDUCE, rc2::rc2_ks_filter<T, DP, true>, gk, blk, 0u, a);
            else if constexpr (PADDED) timed_launch(ctx, PGPS_K_FILTER_REDUCE, rc2::rc2_ks_filter<T, DP, false>, gk, blk, 0u, a);
            break;
        }
        case 5: {       // ... of the smoother's suffix scan
            const dim3 gk((unsigned)((a.ks_n + 1) / 2));
            if (full) timed_launch(ctx, PGPS_K_SMOOTHER_REDUCE, rc2::rc2_ks_smoother<T, DP, true>, gk, blk, 0u, a);
            else if constexpr (PADDED) timed_launch(ctx, PGPS_K_SMOOTHER_REDUCE, rc2::rc2_ks_smoother<T, DP, false>, gk, blk, 0u, a);
            break;
        }
        default: return PGPS_E_INVALID;
    }
#undef PGPS_RC2_GO
    HIPCHK(ctx, hipGetLastError());
    return PGPS_OK;
}

// (instantiated only where the kernels compile without scratch memory: pgps_wc_args.h, rc2_covers)
int PGPS_RC2_LAUNCH(pgps_ctx* ctx, int which, const wc::WcArgs<double>& a) {
    if constexpr (wc::rc2_covers<double>(PGPS_RC2_DP)) return launch_one<double>(ctx, which, a);
    else return PGPS_E_UNSUPPORTED_DIM;
}
int PGPS_RC2_LAUNCH(pgps_ctx* ctx, int which, const wc::WcArgs<float>& a) {
    if constexpr (wc::rc2_covers<float>(PGPS_RC2_DP)) return launch_one<float>(ctx, which, a);
    else return PGPS_E_UNSUPPORTED_DIM;
}

}  // namespace pgps
