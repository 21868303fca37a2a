// pgps_comm.hip -- the communicator a context owns for series sharded over the GPUs of a node: RCCL over xGMI,
// one process per GPU, the context's stream.  No reference equivalent (the reference is single-device:
// experiments/toy_models/speed_and_stability.sh:8-10 pins one device per model); this is the boundary SURVEY.md
// section 8(b)/(e) asks for -- the context owns stream, scratch AND the RCCL communicator, so a sharded pass is
// enqueued without a host round trip and without any framework in the product.
//
// The exchange is two all-gathers of a few hundred bytes (segment totals), latency-bound: all-gather = one hop on
// the fully connected xGMI mesh.
#include <cstring>

#include "pgps_internal.h"      // (pgps_dyn.h: RCCL's types from its header, its functions through dlopen on first use)

static_assert(PGPS_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "pgps.h must carry RCCL's unique-id size");

#define NCCLCHK(ctx, expr)                                                                    \
    do {                                                                                      \
        ncclResult_t r_ = (expr);                                                             \
        if (r_ != ncclSuccess) {                                                              \
            (ctx)->hip_err = std::string(#expr) + ": " + nc.GetErrorString(r_);               \
            return PGPS_E_COMM;                                                               \
        }                                                                                     \
    } while (0)
// the library, or PGPS_E_COMM with the reason in the context's detail string
#define RCCL_OR_FAIL(ctx)                                                                     \
    const pgps::dyn::Rccl& nc = pgps::dyn::rccl();                                            \
    if (!nc.ok) {                                                                             \
        if (ctx) (ctx)->hip_err = "RCCL: " + nc.err;                                          \
        return PGPS_E_COMM;                                                                   \
    }

extern "C" int pgps_comm_get_unique_id(void* id) {
    if (!id) return PGPS_E_INVALID;
    RCCL_OR_FAIL((pgps_ctx*)nullptr)
    ncclUniqueId u;
    if (nc.GetUniqueId(&u) != ncclSuccess) return PGPS_E_COMM;
    std::memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return PGPS_OK;
}

extern "C" int pgps_comm_init(pgps_ctx* ctx, const void* id, int rank, int nranks) {
    if (!ctx || !id || nranks < 1 || rank < 0 || rank >= nranks) return PGPS_E_INVALID;
    if (ctx->comm) return PGPS_E_INVALID;           // one communicator per context; pgps_comm_destroy first
    RCCL_OR_FAIL(ctx)
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ncclUniqueId u;
    std::memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t comm = nullptr;
    NCCLCHK(ctx, nc.CommInitRank(&comm, nranks, u, rank));
    ctx->comm = comm;
    ctx->comm_rank = rank;
    ctx->comm_nranks = nranks;
    return PGPS_OK;
}

extern "C" int pgps_comm_destroy(pgps_ctx* ctx) {
    if (!ctx) return PGPS_E_INVALID;
    if (!ctx->comm) return PGPS_OK;
    RCCL_OR_FAIL(ctx)
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    ncclResult_t r = nc.CommDestroy((ncclComm_t)ctx->comm);
    ctx->comm = nullptr;
    ctx->comm_rank = 0;
    ctx->comm_nranks = 0;
    return r == ncclSuccess ? PGPS_OK : PGPS_E_COMM;
}

// which RCCL the process uses ("librccl.so.1 (already in the process)" when e.g. torch brought its own): loads it
extern "C" int pgps_comm_library(char* buf, size_t n) {
    if (!buf || n == 0) return PGPS_E_INVALID;
    const pgps::dyn::Rccl& nc = pgps::dyn::rccl();
    const std::string& s = nc.ok ? nc.path : nc.err;
    std::strncpy(buf, s.c_str(), n - 1);
    buf[n - 1] = '\0';
    return nc.ok ? PGPS_OK : PGPS_E_COMM;
}

extern "C" int pgps_comm_info(pgps_ctx* ctx, int* rank, int* nranks) {
    if (!ctx || !rank || !nranks) return PGPS_E_INVALID;
    *rank = ctx->comm ? ctx->comm_rank : 0;
    *nranks = ctx->comm ? ctx->comm_nranks : 0;
    return PGPS_OK;
}

// what RCCL itself reports for the communicator (ncclCommCount / ncclCommUserRank): 0 ranks = no communicator
extern "C" int pgps_comm_count(pgps_ctx* ctx, int* nranks, int* rank) {
    if (!ctx || !nranks) return PGPS_E_INVALID;
    *nranks = 0;
    if (rank) *rank = 0;
    if (!ctx->comm) return PGPS_OK;
    RCCL_OR_FAIL(ctx)
    int n = 0, r = 0;
    NCCLCHK(ctx, nc.CommCount((ncclComm_t)ctx->comm, &n));
    NCCLCHK(ctx, nc.CommUserRank((ncclComm_t)ctx->comm, &r));
    *nranks = n;
    if (rank) *rank = r;
    return PGPS_OK;
}

namespace pgps {
int comm_allgather(pgps_ctx* ctx, const void* send, void* recv, size_t bytes) {
    if (!ctx->comm) return PGPS_E_INVALID;
    RCCL_OR_FAIL(ctx)
    NCCLCHK(ctx, nc.AllGather(send, recv, bytes, ncclChar, (ncclComm_t)ctx->comm, ctx->stream));
    return PGPS_OK;
}
}  // namespace pgps

extern "C" int pgps_comm_allgather_dev(pgps_ctx* ctx, const void* send, void* recv, size_t bytes) {
    if (!ctx || !send || !recv || !bytes) return PGPS_E_INVALID;
    return pgps::comm_allgather(ctx, send, recv, bytes);
}
