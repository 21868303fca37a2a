#include <chrono>
// pgps_core.hip -- the C ABI of libpgps.so (include/pgps.h): context, scratch, staging and
// dimension dispatch.  The kernels live in pgps_inst.hip (one unit per dtype x state dim).
// No PyTorch, no TensorFlow: HIP runtime only.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>
#include <cstdio>
#include <cstring>
#include <new>

#include "pgps_internal.h"
#include "pgps_gradlti.h"
#include "pgps_wc_args.h"

using namespace pgps;

namespace pgps {
int ensure(pgps_ctx* ctx, DevBuf& b, size_t bytes) {
    if (&b == &ctx->ws) ++ctx->ws_epoch;            // somebody is about to lay the scratch out (again)
    if (bytes <= b.cap) return PGPS_OK;
    if (b.p) HIPCHK(ctx, hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        ctx->hip_err = std::string("hipMalloc: ") + hipGetErrorString(e);
        b.p = nullptr;
        return PGPS_E_NOMEM;
    }
    b.cap = want;
    return PGPS_OK;
}
}  // namespace pgps

extern "C" int pgps_version(void) { return 100; }

extern "C" const char* pgps_strerror(int code) {
    switch (code) {
        case PGPS_OK: return "ok";
        case PGPS_E_INVALID: return "invalid argument";
        case PGPS_E_UNSUPPORTED_DIM: return "state dimension not supported by the compiled kernels";
        case PGPS_E_HIP: return "HIP runtime error";
        case PGPS_E_NOMEM: return "out of memory";
        case PGPS_E_NUMERIC: return "non-finite result";
        case PGPS_E_NO_DEVICE: return "no HIP device";
        case PGPS_E_COMM: return "RCCL communicator error (pgps_last_hip_error)";
        default: return "unknown error";
    }
}

extern "C" int pgps_device_count(int* n) {
    if (!n) return PGPS_E_INVALID;
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) c = 0;
    *n = c;
    return PGPS_OK;
}

extern "C" int pgps_create(int device, pgps_ctx** out) {
    if (!out) return PGPS_E_INVALID;
    *out = nullptr;
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess || c <= 0) return PGPS_E_NO_DEVICE;
    if (device < 0 || device >= c) return PGPS_E_INVALID;
    pgps_ctx* ctx = new (std::nothrow) pgps_ctx();
    if (!ctx) return PGPS_E_NOMEM;
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return PGPS_E_HIP;
    }
    ctx->stream = ctx->own_stream;
    if (hipDeviceGetAttribute(&ctx->n_cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) ctx->n_cu = 0;
    if (const char* e = std::getenv("PGPS_WC_ROWS2")) ctx->wc_rows2 = std::atoi(e) & 15;            // diagnostic, see pgps_wc.hip
    if (const char* e = std::getenv("PGPS_WC_SERIAL3")) ctx->wc_serial3 = (e[0] == '1');      // diagnostic, see pgps_wc.hip
    if (hipMalloc((void**)&ctx->status_word, pgps::kStatusBytes) != hipSuccess ||
        hipMemset(ctx->status_word, 0, pgps::kStatusBytes) != hipSuccess) {
        (void)hipStreamDestroy(ctx->own_stream);
        delete ctx;
        return PGPS_E_NOMEM;
    }
    *out = ctx;
    return PGPS_OK;
}

extern "C" int pgps_destroy(pgps_ctx* ctx) {
    if (!ctx) return PGPS_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->comm) (void)pgps_comm_destroy(ctx);
    if (ctx->comm_buf.p) (void)hipFree(ctx->comm_buf.p);
    if (ctx->ws.p) (void)hipFree(ctx->ws.p);
    if (ctx->stamps.p) (void)hipFree(ctx->stamps.p);
    if (ctx->res_stamps.p) (void)hipFree(ctx->res_stamps.p);
    if (ctx->gadj.p) (void)hipFree(ctx->gadj.p);
    if (ctx->pin_d.p) (void)hipFree(ctx->pin_d.p);
    if (ctx->pin_h) (void)hipHostFree(ctx->pin_h);
    if (ctx->status_word) (void)hipFree(ctx->status_word);
    for (auto& b : ctx->st)
        if (b.p) (void)hipFree(b.p);
    for (auto& b : ctx->lti)
        if (b.p) (void)hipFree(b.p);
    for (auto& b : ctx->wide)
        if (b.p) (void)hipFree(b.p);
    if (ctx->probe_host) (void)hipHostFree(ctx->probe_host);
    for (auto& e : ctx->ev_pool) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return PGPS_OK;
}

extern "C" int pgps_set_stream(pgps_ctx* ctx, void* s) {
    if (!ctx) return PGPS_E_INVALID;
    ctx->stream = (hipStream_t)s;       // NULL = the HIP null (default) stream, as in HIP itself
    return PGPS_OK;
}

extern "C" int pgps_use_own_stream(pgps_ctx* ctx) {
    if (!ctx) return PGPS_E_INVALID;
    ctx->stream = ctx->own_stream;
    return PGPS_OK;
}

extern "C" int pgps_synchronize(pgps_ctx* ctx) {
    if (!ctx) return PGPS_E_INVALID;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return PGPS_OK;
}

extern "C" int pgps_set_chunk(pgps_ctx* ctx, int c) {
    if (!ctx || c < 0) return PGPS_E_INVALID;
    ctx->chunk = c;
    return PGPS_OK;
}

// diagnostic build only: copy the (3, nblocks, 8) stamp buffer of the last scan to the host
extern "C" int pgps_debug_read_stamps(pgps_ctx* ctx, long long* out, long n_values) {
    if (!ctx || !out) return PGPS_E_INVALID;
    if (!ctx->stamps.p || (size_t)n_values * sizeof(long long) > ctx->stamps.cap) return PGPS_E_INVALID;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipMemcpy(out, ctx->stamps.p, (size_t)n_values * sizeof(long long), hipMemcpyDeviceToHost));
    return PGPS_OK;
}

// Diagnostic flags raised by kernels (bit 1: a look-back spin of the single-pass filter hit its bound).
// Synchronises, returns and clears them.
extern "C" int pgps_status(pgps_ctx* ctx, int* flags) {
    if (!ctx || !flags) return PGPS_E_INVALID;
    *flags = 0;
    if (!ctx->status_word) return PGPS_OK;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipMemcpy(flags, ctx->status_word, sizeof(int), hipMemcpyDeviceToHost));
    if (*flags) HIPCHK(ctx, hipMemset(ctx->status_word, 0, sizeof(int)));
    *flags |= ctx->host_flags;
    ctx->host_flags = 0;
    return PGPS_OK;
}

extern "C" int pgps_set_f32_policy(pgps_ctx* ctx, int policy) {
    if (!ctx || policy < 0 || policy > 2) return PGPS_E_INVALID;
    ctx->f32_policy = policy;
    return PGPS_OK;
}

extern "C" int pgps_set_single_pass(pgps_ctx* ctx, int mode, int window) {
    if (!ctx || mode < -1 || mode > 1 || window < 0 || window > 256) return PGPS_E_INVALID;
    ctx->single_pass = mode;
    if (window > 0) ctx->lookback_window = window;
    return PGPS_OK;
}

extern "C" int pgps_set_shortcut(pgps_ctx* ctx, int on) {
    if (!ctx || on < 0 || on > 1) return PGPS_E_INVALID;
    ctx->shortcut = on;
    return PGPS_OK;
}

extern "C" int pgps_set_resident(pgps_ctx* ctx, int mode) {
    if (!ctx || mode < -1 || mode > 2) return PGPS_E_INVALID;
    ctx->resident = mode;
    return PGPS_OK;
}

// diagnostics: the cycle stamps of the last resident launch made with pgps_set_resident(ctx, 2): (workgroups, 16) long long
extern "C" int pgps_resident_stamps(pgps_ctx* ctx, long long* out, int max_blocks, int* n_blocks) {
    if (!ctx || !n_blocks) return PGPS_E_INVALID;
    *n_blocks = ctx->res_stamp_blocks;
    if (!out || max_blocks <= 0 || !ctx->res_stamps.p) return PGPS_OK;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const int n = ctx->res_stamp_blocks < max_blocks ? ctx->res_stamp_blocks : max_blocks;
    HIPCHK(ctx, hipMemcpy(out, ctx->res_stamps.p, (size_t)n * 16 * sizeof(long long), hipMemcpyDeviceToHost));
    return PGPS_OK;
}

extern "C" int pgps_set_family(pgps_ctx* ctx, int family) {
    if (!ctx || family < 0 || family > 4) return PGPS_E_INVALID;
    ctx->family = family;
    return PGPS_OK;
}
extern "C" int pgps_set_block(pgps_ctx* ctx, int lanes) {
    if (!ctx || (lanes != 0 && lanes != kBlockNarrow && lanes != 256)) return PGPS_E_INVALID;
    ctx->block = lanes;
    return PGPS_OK;
}

extern "C" int pgps_set_one_launch(pgps_ctx* ctx, int max_steps) {
    if (!ctx || max_steps < -1 || max_steps > (1 << 16)) return PGPS_E_INVALID;
    ctx->one_launch = max_steps;
    return PGPS_OK;
}

extern "C" int pgps_set_grad_pack(pgps_ctx* ctx, long max_steps) {
    if (!ctx || max_steps < -1) return PGPS_E_INVALID;
    ctx->grad_pack = max_steps;
    return PGPS_OK;
}

extern "C" int pgps_set_rc_scan(pgps_ctx* ctx, int mode) {
    if (!ctx || mode < -1 || mode > 1) return PGPS_E_INVALID;
    ctx->rc_scan = mode;
    return PGPS_OK;
}

extern "C" int pgps_set_dma(pgps_ctx* ctx, int mode) {
    if (!ctx || mode < -1 || mode > 1) return PGPS_E_INVALID;
    ctx->dma = mode;
    return PGPS_OK;
}

extern "C" int pgps_set_stage(pgps_ctx* ctx, int g) {
    if (!ctx || !(g == -1 || g == 0 || g == 2 || g == 4)) return PGPS_E_INVALID;
    ctx->stage_g = g;
    return PGPS_OK;
}

extern "C" const char* pgps_last_hip_error(pgps_ctx* ctx) { return ctx ? ctx->hip_err.c_str() : ""; }

extern "C" int pgps_malloc(pgps_ctx* ctx, size_t bytes, void** dptr) {
    if (!ctx || !dptr) return PGPS_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(dptr, bytes ? bytes : 16);
    if (e != hipSuccess) { ctx->hip_err = hipGetErrorString(e); return PGPS_E_NOMEM; }
    return PGPS_OK;
}
extern "C" int pgps_free(pgps_ctx* ctx, void* dptr) {
    if (!ctx) return PGPS_E_INVALID;
    if (dptr) HIPCHK(ctx, hipFree(dptr));
    return PGPS_OK;
}
extern "C" int pgps_memcpy_h2d(pgps_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (!ctx || (!dst && bytes) || (!src && bytes)) return PGPS_E_INVALID;
    HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return PGPS_OK;
}
extern "C" int pgps_memcpy_d2h(pgps_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (!ctx || (!dst && bytes) || (!src && bytes)) return PGPS_E_INVALID;
    HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return PGPS_OK;
}

// ---------------------------------------------------------------------------------------------
// profiling
// ---------------------------------------------------------------------------------------------
static const char* kKernelNames[PGPS_K_COUNT] = {"k_filter_reduce", "k_filter_apply", "k_smoother_reduce",
                                                 "k_smoother_apply", "k_ll_finalize", "k_discretise", "k_pkfs_resident"};
extern "C" const char* pgps_kernel_name(int slot) {
    return (slot >= 0 && slot < PGPS_K_COUNT) ? kKernelNames[slot] : "";
}

namespace pgps {
int prof_flush(pgps_ctx* ctx) {
    if (ctx->ev_used == 0) return PGPS_OK;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < ctx->ev_used; ++i) {
        float ms = 0.f;
        HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev_pool[i].a, ctx->ev_pool[i].b));
        ctx->prof_ms[ctx->ev_pool[i].slot] += ms;
        ctx->prof_n[ctx->ev_pool[i].slot] += 1;
    }
    ctx->ev_used = 0;
    return PGPS_OK;
}

pgps_ctx::EvPair* prof_acquire(pgps_ctx* c, int slot) {
    if (!((c->profiling >> slot) & 1u)) return nullptr;
    if ((c->prof_seen[slot]++ % c->prof_every) != 0) return nullptr;
    if (c->ev_used == c->ev_pool.size()) {
        if (c->ev_pool.size() >= 1024) {
            if (prof_flush(c) != PGPS_OK) return nullptr;
        } else {
            pgps_ctx::EvPair p;
            if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return nullptr;
            c->ev_pool.push_back(p);
        }
    }
    pgps_ctx::EvPair* ev = &c->ev_pool[c->ev_used++];
    ev->slot = slot;
    return ev;
}
}  // namespace pgps

extern "C" int pgps_profile_enable(pgps_ctx* ctx, int on) {
    if (!ctx) return PGPS_E_INVALID;
    if (!on) { int rc = prof_flush(ctx); if (rc) return rc; }
    if (on && ctx->ev_pool.empty()) {
        // create the events (and exercise them once) up front: the first hipEventRecord on a fresh
        // event allocates its signal, which can stall the queue for milliseconds
        HIPCHK(ctx, hipSetDevice(ctx->device));
        for (int i = 0; i < 1024; ++i) {
            pgps_ctx::EvPair p;
            HIPCHK(ctx, hipEventCreate(&p.a));
            HIPCHK(ctx, hipEventCreate(&p.b));
            p.slot = 0;
            ctx->ev_pool.push_back(p);
            HIPCHK(ctx, hipEventRecord(p.a, ctx->stream));
            HIPCHK(ctx, hipEventRecord(p.b, ctx->stream));
        }
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    ctx->profiling = (unsigned)on;
    for (int i = 0; i < PGPS_K_COUNT; ++i) ctx->prof_seen[i] = 0;
    return PGPS_OK;
}

// mean elapsed time of an EMPTY hipEvent pair on the context's stream: what a pair adds to the
// duration it brackets (subtract it to compare with a profiler's pure kernel time)
extern "C" int pgps_profile_calibrate(pgps_ctx* ctx, double* empty_pair_ms) {
    if (!ctx || !empty_pair_ms) return PGPS_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipEvent_t a, b;
    HIPCHK(ctx, hipEventCreate(&a));
    HIPCHK(ctx, hipEventCreate(&b));
    double acc = 0.0;
    const int reps = 64;
    for (int i = 0; i < reps + 8; ++i) {
        HIPCHK(ctx, hipEventRecord(a, ctx->stream));
        HIPCHK(ctx, hipEventRecord(b, ctx->stream));
        HIPCHK(ctx, hipEventSynchronize(b));
        float ms = 0.f;
        HIPCHK(ctx, hipEventElapsedTime(&ms, a, b));
        if (i >= 8) acc += ms;
    }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    *empty_pair_ms = acc / reps;
    return PGPS_OK;
}

extern "C" int pgps_profile_sample(pgps_ctx* ctx, int every_n) {
    if (!ctx || every_n < 1) return PGPS_E_INVALID;
    ctx->prof_every = every_n;
    return PGPS_OK;
}

extern "C" int pgps_profile_read(pgps_ctx* ctx, double* total_ms, long* launches, int reset) {
    if (!ctx) return PGPS_E_INVALID;
    int rc = prof_flush(ctx);
    if (rc) return rc;
    for (int i = 0; i < PGPS_K_COUNT; ++i) {
        if (total_ms) total_ms[i] = ctx->prof_ms[i];
        if (launches) launches[i] = ctx->prof_n[i];
        if (reset) { ctx->prof_ms[i] = 0; ctx->prof_n[i] = 0; }
    }
    return PGPS_OK;
}

// ---------------------------------------------------------------------------------------------
// launch geometry
// ---------------------------------------------------------------------------------------------
namespace pgps {
void geometry(const pgps_ctx* ctx, long N, int* Lc, int* nblocks, int d) {
    int c = ctx->chunk;
    if (c <= 0) {
        // 16 steps per lane (the lane-serial part then outweighs the scan trees: measured at 2^20)
        // until that would need more than 1024 workgroups (every workgroup folds the spine entries
        // on its side); shorter series use fewer steps per lane so the chip is still covered;
        // multiples of 4 = whole LDS-staged sub-tiles.
        long v = 16;
        // from 2^21 steps 32 per lane still fill every CU (>= 256 workgroups) and halve the scan trees and their scratch
        // per step: 2^21 0.166 -> 0.154 ms, 2^24 1.27 -> 1.13 ms; at 2^20 half the CUs would idle (0.086 -> 0.092 ms)
        if (d >= 1 && d <= 2 && N >= (long)kBlock * 32 * 256) v = 32;        // (measured on the array path at d = 2)
        // (2^24 steps, d = 2: 2048 workgroups of 32 steps per lane 1.13 ms, 1024 of 64 steps 1.29 ms)
        const long max_blocks = (d >= 1 && d <= 2) ? 2048 : 1024;
        if (N > (long)kBlock * v * max_blocks) v = (N + (long)kBlock * max_blocks - 1) / ((long)kBlock * max_blocks);
        while (v > 4 && (long)kBlock * v * 128 > N) v /= 2;   // keep >= 128 workgroups when N allows
        if (N < (long)kBlock * 4) v = (N + kBlock - 1) / kBlock;
        if (v < 1) v = 1;
        if (v > 4) v = (v + 3) / 4 * 4;
        c = (int)v;
    }
    long nb = (N + (long)kBlock * c - 1) / ((long)kBlock * c);
    if (nb < 1) nb = 1;
    *Lc = c;
    *nblocks = (int)nb;
}
// Steps per lane and workgroups of the 128-lane build.  Up to d = 3 one workgroup per CU where the series allows it (256
// workgroups, up to 32 steps per lane: measured at d = 2 from 2^17 to 2^20 steps and at d = 3), from d = 4 sixteen
// steps per lane (RBF order 4 / 6 at 2^20: 16 and 32 steps per lane 0.40 / 0.40 and 0.59 / 0.62 ms).
void geometry_narrow(const pgps_ctx* ctx, long N, int* Lc, int* nblocks, int d) {
    int c = ctx->chunk;
    if (c <= 0) {
        long v;
        if (d <= 3) {
            v = (N + (long)kBlockNarrow * 256 - 1) / ((long)kBlockNarrow * 256);
            v = v < 4 ? 4 : (v > 32 ? 32 : v);
            const long max_blocks = 4096;
            if (N > (long)kBlockNarrow * v * max_blocks) v = (N + (long)kBlockNarrow * max_blocks - 1) / ((long)kBlockNarrow * max_blocks);
        } else {
            v = 16;
            const long max_blocks = 2048;
            if (N > (long)kBlockNarrow * v * max_blocks) v = (N + (long)kBlockNarrow * max_blocks - 1) / ((long)kBlockNarrow * max_blocks);
            while (v > 4 && (long)kBlockNarrow * v * 256 > N) v /= 2;
        }
        if (N < (long)kBlockNarrow * 4) v = (N + kBlockNarrow - 1) / kBlockNarrow;
        if (v < 1) v = 1;
        if (v > 4) v = (v + 3) / 4 * 4;
        c = (int)v;
    }
    long nb = (N + (long)kBlockNarrow * c - 1) / ((long)kBlockNarrow * c);
    *Lc = c;
    *nblocks = (int)(nb < 1 ? 1 : nb);
}
}  // namespace pgps

namespace pgps {
// which of the two builds of the lane-chunk scan a call (or a rank's segment) of N steps at state dimension d takes
static bool lane_narrow(const pgps_ctx* ctx, int d, long N) {
    return ctx->block != 256 && (ctx->block == kBlockNarrow || !(d <= 3 && N >= (1L << 22)));
}
}  // namespace pgps

namespace pgps {
// The resident launch (pgps_resident.hip.h) serves whole-series filter + smoother calls at d = 2 in fp64 whose
// 256 x 16-step workgroups are all resident at once (one per CU); automatic from kResAutoMin steps, where the scan's
// streaming outweighs the two grid barriers (below that the narrow build's smaller chunks cover more of the chip).
constexpr long kResAutoMin = 1L << 17;       // (2^17 steps: 26.9 against 28.8 us on three launches; equal at 2^16: profiles/r05_experiments.txt item 12)
bool resident_fits(const pgps_ctx* ctx, long N, int d, bool f32) {
    if (f32 || d != 2 || ctx->resident == 0 || ctx->n_cu <= 0) return false;
    // a pinned geometry or variant of the three-launch path was asked for (a chunk of 8 or 16 together with mode >= 1 pins the
    // resident launch's own steps per lane instead: tests, A/B)
    const bool own_chunk = ctx->resident > 0 && (ctx->chunk == 8 || ctx->chunk == 16);
    if ((ctx->chunk > 0 && !own_chunk) || ctx->block != 0 || ctx->stage_g >= 0 || ctx->single_pass > 0 || ctx->dma > 0) return false;
    if (ctx->family != 0 && ctx->family != 1) return false;
    if (N > (long)kBlock * (own_chunk ? ctx->chunk : kResLc) * ctx->n_cu) return false;
    // not while the stream is being captured: the launch's barrier set and hand-off epoch are chosen per launch on the host,
    // and a replayed graph would present the same ones again (counters already at their targets, flags already equal)
    if (!(ctx->resident > 0 || N >= kResAutoMin)) return false;         // (before the query below: short series never pay for it)
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(ctx->stream, &cs) != hipSuccess) { (void)hipGetLastError(); return false; }
    return cs == hipStreamCaptureStatusNone;
}
}  // namespace pgps

extern "C" int pgps_get_geometry(pgps_ctx* ctx, long N, int d, int* lanes, int* Lc, int* nb) {
    if (!ctx || N < 1 || d < 1 || d > PGPS_MAX_DIM_LANE || !lanes || !Lc || !nb) return PGPS_E_INVALID;
    if (lane_narrow(ctx, d, N)) { *lanes = kBlockNarrow; geometry_narrow(ctx, N, Lc, nb, d); }
    else { *lanes = kBlock; geometry(ctx, N, Lc, nb, d); }
    return PGPS_OK;
}

extern "C" int pgps_get_chunk(pgps_ctx* ctx, long N, int* Lc, int* nb) {
    if (!ctx || N < 1 || !Lc || !nb) return PGPS_E_INVALID;
    geometry(ctx, N, Lc, nb);
    return PGPS_OK;
}

// fp32 <-> fp64 conversion passes (the discretisation kernels of 7 <= d <= 16 compute in fp64 whatever the series' type)
namespace pgps {
static __global__ void k_widen(long n, const float* in, double* out) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = (double)in[i];
}
static __global__ void k_narrow(long n, const double* in, float* out) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = (float)in[i];
}
// up to eight arrays in one launch (blockIdx.y selects the array): the promoted float32 calls convert seven inputs and four
// outputs, and a launch costs the host more than converting a short series does
struct ConvJobs {
    const void* src[8];
    void* dst[8];
    long n[8];
};
static __global__ void k_widen_many(ConvJobs j) {
    const float* in = (const float*)j.src[blockIdx.y];
    double* out = (double*)j.dst[blockIdx.y];
    const long n = j.n[blockIdx.y];
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = (double)in[i];
}
static __global__ void k_narrow_many(ConvJobs j) {
    const double* in = (const double*)j.src[blockIdx.y];
    float* out = (float*)j.dst[blockIdx.y];
    const long n = j.n[blockIdx.y];
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = (float)in[i];
}
}  // namespace pgps

// Which kernel family a scan call of N steps at state dimension d takes (PGPS_FAMILY_*, include/pgps.h): the ONE place
// that decides -- dispatch_scan launches what this returns, pgps_get_family reports it (bench.py names the measured
// kernels from it instead of repeating the rule).
template <typename T>
static int choose_family(const pgps_ctx* ctx, int d, long N, Mode mode) {
    const bool rc_ok = d >= rc::kDimMin && d <= rc::kDimMax;
    if constexpr (sizeof(T) == 4) {
        // row-cooperative family in fp32: its own instantiations (16-lane rows, v_fmac_f32_dpp), every mode; automatic
        // above the lane-chunk kernels' range (at d = 6 those still win in fp32: 0.71 against 0.85 ms at 2^20 steps)
        bool to_rc = rc_ok && (ctx->family == 3 || ((ctx->family == 0 || ctx->family == 4) && d > PGPS_MAX_DIM_LANE));
        // d = 6, whole-series filter / filter + smoother: the quad-cooperative kernels are ahead of the lane-chunk ones on
        // short series; from 2^19 steps the lane-chunk kernels' workgroups span >= 2048 steps and take their carries by the
        // forgetting shortcut (round 5), which put them ahead at every longer size (same box, ms per pass, lane-chunk / quad,
        // profiles/r05_d6_crossover.txt: 2^14 0.177 / 0.128, 2^16 0.191 / 0.146, 2^17 0.202 / 0.182, 2^18 0.245 / 0.222,
        // 2^19 0.258 / 0.354, 2^20 0.549 / 0.628, 2^21 1.042 / 1.163, 2^22 2.039 / 2.204; round 4, without the shortcut:
        // 2^19 0.395 / 0.398, 2^20 0.627 / 0.650, 2^21 1.266 / 1.217, 2^22 2.566 / 2.357)
        if (ctx->family == 0 && d == 6 && (mode == MODE_PKF || mode == MODE_PKFS) && ctx->chunk == 0 && ctx->stage_g < 0 &&
            N <= (3L << 17) && N >= 64)
            to_rc = true;
        // quad-cooperative level-1 kernels under the row-cooperative driver: family 4 (fp32, 5 <= d <= 8)
        if (ctx->family == 4) {
            if (d < qc::kDimMin || d > qc::kDimMax || mode == MODE_PKS) return PGPS_E_UNSUPPORTED_DIM;
            to_rc = true;
        }
        if (to_rc) {
            // (what scan_rc_entry then decides: the quad level-1 kernels at d = 8 and, where this rule sends it there, d = 6)
            const bool quad = (ctx->family == 4 || (ctx->family == 0 && (d == 8 || d == 6))) && d >= qc::kDimMin && d <= qc::kDimMax &&
                              mode != MODE_PKS;
            return quad ? PGPS_FAMILY_QUAD : PGPS_FAMILY_ROW;
        }
    }
    if constexpr (sizeof(T) == 8) {
        // filter + smoother of a whole series that fits the chip: one resident launch (pgps_resident.hip.h)
        if ((mode == MODE_PKFS || mode == MODE_PKF) && resident_fits(ctx, N, d, false)) return PGPS_FAMILY_RESIDENT;
        // row-cooperative family: fp64, d <= 16, whole-series filter / filter+smoother
        const bool whole = mode == MODE_PKF || mode == MODE_PKFS || mode == MODE_PKS;
        // automatic choice from d = 5: at d = 6 the lane-chunk kernels spill (2^18 steps: 1.29 ms against 0.53 ms);
        // the segment protocol (multi-GPU) moves over where the lane-chunk family ends
        if (rc_ok && (ctx->family == 3 || (ctx->family == 0 && (whole ? d >= 5 : d > PGPS_MAX_DIM_LANE)))) return PGPS_FAMILY_ROW;
    }
    if (ctx->family == 3) return PGPS_E_UNSUPPORTED_DIM;
    if (ctx->family == 2 || (ctx->family == 0 && d > PGPS_MAX_DIM_LANE)) {
        if (d > 32 || mode == MODE_PKS) return PGPS_E_UNSUPPORTED_DIM;
        return (wc::rc2_covers<T>(d) && ctx->wc_rows2) ? PGPS_FAMILY_TWO_ROWS : PGPS_FAMILY_WAVE;
    }
    if (d < 1 || d > PGPS_MAX_DIM_LANE) return PGPS_E_UNSUPPORTED_DIM;
    // Lane-chunk family: whole-series calls run the build with 128-lane workgroups (pgps_inst.hip, PGPS_NARROW) --
    // except the long series of the LDS-staged dimensions: from 2^22 steps there are two waves per SIMD to cover each
    // other's loads, and the narrow build's prefetch registers cost it that (d = 2, 2^22 steps: 0.303 ms against 0.307 for
    // 256 lanes; 2^24: 1.28 against 1.25).  The three phases of the segment protocol follow the same rule (it depends on
    // this rank's N and d only, so they agree with each other: a rank's 2^21 steps of c4 0.152 -> 0.147 ms).
    return lane_narrow(ctx, d, N) ? PGPS_FAMILY_LANE_NARROW : PGPS_FAMILY_LANE;
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

template <typename T>
static int dispatch_scan(pgps_ctx* ctx, int d, const ScanArgs<T>& a, Mode mode) {
    int fam = choose_family<T>(ctx, d, a.N, mode);
    if (fam < 0) return fam;
    if constexpr (sizeof(T) == 8) {
        if (fam == PGPS_FAMILY_RESIDENT) {
            if (aligned16(a.ys)) {
                ResArgs<double> ra{};
                ra.s = a;
                return launch_resident<double, 2>(ctx, ra, false, mode == MODE_PKFS);
            }
            fam = lane_narrow(ctx, d, a.N) ? PGPS_FAMILY_LANE_NARROW : PGPS_FAMILY_LANE;      // (a misaligned ys: three launches)
        }
    }
    if (fam == PGPS_FAMILY_ROW || fam == PGPS_FAMILY_QUAD) return launch_scan_rc<T>(ctx, a, d, mode);
    if (fam == PGPS_FAMILY_WAVE || fam == PGPS_FAMILY_TWO_ROWS) return launch_scan_wc<T>(ctx, a, d, mode);
    if (fam == PGPS_FAMILY_LANE_NARROW) {
        switch (d) {
            case 1: return launch_scan_narrow<T, 1>(ctx, a, mode);
            case 2: return launch_scan_narrow<T, 2>(ctx, a, mode);
            case 3: return launch_scan_narrow<T, 3>(ctx, a, mode);
            case 4: return launch_scan_narrow<T, 4>(ctx, a, mode);
            case 5: return launch_scan_narrow<T, 5>(ctx, a, mode);
            case 6: return launch_scan_narrow<T, 6>(ctx, a, mode);
            default: return PGPS_E_UNSUPPORTED_DIM;
        }
    }
    switch (d) {
        case 1: return launch_scan<T, 1>(ctx, a, mode);
        case 2: return launch_scan<T, 2>(ctx, a, mode);
        case 3: return launch_scan<T, 3>(ctx, a, mode);
        case 4: return launch_scan<T, 4>(ctx, a, mode);
        case 5: return launch_scan<T, 5>(ctx, a, mode);
        case 6: return launch_scan<T, 6>(ctx, a, mode);
        default: return PGPS_E_UNSUPPORTED_DIM;
    }
}

// `what`: 0 = pkf, 1 = pks, 2 = pkfs, 3 = a phase of the segment protocol; fp64 unless f32 != 0
extern "C" int pgps_get_family(pgps_ctx* ctx, long N, int d, int f32, int what, int* family) {
    if (!ctx || N < 1 || !family || what < 0 || what > 3) return PGPS_E_INVALID;
    const Mode mode = what == 0 ? MODE_PKF : what == 1 ? MODE_PKS : what == 2 ? MODE_PKFS : MODE_SEG_FILTER;
    const int fam = f32 ? choose_family<float>(ctx, d, N, mode) : choose_family<double>(ctx, d, N, mode);
    if (fam < 0) return fam;
    *family = fam;
    return PGPS_OK;
}


// ---------------------------------------------------------------------------------------------
// float32 series on DENSE grids: fp64 arithmetic behind float32 arrays, chosen per call.
// The reference's speed protocol takes --dtype (pssgp/experiments/toy_models/speed_and_stability.py:68) on
// np.linspace(0, 4, N) (toy_models/common.py:31-32): at 2^20 points F_k is the identity to five digits, and the smoothing
// elements' L = P - E Pp E^T (pssgp/kalman/parallel.py:159-166) -- like the sequential form P + G (sP' - Pp) G^T of
// sequential.py:57-61 -- is a difference of nearly equal matrices behind a solve with cond(Pp) ~ 1e5: float32 ARITHMETIC
// misses the north star's 1e-3 there whatever the kernel family (profiles/r03_fp32_reference_grid.txt), while fp64
// arithmetic on the float32 ARRAYS holds 1e-4 (profiles/r04_fp32_reference_grid.txt: the rounding of the inputs is not
// the problem).  So calls that run a smoother (pkfs, pks) on float32 arrays probe the grid first: a few thousand
// transition matrices spread over the series, ||F_k - I||_max against a threshold that grows with the state dimension
// (the error of the float32 smoother does: measured at d = 2, 3, 6); when an eighth of them or more are that close to the
// identity, the arrays are widened into scratch, the fp64 kernels run, the results are rounded back -- and
// pgps_status reports PGPS_STATUS_F32_PROMOTED.  pgps_set_f32_policy(ctx, 1) keeps float32 arithmetic whatever the
// grid, 2 always widens.  Filter-only calls (pkf) hold 1e-3 natively on every grid measured and are never probed.
// ---------------------------------------------------------------------------------------------
namespace pgps {
// ONE workgroup of 1024 lanes: sixteen lanes share a sampled transition matrix (consecutive lanes read consecutive entries:
// whole 64-byte segments -- one lane per matrix made every load instruction touch 64 cache lines, 17 us of one CU's address
// unit for 1024 samples), kProbeRounds samples per group; the count of dense samples comes out of a workgroup reduction -- no
// inter-workgroup atomics, no counters to reset.  result[1] = the count, then result[0] = the call's sequence number
// (system-scope release): the host spins on that word, no event involved.
constexpr int kProbeGroup = 16, kProbeRounds = 4, kProbeSamples = 1024 / kProbeGroup * kProbeRounds;
static __global__ __launch_bounds__(1024) void k_f32_probe(long N, int d, const float* __restrict__ Fs, float tau, long stride,
                                                             int nsamp, int* result, int seq) {
    const int g = threadIdx.x / kProbeGroup, j = threadIdx.x % kProbeGroup;
    const int dd = d * d;
    int dense = 0;
    for (int r = 0; r < kProbeRounds; ++r) {
        const int s = r * (1024 / kProbeGroup) + g;
        float m = 0.f;
        if (s < nsamp) {
            long k = 1 + (long)s * stride;          // (step 0 spans t0 .. t_0: whatever the grid, it may be long)
            if (k >= N) k = N - 1;
            const float* F = Fs + k * (long)dd;
            for (int e = j; e < dd; e += kProbeGroup) m = fmaxf(m, fabsf(F[e] - ((e / d == e % d) ? 1.f : 0.f)));
        }
        for (int o = kProbeGroup / 2; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, kProbeGroup));
        dense += (j == 0 && s < nsamp && m < tau) ? 1 : 0;
    }
    // (a lane's count is 0..kProbeRounds: sum them over the workgroup)
    int total = 0;
    for (int c = 1; c <= kProbeRounds; ++c) total += __syncthreads_count(dense >= c);
    if (threadIdx.x == 0) {
        __hip_atomic_store(result + 1, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(result, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
}  // namespace pgps

static float f32_dense_threshold(int d) {
    // ||F - I||_max below which the float32 smoother is not trusted.  Measured (max-norm relative error of the smoothed
    // covariance, reference grid): d = 2 / 3: 5.5e-4 / 4.0e-4 at 1e-5; d = 6: 1.1e-4 at 8e-3, 3.7e-3 at 1e-3.  From d = 17
    // (two-rows kernels, 2e-3 on an ordinary grid) every smoother call is promoted.
    if (d <= 3) return 1e-4f;
    if (d == 4) return 5e-4f;
    if (d == 5) return 2e-3f;
    if (d <= 8) return 5e-3f;
    if (d <= 16) return 1e-2f;
    return 3.0e38f;
}

// The probe of a float32 smoother call: ONE small launch on the context's stream, in front of the call's own kernels, whose
// last workgroup writes its verdict and the call's sequence number to pinned host memory.  *fixed: the policy already decides
// (no probe): 0 / 1 = float32 / fp64 arithmetic.
// (Round 4 ran the probe on a stream of its own between two events -- hipEventRecord on the context's stream, a
// hipStreamWaitEvent, a second event the host synchronised on: beside the call's first kernel instead of in front of it, but
// each event record is a barrier packet on the stream that costs the pass ~5 us (the same effect bench.py's per-launch stamps
// showed in round 5), which is where the probe's +3 .. 4.6 % on c3 came from.  One 4 us kernel costs less than its plumbing.)
static int f32_probe_launch(pgps_ctx* ctx, long N, int d, const float* Fs, int* fixed, int* nsamp_out) {
    *fixed = -1;
    if (ctx->f32_policy == 1) { *fixed = 0; return PGPS_OK; }
    if (ctx->f32_policy == 2 || d > 16) { *fixed = 1; return PGPS_OK; }
    if (N < 3) { *fixed = 0; return PGPS_OK; }
    if (!ctx->probe_host) {
        // all or nothing: the context keeps the pinned words only once every step of the setup has succeeded
        int* host = nullptr;
        int* dev = nullptr;
        HIPCHK(ctx, hipHostMalloc((void**)&host, 64, hipHostMallocDefault));
        if (hipHostGetDevicePointer((void**)&dev, host, 0) != hipSuccess) {
            (void)hipHostFree(host);
            ctx->hip_err = "float32 probe: pinned result words could not be set up";
            return PGPS_E_HIP;
        }
        host[0] = 0;
        host[1] = 0;
        ctx->probe_host = host;
        ctx->probe_dev = dev;
        ctx->probe_seq = 0;
    }
    const int nsamp = (int)std::min<long>(pgps::kProbeSamples, N - 1);
    const long stride = std::max<long>(1, (N - 1) / nsamp);
    ctx->probe_seq = (ctx->probe_seq % 0x3fffffff) + 1;         // never 0: the words start at 0
    hipLaunchKernelGGL(pgps::k_f32_probe, dim3(1), dim3(1024), 0, ctx->stream, N, d, Fs, f32_dense_threshold(d), stride, nsamp,
                       ctx->probe_dev, ctx->probe_seq);
    HIPCHK(ctx, hipGetLastError());
    *nsamp_out = nsamp;
    return PGPS_OK;
}
// ... and its answer: the host spins on the pinned sequence word (no HIP call, no event: the probe is the first thing this
// call put on the stream, whatever the call enqueues behind it keeps the GPU busy meanwhile).  Bounded: a stream that never
// reaches the probe (a hung predecessor) ends the call with PGPS_E_HIP after ~20 s instead of hanging the host.
static int f32_probe_result(pgps_ctx* ctx, int nsamp, int* dense) {
    volatile int* w = ctx->probe_host;
    const auto t0 = std::chrono::steady_clock::now();
    long spins = 0;
    while (__atomic_load_n(&w[0], __ATOMIC_ACQUIRE) != ctx->probe_seq) {
        if ((++spins & 0xfff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) {
            ctx->hip_err = "float32 probe: no verdict from the device within 20 s";
            return PGPS_E_HIP;
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    *dense = (long)w[1] * 8 >= nsamp;
    return PGPS_OK;
}

static int f32_run_wide(pgps_ctx* ctx, int d, const ScanArgs<float>& a, Mode mode);

// A float32 smoother call (pkfs or pks).  Waiting for the probe before anything else is enqueued leaves the GPU idle for the
// host's launch latency behind every call (3.4 % of BASELINE's c3 pass, measured); enqueuing the float32 pass first and
// asking afterwards wastes that pass when the grid turns out dense.  The context remembers which way its last probed call
// went and orders the next one accordingly -- float32 pass first after a float32 call (the wait then hides behind the
// call's own kernels: no idle time, and the float32 pass is simply overwritten by the fp64 one if the grid has become
// dense), probe first after a promoted call.  The RESULT never depends on the memory: only what is enqueued when.
static int f32_smoother_call(pgps_ctx* ctx, int d, const ScanArgs<float>& a, Mode mode) {
    int fixed = -1, nsamp = 0, dense = 0;
    int rc_ = f32_probe_launch(ctx, a.N, d, a.Fs, &fixed, &nsamp);
    if (rc_) return rc_;
    if (fixed == 0) return dispatch_scan<float>(ctx, d, a, mode);
    if (fixed == 1) return f32_run_wide(ctx, d, a, mode);
    if (ctx->f32_last_promoted) {
        if ((rc_ = f32_probe_result(ctx, nsamp, &dense))) return rc_;
        ctx->f32_last_promoted = dense;
        return dense ? f32_run_wide(ctx, d, a, mode) : dispatch_scan<float>(ctx, d, a, mode);
    }
    if ((rc_ = dispatch_scan<float>(ctx, d, a, mode))) {
        (void)f32_probe_result(ctx, nsamp, &dense);      // (the probe reads the caller's Fs: it has run before the error is returned)
        return rc_;
    }
    if ((rc_ = f32_probe_result(ctx, nsamp, &dense))) return rc_;
    ctx->f32_last_promoted = dense;
    return dense ? f32_run_wide(ctx, d, a, mode) : PGPS_OK;
}

// the float32 call `a` (whole series: pkfs or pks) in fp64 arithmetic
static int f32_run_wide(pgps_ctx* ctx, int d, const ScanArgs<float>& a, Mode mode) {
    const size_t n = (size_t)a.N, dd = (size_t)d * d;
    const size_t sizes[9] = {dd, (size_t)d, n * dd, n * dd, n, n * d, n * dd, n * d, n * dd};
    double* w[9];
    for (int i = 0; i < 9; ++i) {
        int rc_ = ensure(ctx, ctx->wide[i], sizes[i] * sizeof(double));
        if (rc_) return rc_;
        w[i] = (double*)ctx->wide[i].p;
    }
    // one conversion launch in, one out
    auto convert = [&](bool widen_dir, std::initializer_list<std::pair<const void*, int>> jobs) {
        pgps::ConvJobs cj{};
        int nj = 0;
        size_t most = 0;
        for (auto& jb : jobs) {
            if (!jb.first) continue;
            const int i = jb.second;
            cj.src[nj] = widen_dir ? jb.first : (const void*)w[i];
            cj.dst[nj] = widen_dir ? (void*)w[i] : const_cast<void*>(jb.first);
            cj.n[nj] = (long)sizes[i];
            most = std::max(most, sizes[i]);
            ++nj;
        }
        if (!nj) return;
        const dim3 g((unsigned)std::min<size_t>(4096, (most + 255) / 256), (unsigned)nj);
        if (widen_dir) hipLaunchKernelGGL(pgps::k_widen_many, g, dim3(256), 0, ctx->stream, cj);
        else hipLaunchKernelGGL(pgps::k_narrow_many, g, dim3(256), 0, ctx->stream, cj);
    };
    if (mode == MODE_PKS) convert(true, {{a.P0, 0}, {a.H, 1}, {a.Fs, 2}, {a.Qs, 3}, {a.ys, 4}, {a.fms, 5}, {a.fPs, 6}});
    else convert(true, {{a.P0, 0}, {a.H, 1}, {a.Fs, 2}, {a.Qs, 3}, {a.ys, 4}});
    ScanArgs<double> b{};
    b.N = a.N; b.seg_first = 1; b.seg_last = 1;
    b.P0 = a.P0 ? w[0] : nullptr; b.H = a.H ? w[1] : nullptr; b.R = (double)a.R;
    b.Fs = w[2]; b.Qs = w[3]; b.ys = a.ys ? w[4] : nullptr;
    b.fms = w[5]; b.fPs = w[6]; b.sms = w[7]; b.sPs = w[8]; b.ll = a.ll;
    int rc_ = dispatch_scan<double>(ctx, d, b, mode);
    if (rc_) return rc_;
    if (mode != MODE_PKS) convert(false, {{a.fms, 5}, {a.fPs, 6}, {a.sms, 7}, {a.sPs, 8}});
    else convert(false, {{a.sms, 7}, {a.sPs, 8}});
    HIPCHK(ctx, hipGetLastError());
    ctx->host_flags |= PGPS_STATUS_F32_PROMOTED;
    return PGPS_OK;
}

// ---------------------------------------------------------------------------------------------
// device-pointer entry points
// ---------------------------------------------------------------------------------------------
template <typename T>
static int pkf_dev(pgps_ctx* ctx, long N, int d, const T* P0, const T* Fs, const T* Qs, const T* H, T R,
                   const T* ys, T* fms, T* fPs, double* ll) {
    RoctxRange range_("parallel_filter");
    if (!ctx || N < 1 || !P0 || !Fs || !Qs || !H || !ys || !fms || !fPs) return PGPS_E_INVALID;
    if (d < 1 || d > PGPS_MAX_DIM) return PGPS_E_UNSUPPORTED_DIM;
    if (!aligned16(Fs) || !aligned16(Qs) || !aligned16(fms) || !aligned16(fPs)) return PGPS_E_INVALID;
    ScanArgs<T> a{};
    a.N = N; a.seg_first = 1; a.seg_last = 1;
    a.P0 = P0; a.H = H; a.R = R; a.Fs = Fs; a.Qs = Qs; a.ys = ys;
    a.fms = fms; a.fPs = fPs; a.ll = ll;
    return dispatch_scan<T>(ctx, d, a, MODE_PKF);
}

template <typename T>
static int pks_dev(pgps_ctx* ctx, long N, int d, const T* Fs, const T* Qs, const T* fms, const T* fPs,
                   T* sms, T* sPs) {
    RoctxRange range_("parallel_smoother");
    if (!ctx || N < 1 || !Fs || !Qs || !fms || !fPs || !sms || !sPs) return PGPS_E_INVALID;
    if (d < 1 || d > PGPS_MAX_DIM) return PGPS_E_UNSUPPORTED_DIM;
    if (!aligned16(Fs) || !aligned16(Qs) || !aligned16(fms) || !aligned16(fPs) || !aligned16(sms) ||
        !aligned16(sPs))
        return PGPS_E_INVALID;
    ScanArgs<T> a{};
    a.N = N; a.seg_first = 1; a.seg_last = 1;
    a.Fs = Fs; a.Qs = Qs;
    a.fms = const_cast<T*>(fms); a.fPs = const_cast<T*>(fPs); a.sms = sms; a.sPs = sPs;
    if constexpr (sizeof(T) == 4) return f32_smoother_call(ctx, d, a, MODE_PKS);
    else return dispatch_scan<T>(ctx, d, a, MODE_PKS);
}

template <typename T>
static int pkfs_dev(pgps_ctx* ctx, long N, int d, const T* P0, const T* Fs, const T* Qs, const T* H, T R,
                    const T* ys, T* fms, T* fPs, T* sms, T* sPs, double* ll) {
    RoctxRange range_("parallel_filter");
    if (!ctx || N < 1 || !P0 || !Fs || !Qs || !H || !ys || !fms || !fPs || !sms || !sPs) return PGPS_E_INVALID;
    if (d < 1 || d > PGPS_MAX_DIM) return PGPS_E_UNSUPPORTED_DIM;
    if (!aligned16(Fs) || !aligned16(Qs) || !aligned16(fms) || !aligned16(fPs) || !aligned16(sms) ||
        !aligned16(sPs))
        return PGPS_E_INVALID;
    ScanArgs<T> a{};
    a.N = N; a.seg_first = 1; a.seg_last = 1;
    a.P0 = P0; a.H = H; a.R = R; a.Fs = Fs; a.Qs = Qs; a.ys = ys;
    a.fms = fms; a.fPs = fPs; a.sms = sms; a.sPs = sPs; a.ll = ll;
    if constexpr (sizeof(T) == 4) return f32_smoother_call(ctx, d, a, MODE_PKFS);
    else return dispatch_scan<T>(ctx, d, a, MODE_PKFS);
}

template <typename T>
static int disc_dev(pgps_ctx* ctx, long N, int d, const T* F, const T* Pinf, const T* ts, T t0, T* Fs, T* Qs) {
    RoctxRange range_("make_model");
    if (!ctx || N < 1 || !F || !Pinf || !ts || !Fs || !Qs) return PGPS_E_INVALID;
    if constexpr (sizeof(T) == 8) {
        const bool rc_ok = d >= rc::kDimMin && d <= rc::kDimMax;
        if (rc_ok && (ctx->family == 3 || ((ctx->family == 0 || ctx->family == 4) && d > PGPS_MAX_DIM_LANE)))
            return launch_disc_rc(ctx, N, d, F, Pinf, ts, t0, Fs, Qs);
    } else {
        if ((((ctx->family == 0 || ctx->family == 4) && d > PGPS_MAX_DIM_LANE) || ctx->family == 3) && d >= rc::kDimMin &&
            d <= rc::kDimMax) {
            // fp32 at 7 <= d <= 16 (or with the row-cooperative family forced): the arithmetic is fp64 in every
            // discretisation kernel anyway; widen the inputs,
            // run the row-cooperative kernel, narrow the results
            const size_t n = (size_t)N, dd = (size_t)d * d;
            int rc_ = ensure(ctx, ctx->lti[7], (2 * dd + n + 2 * n * dd) * sizeof(double));
            if (rc_) return rc_;
            double* base = (double*)ctx->lti[7].p;
            double *F64 = base, *P64 = F64 + dd, *t64 = P64 + dd, *Fs64 = t64 + n, *Qs64 = Fs64 + n * dd;
            auto grid = [](size_t m) { return dim3((unsigned)std::min<size_t>(4096, (m + 255) / 256)); };
            hipLaunchKernelGGL(pgps::k_widen, grid(dd), dim3(256), 0, ctx->stream, (long)dd, (const float*)F, F64);
            hipLaunchKernelGGL(pgps::k_widen, grid(dd), dim3(256), 0, ctx->stream, (long)dd, (const float*)Pinf, P64);
            hipLaunchKernelGGL(pgps::k_widen, grid(n), dim3(256), 0, ctx->stream, (long)n, (const float*)ts, t64);
            rc_ = launch_disc_rc(ctx, N, d, F64, P64, t64, (double)t0, Fs64, Qs64);
            if (rc_) return rc_;
            hipLaunchKernelGGL(pgps::k_narrow, grid(n * dd), dim3(256), 0, ctx->stream, (long)(n * dd), (const double*)Fs64, (float*)Fs);
            hipLaunchKernelGGL(pgps::k_narrow, grid(n * dd), dim3(256), 0, ctx->stream, (long)(n * dd), (const double*)Qs64, (float*)Qs);
            HIPCHK(ctx, hipGetLastError());
            return PGPS_OK;
        }
    }
    if (ctx->family == 3) return PGPS_E_UNSUPPORTED_DIM;
    if (ctx->family == 2 || (ctx->family == 0 && d > PGPS_MAX_DIM_LANE)) return launch_disc_wc<T>(ctx, N, d, F, Pinf, ts, t0, Fs, Qs);
    switch (d) {
        case 1: return launch_disc<T, 1>(ctx, N, F, Pinf, ts, t0, Fs, Qs);
        case 2: return launch_disc<T, 2>(ctx, N, F, Pinf, ts, t0, Fs, Qs);
        case 3: return launch_disc<T, 3>(ctx, N, F, Pinf, ts, t0, Fs, Qs);
        case 4: return launch_disc<T, 4>(ctx, N, F, Pinf, ts, t0, Fs, Qs);
        case 5: return launch_disc<T, 5>(ctx, N, F, Pinf, ts, t0, Fs, Qs);
        case 6: return launch_disc<T, 6>(ctx, N, F, Pinf, ts, t0, Fs, Qs);
        default: return PGPS_E_UNSUPPORTED_DIM;
    }
}

// ---------------------------------------------------------------------------------------------
// host-pointer entry points (stage -> run -> copy back)
// ---------------------------------------------------------------------------------------------
template <typename T>
static int stage_in(pgps_ctx* ctx, DevBuf& b, const T* host, size_t n, T** dev) {
    int rc = ensure(ctx, b, n * sizeof(T));
    if (rc) return rc;
    *dev = (T*)b.p;
    if (host) HIPCHK(ctx, hipMemcpyAsync(b.p, host, n * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    return PGPS_OK;
}
template <typename T>
static int stage_out(pgps_ctx* ctx, T* host, const T* dev, size_t n) {
    if (host) HIPCHK(ctx, hipMemcpyAsync(host, dev, n * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
    return PGPS_OK;
}

#define TRY(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)

// A call whose host arrays are small -- the first evaluation of a model at the reference's series lengths, the whole of its
// speed protocol (experiments/toy_models/speed_and_stability.py:73-87) -- pays for every hipMemcpy from pageable memory (a
// staged, host-synchronous copy of ~10 us whatever its size).  SmallStage lays the call's inputs out in one pinned arena
// (plain memcpy), sends them with ONE asynchronous copy, and brings the outputs back with ONE: three inputs and three
// outputs of pgps_gp_predict_f64 at N = K = 4096 were six such copies (model + predict_f 178 -> 140 us).
constexpr size_t kPinArena = 2u << 20;
struct SmallStage {
    pgps_ctx* ctx;
    size_t in_bytes = 0, out_bytes = 0, in_cap, out_off;
    bool ok = false;
    bool in_flight = false;         // send() has queued a copy out of the pinned arena and finish() has not synchronised yet
    struct Out { void* host; size_t off, bytes; } outs[4];
    int nout = 0;
    // in_total / out_total: bytes of all inputs / outputs (each rounded up to 16)
    SmallStage(pgps_ctx* c, size_t in_total, size_t out_total) : ctx(c), in_cap(in_total), out_off(in_total) {
        if (in_total + out_total > kPinArena) return;
        if (!ctx->pin_h && hipHostMalloc((void**)&ctx->pin_h, kPinArena, hipHostMallocDefault) != hipSuccess) { ctx->pin_h = nullptr; return; }
        if (ensure(ctx, ctx->pin_d, kPinArena) != PGPS_OK) return;
        ok = true;
    }
    // an error return between send() and finish() (TRY leaves the function) must not leave the arena's host-to-device copy
    // in flight: the next small call would memcpy its inputs into the same pinned bytes underneath it
    ~SmallStage() {
        if (in_flight) (void)hipStreamSynchronize(ctx->stream);
    }
    SmallStage(const SmallStage&) = delete;
    SmallStage& operator=(const SmallStage&) = delete;
    static size_t up(size_t b) { return (b + 15) / 16 * 16; }
    template <typename T> T* in(const T* host, size_t n) {          // -> device pointer of the staged copy
        T* dev = (T*)((char*)ctx->pin_d.p + in_bytes);
        memcpy(ctx->pin_h + in_bytes, host, n * sizeof(T));
        in_bytes += up(n * sizeof(T));
        return dev;
    }
    template <typename T> T* out(T* host, size_t n) {               // -> device pointer the call writes, copied back by finish()
        T* dev = (T*)((char*)ctx->pin_d.p + out_off + out_bytes);
        outs[nout++] = {(void*)host, out_off + out_bytes, n * sizeof(T)};
        out_bytes += up(n * sizeof(T));
        return dev;
    }
    int send() {
        in_flight = true;
        HIPCHK(ctx, hipMemcpyAsync(ctx->pin_d.p, ctx->pin_h, in_bytes, hipMemcpyHostToDevice, ctx->stream));
        return PGPS_OK;
    }
    int finish() {                                                  // one copy back, the synchronisation, the scatter
        HIPCHK(ctx, hipMemcpyAsync(ctx->pin_h + out_off, (char*)ctx->pin_d.p + out_off, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        in_flight = false;
        for (int i = 0; i < nout; ++i)
            if (outs[i].host) memcpy(outs[i].host, ctx->pin_h + outs[i].off, outs[i].bytes);
        return PGPS_OK;
    }
};

template <typename T>
static int pkf_host(pgps_ctx* ctx, long N, int d, const T* P0, const T* Fs, const T* Qs, const T* H, T R,
                    const T* ys, T* fms, T* fPs, double* ll) {
    if (!ctx || N < 1 || !P0 || !Fs || !Qs || !H || !ys || !fms || !fPs) return PGPS_E_INVALID;
    if (d < 1 || d > PGPS_MAX_DIM) return PGPS_E_UNSUPPORTED_DIM;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)N, dd = (size_t)d * d;
    T *dP0, *dFs, *dQs, *dH, *dys, *dfms, *dfPs;
    double* dll;
    TRY(stage_in(ctx, ctx->st[0], P0, dd, &dP0));
    TRY(stage_in(ctx, ctx->st[1], Fs, n * dd, &dFs));
    TRY(stage_in(ctx, ctx->st[2], Qs, n * dd, &dQs));
    TRY(stage_in(ctx, ctx->st[3], H, (size_t)d, &dH));
    TRY(stage_in(ctx, ctx->st[4], ys, n, &dys));
    TRY(stage_in<T>(ctx, ctx->st[5], nullptr, n * d, &dfms));
    TRY(stage_in<T>(ctx, ctx->st[6], nullptr, n * dd, &dfPs));
    TRY(stage_in<double>(ctx, ctx->st[9], nullptr, 2, &dll));
    TRY(pkf_dev<T>(ctx, N, d, dP0, dFs, dQs, dH, R, dys, dfms, dfPs, ll ? dll : nullptr));
    TRY(stage_out(ctx, fms, dfms, n * d));
    TRY(stage_out(ctx, fPs, dfPs, n * dd));
    TRY(stage_out(ctx, ll, dll, 1));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (ll && !std::isfinite(*ll)) return PGPS_E_NUMERIC;
    return PGPS_OK;
}

template <typename T>
static int pks_host(pgps_ctx* ctx, long N, int d, const T* Fs, const T* Qs, const T* fms, const T* fPs, T* sms,
                    T* sPs) {
    if (!ctx || N < 1 || !Fs || !Qs || !fms || !fPs || !sms || !sPs) return PGPS_E_INVALID;
    if (d < 1 || d > PGPS_MAX_DIM) return PGPS_E_UNSUPPORTED_DIM;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)N, dd = (size_t)d * d;
    T *dFs, *dQs, *dfms, *dfPs, *dsms, *dsPs;
    TRY(stage_in(ctx, ctx->st[1], Fs, n * dd, &dFs));
    TRY(stage_in(ctx, ctx->st[2], Qs, n * dd, &dQs));
    TRY(stage_in(ctx, ctx->st[5], fms, n * d, &dfms));
    TRY(stage_in(ctx, ctx->st[6], fPs, n * dd, &dfPs));
    TRY(stage_in<T>(ctx, ctx->st[7], nullptr, n * d, &dsms));
    TRY(stage_in<T>(ctx, ctx->st[8], nullptr, n * dd, &dsPs));
    TRY(pks_dev<T>(ctx, N, d, dFs, dQs, dfms, dfPs, dsms, dsPs));
    TRY(stage_out(ctx, sms, dsms, n * d));
    TRY(stage_out(ctx, sPs, dsPs, n * dd));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return PGPS_OK;
}

template <typename T>
static int pkfs_host(pgps_ctx* ctx, long N, int d, const T* P0, const T* Fs, const T* Qs, const T* H, T R,
                     const T* ys, T* fms, T* fPs, T* sms, T* sPs, double* ll) {
    if (!ctx || N < 1 || !P0 || !Fs || !Qs || !H || !ys || !sms || !sPs) return PGPS_E_INVALID;
    if (d < 1 || d > PGPS_MAX_DIM) return PGPS_E_UNSUPPORTED_DIM;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)N, dd = (size_t)d * d;
    T *dP0, *dFs, *dQs, *dH, *dys, *dfms, *dfPs, *dsms, *dsPs;
    double* dll;
    TRY(stage_in(ctx, ctx->st[0], P0, dd, &dP0));
    TRY(stage_in(ctx, ctx->st[1], Fs, n * dd, &dFs));
    TRY(stage_in(ctx, ctx->st[2], Qs, n * dd, &dQs));
    TRY(stage_in(ctx, ctx->st[3], H, (size_t)d, &dH));
    TRY(stage_in(ctx, ctx->st[4], ys, n, &dys));
    TRY(stage_in<T>(ctx, ctx->st[5], nullptr, n * d, &dfms));
    TRY(stage_in<T>(ctx, ctx->st[6], nullptr, n * dd, &dfPs));
    TRY(stage_in<T>(ctx, ctx->st[7], nullptr, n * d, &dsms));
    TRY(stage_in<T>(ctx, ctx->st[8], nullptr, n * dd, &dsPs));
    TRY(stage_in<double>(ctx, ctx->st[9], nullptr, 2, &dll));
    TRY(pkfs_dev<T>(ctx, N, d, dP0, dFs, dQs, dH, R, dys, dfms, dfPs, dsms, dsPs, dll));
    TRY(stage_out(ctx, fms, dfms, n * d));
    TRY(stage_out(ctx, fPs, dfPs, n * dd));
    TRY(stage_out(ctx, sms, dsms, n * d));
    TRY(stage_out(ctx, sPs, dsPs, n * dd));
    double llh = 0.0;
    TRY(stage_out(ctx, &llh, dll, 1));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (ll) *ll = llh;
    if (!std::isfinite(llh)) return PGPS_E_NUMERIC;
    return PGPS_OK;
}

template <typename T>
static int disc_host(pgps_ctx* ctx, long N, int d, const T* F, const T* Pinf, const T* ts, T t0, T* Fs, T* Qs) {
    if (!ctx || N < 1 || !F || !Pinf || !ts || !Fs || !Qs) return PGPS_E_INVALID;
    if (d < 1 || d > PGPS_MAX_DIM) return PGPS_E_UNSUPPORTED_DIM;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)N, dd = (size_t)d * d;
    T *dF, *dP, *dts, *dFs, *dQs;
    TRY(stage_in(ctx, ctx->st[0], F, dd, &dF));
    TRY(stage_in(ctx, ctx->st[3], Pinf, dd, &dP));
    TRY(stage_in(ctx, ctx->st[4], ts, n, &dts));
    TRY(stage_in<T>(ctx, ctx->st[1], nullptr, n * dd, &dFs));
    TRY(stage_in<T>(ctx, ctx->st[2], nullptr, n * dd, &dQs));
    TRY(disc_dev<T>(ctx, N, d, dF, dP, dts, t0, dFs, dQs));
    TRY(stage_out(ctx, Fs, dFs, n * dd));
    TRY(stage_out(ctx, Qs, dQs, n * dd));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return PGPS_OK;
}

// ---------------------------------------------------------------------------------------------
// extern "C" surface
// ---------------------------------------------------------------------------------------------
#define PGPS_DEFINE(SUF, T)                                                                                          \
    extern "C" int pgps_discretise_##SUF(pgps_ctx* c, long N, int d, const T* F, const T* P, const T* ts, T t0,     \
                                         T* Fs, T* Qs) {                                                            \
        return disc_host<T>(c, N, d, F, P, ts, t0, Fs, Qs);                                                         \
    }                                                                                                                \
    extern "C" int pgps_discretise_dev_##SUF(pgps_ctx* c, long N, int d, const T* F, const T* P, const T* ts, T t0, \
                                             T* Fs, T* Qs) {                                                        \
        if (d < 1 || d > PGPS_MAX_DIM) return PGPS_E_UNSUPPORTED_DIM;                                               \
        return disc_dev<T>(c, N, d, F, P, ts, t0, Fs, Qs);                                                          \
    }                                                                                                                \
    extern "C" int pgps_pkf_##SUF(pgps_ctx* c, long N, int d, const T* P0, const T* Fs, const T* Qs, const T* H,    \
                                  T R, const T* ys, T* fms, T* fPs, double* ll) {                                   \
        return pkf_host<T>(c, N, d, P0, Fs, Qs, H, R, ys, fms, fPs, ll);                                            \
    }                                                                                                                \
    extern "C" int pgps_pkf_dev_##SUF(pgps_ctx* c, long N, int d, const T* P0, const T* Fs, const T* Qs,            \
                                      const T* H, T R, const T* ys, T* fms, T* fPs, double* ll) {                   \
        return pkf_dev<T>(c, N, d, P0, Fs, Qs, H, R, ys, fms, fPs, ll);                                             \
    }                                                                                                                \
    extern "C" int pgps_pks_##SUF(pgps_ctx* c, long N, int d, const T* Fs, const T* Qs, const T* fms,               \
                                  const T* fPs, T* sms, T* sPs) {                                                   \
        return pks_host<T>(c, N, d, Fs, Qs, fms, fPs, sms, sPs);                                                    \
    }                                                                                                                \
    extern "C" int pgps_pks_dev_##SUF(pgps_ctx* c, long N, int d, const T* Fs, const T* Qs, const T* fms,           \
                                      const T* fPs, T* sms, T* sPs) {                                               \
        return pks_dev<T>(c, N, d, Fs, Qs, fms, fPs, sms, sPs);                                                     \
    }                                                                                                                \
    extern "C" int pgps_pkfs_##SUF(pgps_ctx* c, long N, int d, const T* P0, const T* Fs, const T* Qs, const T* H,   \
                                   T R, const T* ys, T* fms, T* fPs, T* sms, T* sPs, double* ll) {                  \
        return pkfs_host<T>(c, N, d, P0, Fs, Qs, H, R, ys, fms, fPs, sms, sPs, ll);                                 \
    }                                                                                                                \
    extern "C" int pgps_pkfs_dev_##SUF(pgps_ctx* c, long N, int d, const T* P0, const T* Fs, const T* Qs,           \
                                       const T* H, T R, const T* ys, T* fms, T* fPs, T* sms, T* sPs, double* ll) {  \
        return pkfs_dev<T>(c, N, d, P0, Fs, Qs, H, R, ys, fms, fPs, sms, sPs, ll);                                  \
    }

PGPS_DEFINE(f64, double)
PGPS_DEFINE(f32, float)

// ---------------------------------------------------------------------------------------------
// segment (multi-GPU) entry points
// ---------------------------------------------------------------------------------------------
extern "C" int pgps_seg_record_len(int d, int* rec_filter, int* rec_smoother) {
    if (d < 1 || !rec_filter || !rec_smoother) return PGPS_E_INVALID;
    *rec_filter = seg_rec_f_len(d);
    *rec_smoother = seg_rec_s_len(d);
    return PGPS_OK;
}

template <typename T>
static int seg_common(pgps_ctx* ctx, long N, int d, int rank, int nranks, ScanArgs<T>& a) {
    if (!ctx || N < 1 || rank < 0 || nranks < 1 || rank >= nranks) return PGPS_E_INVALID;
    if (d < 1 || d > PGPS_MAX_DIM) return PGPS_E_UNSUPPORTED_DIM;
    a.N = N;
    a.rank = rank;
    a.nranks = nranks;
    return PGPS_OK;
}

// Phase `phase` (2, 3) may only follow phase - 1 of the same pass with nothing else on the context in between: the
// scratch it reads (chain totals, their scans, stored smoothing elements) is whatever the last call left in `ws`.
static bool seg_follows(const pgps_ctx* ctx, int phase, long N, int d, int rank, int nranks) {
    const auto& t = ctx->seg_tag;
    return t.phase == phase - 1 && t.N == N && t.d == d && t.rank == rank && t.nranks == nranks && t.chunk == ctx->chunk && t.block == ctx->block &&
           t.family == ctx->family && t.stage_g == ctx->stage_g && t.dma == ctx->dma && t.rc_scan == ctx->rc_scan && t.epoch == ctx->ws_epoch;
}
static void seg_mark(pgps_ctx* ctx, int phase, long N, int d, int rank, int nranks) {
    ctx->seg_tag.phase = phase; ctx->seg_tag.N = N; ctx->seg_tag.d = d; ctx->seg_tag.rank = rank;
    ctx->seg_tag.nranks = nranks; ctx->seg_tag.chunk = ctx->chunk; ctx->seg_tag.block = ctx->block; ctx->seg_tag.family = ctx->family;
    ctx->seg_tag.stage_g = ctx->stage_g; ctx->seg_tag.dma = ctx->dma; ctx->seg_tag.rc_scan = ctx->rc_scan; ctx->seg_tag.epoch = ctx->ws_epoch;
}

template <typename T>
static int seg_reduce(pgps_ctx* ctx, long N, int d, int rank, int nranks, const T* P0, const T* Fs, const T* Qs,
                      const T* H, T R, const T* ys, T* rec_f) {
    ScanArgs<T> a{};
    TRY(seg_common<T>(ctx, N, d, rank, nranks, a));
    if (!P0 || !Fs || !Qs || !H || !ys || !rec_f || !aligned16(Fs) || !aligned16(Qs)) return PGPS_E_INVALID;
    a.P0 = P0; a.H = H; a.R = R; a.Fs = Fs; a.Qs = Qs; a.ys = ys; a.rec_f = rec_f;
    ctx->seg_tag.phase = 0;
    TRY(dispatch_scan<T>(ctx, d, a, MODE_SEG_REDUCE));
    seg_mark(ctx, 1, N, d, rank, nranks);
    return PGPS_OK;
}

template <typename T>
static int seg_filter(pgps_ctx* ctx, long N, int d, int rank, int nranks, const T* P0, const T* Fs, const T* Qs,
                      const T* H, T R, const T* ys, const T* gathered_f, T* fms, T* fPs, T* rec_s) {
    ScanArgs<T> a{};
    TRY(seg_common<T>(ctx, N, d, rank, nranks, a));
    if (!P0 || !Fs || !Qs || !H || !ys || !gathered_f || !fms || !fPs || !rec_s) return PGPS_E_INVALID;
    if (!aligned16(Fs) || !aligned16(Qs) || !aligned16(fms) || !aligned16(fPs) || !aligned16(rec_s))
        return PGPS_E_INVALID;
    a.P0 = P0; a.H = H; a.R = R; a.Fs = Fs; a.Qs = Qs; a.ys = ys;
    a.gathered_f = gathered_f; a.fms = fms; a.fPs = fPs; a.rec_s = rec_s;
    if (!seg_follows(ctx, 2, N, d, rank, nranks)) return PGPS_E_INVALID;
    ctx->seg_tag.phase = 0;
    TRY(dispatch_scan<T>(ctx, d, a, MODE_SEG_FILTER));
    seg_mark(ctx, 2, N, d, rank, nranks);
    return PGPS_OK;
}

template <typename T>
static int seg_smoother(pgps_ctx* ctx, long N, int d, int rank, int nranks, const T* Fs, const T* Qs, const T* fms,
                        const T* fPs, const T* gathered_s, T* sms, T* sPs, double* ll) {
    ScanArgs<T> a{};
    TRY(seg_common<T>(ctx, N, d, rank, nranks, a));
    if (!Fs || !Qs || !fms || !fPs || !gathered_s || !sms || !sPs) return PGPS_E_INVALID;
    if (!aligned16(Fs) || !aligned16(Qs) || !aligned16(fms) || !aligned16(fPs) || !aligned16(sms) ||
        !aligned16(sPs) || !aligned16(gathered_s))
        return PGPS_E_INVALID;
    a.Fs = Fs; a.Qs = Qs; a.fms = const_cast<T*>(fms); a.fPs = const_cast<T*>(fPs);
    a.gathered_s = gathered_s; a.sms = sms; a.sPs = sPs; a.ll = ll;
    if (!seg_follows(ctx, 3, N, d, rank, nranks)) return PGPS_E_INVALID;
    ctx->seg_tag.phase = 0;
    return dispatch_scan<T>(ctx, d, a, MODE_SEG_SMOOTHER);
}

#define PGPS_DEFINE_SEG(SUF, T)                                                                                      \
    extern "C" int pgps_seg_filter_reduce_dev_##SUF(pgps_ctx* c, long N, int d, int rank, int nranks, const T* P0,   \
                                                    const T* Fs, const T* Qs, const T* H, T R, const T* ys,         \
                                                    T* rec_f) {                                                     \
        return seg_reduce<T>(c, N, d, rank, nranks, P0, Fs, Qs, H, R, ys, rec_f);                                   \
    }                                                                                                                \
    extern "C" int pgps_seg_filter_apply_dev_##SUF(pgps_ctx* c, long N, int d, int rank, int nranks, const T* P0,    \
                                                   const T* Fs, const T* Qs, const T* H, T R, const T* ys,          \
                                                   const T* gathered_f, T* fms, T* fPs, T* rec_s) {                 \
        return seg_filter<T>(c, N, d, rank, nranks, P0, Fs, Qs, H, R, ys, gathered_f, fms, fPs, rec_s);             \
    }                                                                                                                \
    extern "C" int pgps_seg_smoother_apply_dev_##SUF(pgps_ctx* c, long N, int d, int rank, int nranks, const T* Fs,  \
                                                     const T* Qs, const T* fms, const T* fPs, const T* gathered_s,  \
                                                     T* sms, T* sPs, double* ll) {                                  \
        return seg_smoother<T>(c, N, d, rank, nranks, Fs, Qs, fms, fPs, gathered_s, sms, sPs, ll);                  \
    }

PGPS_DEFINE_SEG(f64, double)
PGPS_DEFINE_SEG(f32, float)

// One call per pass: reduce -> all-gather -> filter -> all-gather -> smoother, all enqueued on the context's stream
// through the context's own RCCL communicator -- no host round trip, no framework in between.
template <typename T>
static int pkfs_seg_run(pgps_ctx* ctx, long N, int d, const T* P0, const T* Fs, const T* Qs, const T* H, T R, const T* ys,
                        T* fms, T* fPs, T* sms, T* sPs, double* ll) {
    RoctxRange range_("parallel_filter");
    const int rank = ctx->comm_rank, nranks = ctx->comm_nranks;
    const size_t rf = ((size_t)seg_rec_f_len(d) * sizeof(T) + 15) / 16 * 16, rs = ((size_t)seg_rec_s_len(d) * sizeof(T) + 15) / 16 * 16;
    TRY(ensure(ctx, ctx->comm_buf, (rf + rs) * (size_t)(nranks + 1)));
    char* base = (char*)ctx->comm_buf.p;
    T* rec_f = (T*)base;
    T* rec_s = (T*)(base + rf);
    T* gat_f = (T*)(base + rf + rs);
    T* gat_s = (T*)(base + rf + rs + rf * (size_t)nranks);
    // records travel at their natural length (the ranks' slots in gathered_* are seg_rec_*_len(d) apart)
    TRY(seg_reduce<T>(ctx, N, d, rank, nranks, P0, Fs, Qs, H, R, ys, rec_f));
    TRY(comm_allgather(ctx, rec_f, gat_f, (size_t)seg_rec_f_len(d) * sizeof(T)));
    TRY(seg_filter<T>(ctx, N, d, rank, nranks, P0, Fs, Qs, H, R, ys, gat_f, fms, fPs, rec_s));
    TRY(comm_allgather(ctx, rec_s, gat_s, (size_t)seg_rec_s_len(d) * sizeof(T)));
    return seg_smoother<T>(ctx, N, d, rank, nranks, Fs, Qs, fms, fPs, gat_s, sms, sPs, ll);
}

template <typename T>
static int pkfs_seg_dev(pgps_ctx* ctx, long N, int d, const T* P0, const T* Fs, const T* Qs, const T* H, T R, const T* ys,
                        T* fms, T* fPs, T* sms, T* sPs, double* ll) {
    if (!ctx || N < 1 || !P0 || !Fs || !Qs || !H || !ys || !fms || !fPs || !sms || !sPs) return PGPS_E_INVALID;
    if (!ctx->comm) return PGPS_E_INVALID;                      // pgps_comm_init first (also for one rank)
    if (d < 1 || d > PGPS_MAX_DIM) return PGPS_E_UNSUPPORTED_DIM;
    return pkfs_seg_run<T>(ctx, N, d, P0, Fs, Qs, H, R, ys, fms, fPs, sms, sPs, ll);
}

extern "C" int pgps_pkfs_seg_dev_f64(pgps_ctx* c, long N, int d, const double* P0, const double* Fs, const double* Qs,
                                     const double* H, double R, const double* ys, double* fms, double* fPs, double* sms,
                                     double* sPs, double* ll) {
    return pkfs_seg_dev<double>(c, N, d, P0, Fs, Qs, H, R, ys, fms, fPs, sms, sPs, ll);
}
extern "C" int pgps_pkfs_seg_dev_f32(pgps_ctx* c, long N, int d, const float* P0, const float* Fs, const float* Qs,
                                     const float* H, float R, const float* ys, float* fms, float* fPs, float* sms, float* sPs,
                                     double* ll) {
    return pkfs_seg_dev<float>(c, N, d, P0, Fs, Qs, H, R, ys, fms, fPs, sms, sPs, ll);
}

// ---------------------------------------------------------------------------------------------
// fused-discretisation ("gp") entry points
// ---------------------------------------------------------------------------------------------
template <typename T>
static int gp_dev(pgps_ctx* ctx, long N, int d, double lam, const double* N1, const double* N2, const double* Pinf,
                  const double* H, double R, const T* ts, double t0, const T* ys, T* fms, T* fPs, T* sms, T* sPs,
                  double* ll) {
    RoctxRange range_("parallel_filter");
    if (!ctx || N < 1 || !N1 || !Pinf || !H || !ts || !ys) return PGPS_E_INVALID;
    if (d < 1 || d > 3) return PGPS_E_UNSUPPORTED_DIM;
    if ((fms == nullptr) != (fPs == nullptr) || (sms == nullptr) != (sPs == nullptr)) return PGPS_E_INVALID;
    if (sms && !fms) return PGPS_E_INVALID;             // the smoother reads the filtered moments back
    if (!fms && !ll) return PGPS_E_INVALID;
    if ((fms && (!aligned16(fms) || !aligned16(fPs))) || (sms && (!aligned16(sms) || !aligned16(sPs))))
        return PGPS_E_INVALID;
    GpArgs<T> g{};
    g.s.N = N;
    g.s.R = (T)R;
    g.s.ys = ys;
    g.s.fms = fms; g.s.fPs = fPs; g.s.sms = sms; g.s.sPs = sPs; g.s.ll = ll;
    g.m.lam = lam;
    for (int i = 0; i < 9; ++i) { g.m.N1[i] = 0; g.m.N2[i] = 0; g.m.Pinf[i] = 0; }
    for (int i = 0; i < d * d; ++i) { g.m.N1[i] = N1[i]; g.m.N2[i] = N2 ? N2[i] : 0.0; g.m.Pinf[i] = Pinf[i]; }
    for (int i = 0; i < 3; ++i) g.m.H[i] = i < d ? (T)H[i] : T(0);
    g.m.ts = ts;
    g.m.t_prev = (T)t0;
    if constexpr (sizeof(T) == 8) {
        if (resident_fits(ctx, N, d, false) && aligned16(ts) && aligned16(ys)) {
            ResArgs<double> ra{};
            ra.s = g.s;
            ra.m = g.m;
            return launch_resident<double, 2>(ctx, ra, true, sms != nullptr);
        }
    }
    switch (d) {
        case 1: return launch_gp<T, 1>(ctx, g, fms != nullptr, sms != nullptr);
        case 2: return launch_gp<T, 2>(ctx, g, fms != nullptr, sms != nullptr);
        case 3: return launch_gp<T, 3>(ctx, g, fms != nullptr, sms != nullptr);
        default: return PGPS_E_UNSUPPORTED_DIM;
    }
}

// ll and the model's adjoints on the fused path: out = [ll | Abar (d d) | Ubar (d) | Hbar (d) | Rbar] on the device
static int gp_adj_dev(pgps_ctx* ctx, long N, int d, double lam, const double* N1, const double* N2, const double* Pinf,
                      const double* H, double R, const double* ts, double t0, const double* ys, double* out) {
    RoctxRange range_("parallel_filter");
    if (!ctx || N < 1 || !N1 || !Pinf || !H || !ts || !ys || !out) return PGPS_E_INVALID;
    if (d < 1 || d > 3) return PGPS_E_UNSUPPORTED_DIM;
    GpArgs<double> g{};
    g.s.N = N;
    g.s.R = R;
    g.s.ys = ys;
    g.m.lam = lam;
    for (int i = 0; i < 9; ++i) { g.m.N1[i] = 0; g.m.N2[i] = 0; g.m.Pinf[i] = 0; }
    for (int i = 0; i < d * d; ++i) { g.m.N1[i] = N1[i]; g.m.N2[i] = N2 ? N2[i] : 0.0; g.m.Pinf[i] = Pinf[i]; }
    for (int i = 0; i < 3; ++i) g.m.H[i] = i < d ? H[i] : 0.0;
    g.m.ts = ts;
    g.m.t_prev = t0;
    switch (d) {
        case 1: return launch_gp_adj<double, 1>(ctx, g, out);
        case 2: return launch_gp_adj<double, 2>(ctx, g, out);
        case 3: return launch_gp_adj<double, 3>(ctx, g, out);
        default: return PGPS_E_UNSUPPORTED_DIM;
    }
}

extern "C" int pgps_gp_ll_grad_adj_dev_f64(pgps_ctx* ctx, long N, int d, double lam, const double* N1, const double* N2,
                                           const double* Pinf, const double* H, double R, const double* ts, double t0,
                                           const double* ys, double* out) {
    return gp_adj_dev(ctx, N, d, lam, N1, N2, Pinf, H, R, ts, t0, ys, out);
}

template <typename T>
static int gp_host(pgps_ctx* ctx, long N, int d, double lam, const double* N1, const double* N2, const double* Pinf,
                   const double* H, double R, const T* ts, double t0, const T* ys, T* fms, T* fPs, T* sms, T* sPs,
                   double* ll) {
    if (!ctx || N < 1 || !ts || !ys) return PGPS_E_INVALID;
    if (d < 1 || d > 3) return PGPS_E_UNSUPPORTED_DIM;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)N, dd = (size_t)d * d;
    const bool wf = fms || fPs || sms || sPs, wsm = sms || sPs;
    if (!wf) {                  // log-likelihood only (a model's first objective): the small-call road
        SmallStage st(ctx, 2 * SmallStage::up(n * sizeof(T)), 16);
        if (st.ok) {
            double llh = 0.0;
            T* dys_ = st.in(ys, n);
            T* dts_ = st.in(ts, n);
            double* dll_ = st.out(&llh, 1);
            TRY(st.send());
            TRY(gp_dev<T>(ctx, N, d, lam, N1, N2, Pinf, H, R, dts_, t0, dys_, nullptr, nullptr, nullptr, nullptr, dll_));
            TRY(st.finish());
            if (ll) *ll = llh;
            return std::isfinite(llh) ? PGPS_OK : PGPS_E_NUMERIC;
        }
    }
    T *dts, *dys, *dfms = nullptr, *dfPs = nullptr, *dsms = nullptr, *dsPs = nullptr;
    double* dll;
    TRY(stage_in(ctx, ctx->st[4], ys, n, &dys));
    TRY(stage_in(ctx, ctx->st[10], ts, n, &dts));
    if (wf) {
        TRY(stage_in<T>(ctx, ctx->st[5], nullptr, n * d, &dfms));
        TRY(stage_in<T>(ctx, ctx->st[6], nullptr, n * dd, &dfPs));
    }
    if (wsm) {
        TRY(stage_in<T>(ctx, ctx->st[7], nullptr, n * d, &dsms));
        TRY(stage_in<T>(ctx, ctx->st[8], nullptr, n * dd, &dsPs));
    }
    TRY(stage_in<double>(ctx, ctx->st[9], nullptr, 2, &dll));
    TRY(gp_dev<T>(ctx, N, d, lam, N1, N2, Pinf, H, R, dts, t0, dys, dfms, dfPs, dsms, dsPs, dll));
    if (fms) TRY(stage_out(ctx, fms, dfms, n * d));
    if (fPs) TRY(stage_out(ctx, fPs, dfPs, n * dd));
    if (sms) TRY(stage_out(ctx, sms, dsms, n * d));
    if (sPs) TRY(stage_out(ctx, sPs, dsPs, n * dd));
    double llh = 0.0;
    TRY(stage_out(ctx, &llh, dll, 1));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (ll) *ll = llh;
    if (!std::isfinite(llh)) return PGPS_E_NUMERIC;
    return PGPS_OK;
}

#define PGPS_DEFINE_GP(SUF, T)                                                                                       \
    extern "C" int pgps_gp_dev_##SUF(pgps_ctx* c, long N, int d, double lam, const double* N1, const double* N2,     \
                                     const double* Pinf, const double* H, double R, const T* ts, double t0,         \
                                     const T* ys, T* fms, T* fPs, T* sms, T* sPs, double* ll) {                     \
        return gp_dev<T>(c, N, d, lam, N1, N2, Pinf, H, R, ts, t0, ys, fms, fPs, sms, sPs, ll);                     \
    }                                                                                                                \
    extern "C" int pgps_gp_##SUF(pgps_ctx* c, long N, int d, double lam, const double* N1, const double* N2,         \
                                 const double* Pinf, const double* H, double R, const T* ts, double t0, const T* ys, \
                                 T* fms, T* fPs, T* sms, T* sPs, double* ll) {                                      \
        return gp_host<T>(c, N, d, lam, N1, N2, Pinf, H, R, ts, t0, ys, fms, fPs, sms, sPs, ll);                    \
    }

PGPS_DEFINE_GP(f64, double)
PGPS_DEFINE_GP(f32, float)

// ---------------------------------------------------------------------------------------------
// predict_f on the device: merge of the sorted training / query times, NaN marking of the query rows,
// fused filter + smoother, projection through H at the query rows only (pssgp/model.py:15-55,92-111)
// ---------------------------------------------------------------------------------------------
namespace pgps {

// As _merge_sorted (pssgp/model.py:15-55): the shorter array is scattered into the longer one at
// arange + searchsorted(longer, shorter, side="left"), so on equal times the shorter array's point
// comes first; the training series is the "longer" one when N >= K (model.py:25 swaps only if N < K).
//
// That is a stable merge in which array A = the shorter one wins ties, done here as a tiled merge path: a workgroup owns
// kMergeTile consecutive OUTPUT positions; two of its lanes find where the tile's first and last diagonals cut A and B
// (one binary search each -- per tile, not per element), the at most kMergeTile input times (and the training
// observations that go with them) come into LDS with coalesced loads, every lane finds the cut of its own four
// outputs by a binary search in LDS and merges them serially, and the merged times, observations (NaN at query rows)
// and query slots leave through LDS as whole coalesced rows.  (Before: one 20-level binary search over global memory
// and three scattered stores per element -- 40 of the 131 us of a 2^20 + 2^18 predict_f.)
constexpr int kMergeItems = 4;
constexpr int kMergeTile = kBlock * kMergeItems;

template <typename T, int N>
__device__ __forceinline__ void store_vec(T* p, const T* v) {
    using V4 = __attribute__((ext_vector_type(4))) unsigned int;
    static_assert((N * sizeof(T)) % 16 == 0, "whole 16-byte pieces");
    V4 tmp[N * sizeof(T) / 16];
    __builtin_memcpy(tmp, v, N * sizeof(T));
#pragma unroll
    for (unsigned i = 0; i < N * sizeof(T) / 16; ++i) reinterpret_cast<V4*>(p)[i] = tmp[i];
}

// number of A's elements among the first `diag` outputs of merge(A, B) with A winning ties, found by the whole
// workgroup: a kBlock-ary search (every lane probes one candidate, the count of "goes before" answers narrows the range
// kBlock-fold) -- three dependent rounds of loads for 2^20 elements where a binary search takes twenty
template <typename T>
__device__ __forceinline__ long merge_path_block(const T* A, long nA, const T* B, long nB, long diag) {
    long lo = diag > nB ? diag - nB : 0, hi = diag < nA ? diag : nA;      // the answer lies in [lo, hi]
    while (hi > lo) {                                                      // (uniform: every lane holds the same range)
        const long step = (hi - lo + kBlock - 1) / kBlock;
        const long mid = lo + (long)threadIdx.x * step;
        const bool before = mid < hi && A[mid] <= B[diag - 1 - mid];      // monotone in mid: true ... true false ... false
        const long c = __syncthreads_count(before);
        const long nlo = c > 0 ? lo + (c - 1) * step + 1 : lo;
        const long nhi = lo + c * step < hi ? lo + c * step : hi;
        lo = nlo; hi = nhi;
    }
    return lo;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_merge_sorted(long N, long K, const T* ts, const T* ys, const T* tq, T* ts_m,
                                                          T* ys_m, int* qslot) {
    __shared__ T s_t[kMergeTile];           // A's part of the tile, then B's
    __shared__ T s_y[kMergeTile];           // observations of the training part
    const bool query_first = (N >= K);      // A = the queries when they are the shorter array
    const T* A = query_first ? tq : ts;
    const T* B = query_first ? ts : tq;
    const long nA = query_first ? K : N, nB = query_first ? N : K, M = N + K;
    const long d0 = (long)blockIdx.x * kMergeTile, d1 = d0 + kMergeTile < M ? d0 + kMergeTile : M;
    const long a0 = merge_path_block(A, nA, B, nB, d0), a1 = merge_path_block(A, nA, B, nB, d1);
    const long b0 = d0 - a0, b1 = d1 - a1;
    const int na = (int)(a1 - a0), nb = (int)(b1 - b0);
    for (int e = threadIdx.x; e < na + nb; e += kBlock) {
        const bool inA = e < na;
        const long g = inA ? a0 + e : b0 + (e - na);
        s_t[e] = inA ? A[g] : B[g];
        const bool training = (inA != query_first);
        s_y[e] = training ? ys[g] : (T)__builtin_nan("");
    }
    __syncthreads();
    // this lane's kMergeItems consecutive outputs: cut of its first diagonal inside the tile, then a serial merge
    const int n = na + nb;
    const int ld = min((int)threadIdx.x * kMergeItems, n);
    int lo = ld > nb ? ld - nb : 0, hi = ld < na ? ld : na;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (s_t[mid] <= s_t[na + (ld - 1 - mid)]) lo = mid + 1; else hi = mid;
    }
    int ia = lo, ib = ld - lo;
    T rt[kMergeItems], ry[kMergeItems];
    int rq[kMergeItems];
#pragma unroll
    for (int r = 0; r < kMergeItems; ++r) {
        rt[r] = T(0); ry[r] = T(0); rq[r] = -1;
        if (ld + r < n) {
            const bool takeA = ib >= nb || (ia < na && s_t[ia] <= s_t[na + ib]);
            const int e = takeA ? ia : na + ib;
            rt[r] = s_t[e];
            ry[r] = s_y[e];
            rq[r] = (takeA == query_first) ? (int)(takeA ? a0 + ia : b0 + ib) : -1;
            if (takeA) ++ia; else ++ib;
        }
    }
    // four consecutive outputs per lane: whole 16-byte stores (the staging buffers are 256-byte aligned, tiles whole)
    const long o = d0 + ld;
    if (ld + kMergeItems <= n) {
        store_vec<T, kMergeItems>(ts_m + o, rt);
        store_vec<T, kMergeItems>(ys_m + o, ry);
        store_vec<int, kMergeItems>(qslot + o, rq);
    } else {
        for (int r = 0; r < kMergeItems && ld + r < n; ++r) { ts_m[o + r] = rt[r]; ys_m[o + r] = ry[r]; qslot[o + r] = rq[r]; }
    }
}

template <typename T>
int launch_merge(pgps_ctx* ctx, long N, long K, const T* ts, const T* ys, const T* tq, T* ts_m, T* ys_m, int* qslot) {
    RoctxRange range_("merge_sorted");
    const long M = N + K;
    const dim3 grid((unsigned)((M + kMergeTile - 1) / kMergeTile)), block(kBlock);
    k_merge_sorted<T><<<grid, block, 0, ctx->stream>>>(N, K, ts, ys, tq, ts_m, ys_m, qslot);
    HIPCHK(ctx, hipGetLastError());
    return PGPS_OK;
}

}  // namespace pgps

template <typename T>
static int gp_predict_dev(pgps_ctx* ctx, long N, long K, int d, double lam, const double* N1, const double* N2,
                          const double* Pinf, const double* H, double R, const T* ts, const T* ys, double t0,
                          const T* tq, T* mean, T* var, double* ll) {
    if (!ctx || N < 1 || K < 1 || !N1 || !Pinf || !H || !ts || !ys || !tq || !mean || !var) return PGPS_E_INVALID;
    if (d < 1 || d > 3) return PGPS_E_UNSUPPORTED_DIM;
    if (N + K > 0x7fffffffL) return PGPS_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t m = (size_t)(N + K), dd = (size_t)d * d;
    T *ts_m, *ys_m, *fms, *fPs;
    int* qslot;
    double* dll;
    TRY(stage_in<T>(ctx, ctx->st[0], nullptr, m, &ts_m));
    TRY(stage_in<T>(ctx, ctx->st[1], nullptr, m, &ys_m));
    TRY(stage_in<int>(ctx, ctx->st[2], nullptr, m, &qslot));
    TRY(stage_in<T>(ctx, ctx->st[5], nullptr, m * d, &fms));
    TRY(stage_in<T>(ctx, ctx->st[6], nullptr, m * dd, &fPs));
    TRY(stage_in<double>(ctx, ctx->st[11], nullptr, 2, &dll));
    TRY(launch_merge<T>(ctx, N, K, ts, ys, tq, ts_m, ys_m, qslot));
    GpArgs<T> g{};
    g.s.N = (long)m;
    g.s.R = (T)R;
    g.s.ys = ys_m;
    g.s.fms = fms; g.s.fPs = fPs; g.s.sms = nullptr; g.s.sPs = nullptr;
    g.s.ll = ll ? ll : dll;
    g.m.lam = lam;
    for (int i = 0; i < 9; ++i) { g.m.N1[i] = 0; g.m.N2[i] = 0; g.m.Pinf[i] = 0; }
    for (int i = 0; i < d * d; ++i) { g.m.N1[i] = N1[i]; g.m.N2[i] = N2 ? N2[i] : 0.0; g.m.Pinf[i] = Pinf[i]; }
    for (int i = 0; i < 3; ++i) g.m.H[i] = i < d ? (T)H[i] : T(0);
    g.m.ts = ts_m;
    g.m.t_prev = (T)t0;
    g.qslot = qslot;
    g.pmean = mean;
    g.pvar = var;
    switch (d) {
        case 1: return launch_gp<T, 1>(ctx, g, 1, 1);
        case 2: return launch_gp<T, 2>(ctx, g, 1, 1);
        default: return launch_gp<T, 3>(ctx, g, 1, 1);
    }
}

template <typename T>
static int gp_predict_host(pgps_ctx* ctx, long N, long K, int d, double lam, const double* N1, const double* N2,
                           const double* Pinf, const double* H, double R, const T* ts, const T* ys, double t0,
                           const T* tq, T* mean, T* var, double* ll) {
    if (!ctx || N < 1 || K < 1 || !ts || !ys || !tq || !mean || !var) return PGPS_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    {
        SmallStage st(ctx, 2 * SmallStage::up((size_t)N * sizeof(T)) + SmallStage::up((size_t)K * sizeof(T)),
                      2 * SmallStage::up((size_t)K * sizeof(T)) + 16);
        if (st.ok) {
            double llh = 0.0;
            T* dts = st.in(ts, (size_t)N);
            T* dys = st.in(ys, (size_t)N);
            T* dtq = st.in(tq, (size_t)K);
            double* dll = st.out(&llh, 1);
            T* dmean = st.out(mean, (size_t)K);
            T* dvar = st.out(var, (size_t)K);
            TRY(st.send());
            TRY(gp_predict_dev<T>(ctx, N, K, d, lam, N1, N2, Pinf, H, R, dts, dys, t0, dtq, dmean, dvar, dll));
            TRY(st.finish());
            if (ll) *ll = llh;
            return std::isfinite(llh) ? PGPS_OK : PGPS_E_NUMERIC;
        }
    }
    T *dts, *dys, *dtq, *dmean, *dvar;
    double* dll;
    TRY(stage_in(ctx, ctx->st[10], ts, (size_t)N, &dts));
    TRY(stage_in(ctx, ctx->st[4], ys, (size_t)N, &dys));
    TRY(stage_in(ctx, ctx->st[3], tq, (size_t)K, &dtq));
    TRY(stage_in<T>(ctx, ctx->st[7], nullptr, (size_t)K, &dmean));
    TRY(stage_in<T>(ctx, ctx->st[8], nullptr, (size_t)K, &dvar));
    TRY(stage_in<double>(ctx, ctx->st[9], nullptr, 2, &dll));
    TRY(gp_predict_dev<T>(ctx, N, K, d, lam, N1, N2, Pinf, H, R, dts, dys, t0, dtq, dmean, dvar, dll));
    TRY(stage_out(ctx, mean, dmean, (size_t)K));
    TRY(stage_out(ctx, var, dvar, (size_t)K));
    double llh = 0.0;
    TRY(stage_out(ctx, &llh, dll, 1));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (ll) *ll = llh;
    if (!std::isfinite(llh)) return PGPS_E_NUMERIC;
    return PGPS_OK;
}

#define PGPS_DEFINE_PREDICT(SUF, T)                                                                                  \
    extern "C" int pgps_gp_predict_dev_##SUF(pgps_ctx* c, long N, long K, int d, double lam, const double* N1,       \
                                             const double* N2, const double* Pinf, const double* H, double R,       \
                                             const T* ts, const T* ys, double t0, const T* tq, T* mean, T* var,     \
                                             double* ll) {                                                          \
        return gp_predict_dev<T>(c, N, K, d, lam, N1, N2, Pinf, H, R, ts, ys, t0, tq, mean, var, ll);                \
    }                                                                                                                \
    extern "C" int pgps_gp_predict_##SUF(pgps_ctx* c, long N, long K, int d, double lam, const double* N1,           \
                                         const double* N2, const double* Pinf, const double* H, double R,           \
                                         const T* ts, const T* ys, double t0, const T* tq, T* mean, T* var,         \
                                         double* ll) {                                                              \
        return gp_predict_host<T>(c, N, K, d, lam, N1, N2, Pinf, H, R, ts, ys, t0, tq, mean, var, ll);               \
    }

PGPS_DEFINE_PREDICT(f64, double)
PGPS_DEFINE_PREDICT(f32, float)

// ---------------------------------------------------------------------------------------------
// A series kept on the device across calls (round 3).  The reference's drivers evaluate the SAME (ts, ys) thousands of
// times with changing hyper-parameters (L-BFGS: pssgp/experiments/sunspot/map.py:74-82; HMC: experiments/common.py:95-133;
// the speed mesh calls predict_f on fixed grids: toy_models/speed_and_stability.py:73-87): with the host entry points every
// call copied ts and ys to the device again, merged the query grid again and waited for three or four staged copies.
// A pgps_series holds ts, ys (and, once set, the query grid MERGED with them: times, observations with NaN at the query
// rows, query slots) on the device, so that a call sends the model's few scalars and brings back the log-likelihood
// (+ gradient, or the K means and variances) through a pinned buffer: one short launch set and one copy per call.
// fp64, the fused (Matern-family, d <= 3) entry points.
// ---------------------------------------------------------------------------------------------
struct pgps_series {
    pgps_ctx* ctx = nullptr;
    long N = 0, K = 0;
    double t0 = 0.0;
    double *ts = nullptr, *ys = nullptr, *tq = nullptr;
    double *ts_m = nullptr, *ys_m = nullptr, *fms = nullptr, *fPs = nullptr, *pm = nullptr, *pv = nullptr, *res = nullptr;
    int* qslot = nullptr;
    double* host = nullptr;             // pinned: 2 K + 32 doubles
    double* hdev = nullptr;             // the same buffer as the device sees it: the last kernel of a call writes its results
                                        // straight into it (a D2H copy is a blit kernel of its own: ~5 us each, three per
                                        // predict_f; PGPS_SERIES_ZERO_COPY=0 in the environment brings the copies back)
    bool zero_copy = true;
    size_t host_cap = 0;
};

static void series_free_queries(pgps_series* s) {
    for (void* p : {(void*)s->tq, (void*)s->ts_m, (void*)s->ys_m, (void*)s->fms, (void*)s->fPs, (void*)s->pm, (void*)s->pv, (void*)s->qslot})
        if (p) (void)hipFree(p);
    s->tq = s->ts_m = s->ys_m = s->fms = s->fPs = s->pm = s->pv = nullptr;
    s->qslot = nullptr;
    s->K = 0;
}
static int series_host(pgps_series* s, size_t doubles) {
    if (s->host_cap >= doubles) return PGPS_OK;
    if (s->host) (void)hipHostFree(s->host);
    s->host = nullptr; s->host_cap = 0;
    if (hipHostMalloc((void**)&s->host, doubles * sizeof(double), hipHostMallocDefault) != hipSuccess) return PGPS_E_NOMEM;
    s->host_cap = doubles;
    const char* env = std::getenv("PGPS_SERIES_ZERO_COPY");
    s->zero_copy = !(env && env[0] == '0');
    s->hdev = nullptr;
    if (s->zero_copy && hipHostGetDevicePointer((void**)&s->hdev, s->host, 0) != hipSuccess) { s->hdev = nullptr; s->zero_copy = false; }
    return PGPS_OK;
}

extern "C" int pgps_series_destroy(pgps_series* s) {
    if (!s) return PGPS_OK;
    if (s->ctx) { (void)hipSetDevice(s->ctx->device); (void)hipStreamSynchronize(s->ctx->stream); }
    series_free_queries(s);
    if (s->ts) (void)hipFree(s->ts);
    if (s->ys) (void)hipFree(s->ys);
    if (s->res) (void)hipFree(s->res);
    if (s->host) (void)hipHostFree(s->host);
    delete s;
    return PGPS_OK;
}

extern "C" int pgps_series_create_f64(pgps_ctx* ctx, long N, const double* ts, const double* ys, double t0, pgps_series** out) {
    if (!ctx || N < 1 || !ts || !ys || !out) return PGPS_E_INVALID;
    *out = nullptr;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    pgps_series* s = new (std::nothrow) pgps_series();
    if (!s) return PGPS_E_NOMEM;
    s->ctx = ctx; s->N = N; s->t0 = t0;
    const size_t nb = (size_t)N * sizeof(double);
    if (hipMalloc((void**)&s->ts, nb) != hipSuccess || hipMalloc((void**)&s->ys, nb) != hipSuccess ||
        hipMalloc((void**)&s->res, 64 * sizeof(double)) != hipSuccess || series_host(s, 64) != PGPS_OK) {
        pgps_series_destroy(s);
        return PGPS_E_NOMEM;
    }
    if (hipMemcpyAsync(s->ts, ts, nb, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipMemcpyAsync(s->ys, ys, nb, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) {
        pgps_series_destroy(s);
        return PGPS_E_HIP;
    }
    *out = s;
    return PGPS_OK;
}

// the query grid of predict_f: merged with the training series on the device ONCE (pssgp/model.py:15-55 tie rule)
extern "C" int pgps_series_set_queries_f64(pgps_series* s, long K, const double* tq) {
    if (!s || K < 0 || (K > 0 && !tq)) return PGPS_E_INVALID;
    pgps_ctx* ctx = s->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    series_free_queries(s);
    if (K == 0) return PGPS_OK;
    if (s->N + K > 0x7fffffffL) return PGPS_E_INVALID;
    const size_t m = (size_t)(s->N + K);
    if (hipMalloc((void**)&s->tq, (size_t)K * 8) != hipSuccess || hipMalloc((void**)&s->ts_m, m * 8) != hipSuccess ||
        hipMalloc((void**)&s->ys_m, m * 8) != hipSuccess || hipMalloc((void**)&s->qslot, m * 4) != hipSuccess ||
        hipMalloc((void**)&s->fms, m * 3 * 8) != hipSuccess || hipMalloc((void**)&s->fPs, m * 9 * 8) != hipSuccess ||
        hipMalloc((void**)&s->pm, (size_t)K * 8) != hipSuccess || hipMalloc((void**)&s->pv, (size_t)K * 8) != hipSuccess ||
        series_host(s, 2 * (size_t)K + 64) != PGPS_OK) {
        series_free_queries(s);
        return PGPS_E_NOMEM;
    }
    s->K = K;
    HIPCHK(ctx, hipMemcpyAsync(s->tq, tq, (size_t)K * 8, hipMemcpyHostToDevice, ctx->stream));
    TRY(pgps::launch_merge<double>(ctx, s->N, K, s->ts, s->ys, s->tq, s->ts_m, s->ys_m, s->qslot));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return PGPS_OK;
}

extern "C" int pgps_series_info(pgps_series* s, long* N, long* K) {
    if (!s || !N || !K) return PGPS_E_INVALID;
    *N = s->N; *K = s->K;
    return PGPS_OK;
}

// log-likelihood of the fused model on the resident series; ll on the host when the call returns
extern "C" int pgps_series_gp_ll_f64(pgps_series* s, int d, double lam, const double* N1, const double* N2, const double* Pinf,
                                     const double* H, double R, double* ll) {
    if (!s || !ll) return PGPS_E_INVALID;
    pgps_ctx* ctx = s->ctx;
    double* const res = s->zero_copy ? s->hdev : s->res;
    TRY(gp_dev<double>(ctx, s->N, d, lam, N1, N2, Pinf, H, R, s->ts, s->t0, s->ys, nullptr, nullptr, nullptr, nullptr, res));
    if (!s->zero_copy) HIPCHK(ctx, hipMemcpyAsync(s->host, s->res, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *ll = s->host[0];
    return std::isfinite(*ll) ? PGPS_OK : PGPS_E_NUMERIC;
}

// log-likelihood and its gradient (forward-mode duals, pgps_gp_ll_grad_*): out = [ll, d ll / d theta_1 .. np] on the host
extern "C" int pgps_series_gp_ll_grad_f64(pgps_series* s, int d, int np, const double* model, double* out) {
    if (!s || !model || !out || np < 1 || np > 16) return PGPS_E_INVALID;
    pgps_ctx* ctx = s->ctx;
    TRY(launch_grad(ctx, s->N, d, np, model, s->ts, s->t0, s->ys, s->zero_copy ? s->hdev : s->res));
    if (!s->zero_copy) HIPCHK(ctx, hipMemcpyAsync(s->host, s->res, (size_t)(1 + np) * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i <= np; ++i) out[i] = s->host[i];
    return std::isfinite(out[0]) ? PGPS_OK : PGPS_E_NUMERIC;
}

// log-likelihood and the model's adjoints (the adjoint pass, pgps_gpadj.hip.h): out = 1 + d d + 2 d + 1 doubles on the host
extern "C" int pgps_series_gp_ll_grad_adj_f64(pgps_series* s, int d, double lam, const double* N1, const double* N2,
                                              const double* Pinf, const double* H, double R, double* out) {
    if (!s || !out) return PGPS_E_INVALID;
    if (d < 1 || d > 3) return PGPS_E_UNSUPPORTED_DIM;
    pgps_ctx* ctx = s->ctx;
    const int n = 1 + d * d + 2 * d + 1;
    double* const res = s->zero_copy ? s->hdev : s->res;
    TRY(gp_adj_dev(ctx, s->N, d, lam, N1, N2, Pinf, H, R, s->ts, s->t0, s->ys, res));
    if (!s->zero_copy) HIPCHK(ctx, hipMemcpyAsync(s->host, s->res, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < n; ++i) out[i] = s->host[i];
    return std::isfinite(out[0]) ? PGPS_OK : PGPS_E_NUMERIC;
}

// predict_f at the query grid set by pgps_series_set_queries_f64: K means and variances (and the log-likelihood of the
// training series: the query rows are missing observations and contribute nothing) on the host when the call returns
extern "C" int pgps_series_gp_predict_f64(pgps_series* s, int d, double lam, const double* N1, const double* N2,
                                          const double* Pinf, const double* H, double R, double* mean, double* var, double* ll) {
    if (!s || s->K < 1 || !mean || !var || !N1 || !Pinf || !H) return PGPS_E_INVALID;
    if (d < 1 || d > 3) return PGPS_E_UNSUPPORTED_DIM;
    pgps_ctx* ctx = s->ctx;
    RoctxRange range_("parallel_filter");
    GpArgs<double> g{};
    g.s.N = s->N + s->K;
    g.s.R = R;
    g.s.ys = s->ys_m;
    g.s.fms = s->fms; g.s.fPs = s->fPs; g.s.sms = nullptr; g.s.sPs = nullptr;
    const size_t K = (size_t)s->K;
    g.s.ll = s->zero_copy ? s->hdev + 2 * K : s->res;
    g.m.lam = lam;
    for (int i = 0; i < 9; ++i) { g.m.N1[i] = 0; g.m.N2[i] = 0; g.m.Pinf[i] = 0; }
    for (int i = 0; i < d * d; ++i) { g.m.N1[i] = N1[i]; g.m.N2[i] = N2 ? N2[i] : 0.0; g.m.Pinf[i] = Pinf[i]; }
    for (int i = 0; i < 3; ++i) g.m.H[i] = i < d ? H[i] : 0.0;
    g.m.ts = s->ts_m;
    g.m.t_prev = s->t0;
    g.qslot = s->qslot;
    g.pmean = s->zero_copy ? s->hdev : s->pm;
    g.pvar = s->zero_copy ? s->hdev + K : s->pv;
    int rc;
    switch (d) {
        case 1: rc = launch_gp<double, 1>(ctx, g, 1, 1); break;
        case 2: rc = launch_gp<double, 2>(ctx, g, 1, 1); break;
        default: rc = launch_gp<double, 3>(ctx, g, 1, 1); break;
    }
    if (rc) return rc;
    if (!s->zero_copy) {
        HIPCHK(ctx, hipMemcpyAsync(s->host, s->pm, K * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(s->host + K, s->pv, K * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(s->host + 2 * K, s->res, 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    std::memcpy(mean, s->host, K * 8);
    std::memcpy(var, s->host + K, K * 8);
    if (ll) *ll = s->host[2 * K];
    return std::isfinite(s->host[2 * K]) ? PGPS_OK : PGPS_E_NUMERIC;
}

// ---------------------------------------------------------------------------------------------
// general LTI models on the device (fp64, 2 <= d <= 16): _get_ssm -> pkf / pkfs for any kernel, with
// nothing but the results leaving the GPU (row-cooperative kernels, pgps_rc.hip.h)
// ---------------------------------------------------------------------------------------------
// H sm and H sP H^T at the query rows of a merged series (the general-LTI predict path above d = 16, where the smoother
// writes whole moments): one thread per merged step
__global__ void k_project_rows(long m, int d, const int* __restrict__ qslot, const double* __restrict__ H,
                               const double* __restrict__ sms, const double* __restrict__ sPs, double* __restrict__ mean,
                               double* __restrict__ var) {
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= m) return;
    const int q = qslot[k];
    if (q < 0) return;
    const double* sm = sms + k * d;
    const double* sP = sPs + k * (long)d * d;
    double mu = 0.0, v = 0.0;
    for (int i = 0; i < d; ++i) {
        mu += H[i] * sm[i];
        double r = 0.0;
        for (int j = 0; j < d; ++j) r += sP[(long)i * d + j] * H[j];
        v += H[i] * r;
    }
    mean[q] = mu;
    var[q] = v;
}

// Fs / Qs of a block-diagonal model from the per-block results: element (i, j) of block step k goes to
// (idx[i], idx[j]) of the big step k (the big arrays are zero elsewhere)
// (blockIdx.y = one of several blocks of the same size: their model records -- [F_b | P_b | indices] -- lie `mstride` doubles
// apart, their discretised arrays m db^2 apart)
__global__ void k_scatter_block(long m, int d, int db, const int* __restrict__ idx, const double* __restrict__ Fb,
                                const double* __restrict__ Qb, double* __restrict__ Fs, double* __restrict__ Qs, long mstride) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long per = (long)db * db;
    idx = reinterpret_cast<const int*>(reinterpret_cast<const double*>(idx) + (long)blockIdx.y * mstride);
    Fb += (long)blockIdx.y * m * per;
    Qb += (long)blockIdx.y * m * per;
    if (e >= m * per) return;
    const long k = e / per;
    const int r = (int)(e - k * per), i = r / db, j = r - i * db;
    const long dst = k * (long)d * d + (long)idx[i] * d + idx[j];
    Fs[dst] = Fb[e];
    Qs[dst] = Qb[e];
}

// Sum kernels give block-diagonal F and Pinf (pssgp/kernels/base.py:133-141), so expm(F dt) and Q are block-diagonal
// too: connected components of the sparsity pattern of |F| + |F^T| + |Pinf|, single states attached to the smallest
// block.  Returns the components (each sorted) when there are at least two and none is larger than `cap`.
// An entry counts as a coupling when it exceeds 1e-14 of its matrix's largest entry: a Pinf that came out of a Lyapunov
// solve carries rounding residue of that size between independent blocks, and dropping it moves the discretised
// operands by the same relative amount -- five orders below the parity tolerance.
static bool diagonal_blocks(int d, const double* F, const double* P, int cap, std::vector<std::vector<int>>& blocks) {
    std::vector<int> comp(d);
    for (int i = 0; i < d; ++i) comp[i] = i;
    auto find = [&](int x) { while (comp[x] != x) x = comp[x] = comp[comp[x]]; return x; };
    double fmax = 0.0, pmax = 0.0;
    for (int i = 0; i < d * d; ++i) { fmax = std::max(fmax, std::fabs(F[i])); pmax = std::max(pmax, std::fabs(P[i])); }
    const double ftol = 1e-14 * fmax, ptol = 1e-14 * pmax;
    auto coupled = [&](int i, int j) { return std::fabs(F[i * d + j]) > ftol || std::fabs(P[i * d + j]) > ptol; };
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j)
            if (i != j && (coupled(i, j) || coupled(j, i))) {
                const int a = find(i), b = find(j);
                if (a != b) comp[a] = b;
            }
    blocks.clear();
    std::vector<int> slot(d, -1);
    for (int i = 0; i < d; ++i) {
        const int r = find(i);
        if (slot[r] < 0) { slot[r] = (int)blocks.size(); blocks.emplace_back(); }
        blocks[slot[r]].push_back(i);
    }
    // single states: the row-cooperative discretisation starts at d = 2
    for (size_t b = 0; b < blocks.size();) {
        if (blocks[b].size() == 1 && blocks.size() > 1) {
            size_t best = blocks.size();
            for (size_t o = 0; o < blocks.size(); ++o)
                if (o != b && (best == blocks.size() || blocks[o].size() < blocks[best].size())) best = o;
            blocks[best].push_back(blocks[b][0]);
            std::sort(blocks[best].begin(), blocks[best].end());
            blocks.erase(blocks.begin() + (long)b);
            b = 0;
        } else {
            ++b;
        }
    }
    if (blocks.size() < 2) return false;
    for (auto& bl : blocks)
        if ((int)bl.size() > cap || bl.size() < 2) return false;
    return true;
}

// 16 < d <= 32: the same chain on the wave-cooperative kernels -- discretisation with Qs written out, whole filtered (and
// smoothed) moments into scratch, projection at the query rows by k_project_rows.  Everything stays on the device.
// discretisation of the d <= 32 road into the context's scratch: Fs, Qs (m, d, d)
static int lti_disc_wc(pgps_ctx* ctx, size_t m, int d, const double* model, const double* F_host, const double* P_host,
                       const double* ts_m, double t0, double** Fs_out, double** Qs_out) {
    const size_t dd = (size_t)d * d;
    double *Fs, *Qs;
    TRY(stage_in<double>(ctx, ctx->lti[4], nullptr, m * dd, &Fs));
    TRY(stage_in<double>(ctx, ctx->lti[5], nullptr, m * dd, &Qs));
    std::vector<std::vector<int>> blocks;
    if (diagonal_blocks(d, F_host, P_host, rc::kDimMax, blocks)) {
        // block-diagonal model (a sum kernel): every block through the row-cooperative discretisation (Pade in
        // registers, ~20x the rate of wc_discretise), scattered into zeroed Fs / Qs
        HIPCHK(ctx, hipMemsetAsync(Fs, 0, m * dd * sizeof(double), ctx->stream));
        HIPCHK(ctx, hipMemsetAsync(Qs, 0, m * dd * sizeof(double), ctx->stream));
        // blocks of the same size go together: one batched discretisation launch and one scatter launch per SIZE (the CO2
        // kernel's six blocks -- 4, 4, 4, 2, 2, 2 -- are two launches of each instead of six: a launch is ~8 us of a 0.6 ms
        // evaluation at the experiment's 3192 points)
        std::stable_sort(blocks.begin(), blocks.end(), [](const std::vector<int>& a, const std::vector<int>& b) { return a.size() < b.size(); });
        size_t small = 0, big = 0;
        for (size_t b = 0; b < blocks.size();) {
            size_t e = b;
            while (e < blocks.size() && blocks[e].size() == blocks[b].size()) ++e;
            const size_t db = blocks[b].size();
            small += (e - b) * (2 * db * db + (db + 1) / 2 * 2);                // F_b, P_b, indices (ints, padded)
            big = std::max(big, (e - b) * db * db);
            b = e;
        }
        double *bm, *Fb, *Qb;
        TRY(stage_in<double>(ctx, ctx->lti[10], nullptr, small, &bm));
        TRY(stage_in<double>(ctx, ctx->lti[11], nullptr, 2 * m * big, &Fb));
        Qb = Fb + m * big;
        std::vector<double> hb(small);
        size_t off = 0;
        std::vector<size_t> offs;
        for (auto& bl : blocks) {
            const size_t db = bl.size();
            offs.push_back(off);
            for (size_t i = 0; i < db; ++i)
                for (size_t j = 0; j < db; ++j) {
                    hb[off + i * db + j] = F_host[bl[i] * d + bl[j]];
                    hb[off + db * db + i * db + j] = P_host[bl[i] * d + bl[j]];
                }
            int* ip = reinterpret_cast<int*>(&hb[off + 2 * db * db]);
            for (size_t i = 0; i < db; ++i) ip[i] = bl[i];
            off += 2 * db * db + (db + 1) / 2 * 2;
        }
        HIPCHK(ctx, hipMemcpyAsync(bm, hb.data(), small * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        // (pageable source: staged by the runtime before the call returns, as for the model in lti_dev)
        for (size_t b = 0; b < blocks.size();) {
            size_t e = b;
            while (e < blocks.size() && blocks[e].size() == blocks[b].size()) ++e;
            const int db = (int)blocks[b].size(), nb = (int)(e - b);
            const long mstride = 2L * db * db + (db + 1) / 2 * 2;
            const double* Fd = bm + offs[b];
            TRY(launch_disc_rc(ctx, (long)m, db, Fd, Fd + db * db, ts_m, t0, Fb, Qb, nb, mstride));
            const long total = (long)m * db * db;
            hipLaunchKernelGGL(k_scatter_block, dim3((unsigned)((total + 255) / 256), (unsigned)nb), dim3(256), 0, ctx->stream,
                               (long)m, d, db, reinterpret_cast<const int*>(Fd + 2 * db * db), (const double*)Fb,
                               (const double*)Qb, Fs, Qs, mstride);
            HIPCHK(ctx, hipGetLastError());
            b = e;
        }
    } else {
        TRY(launch_disc_wc<double>(ctx, (long)m, d, model, model + dd, ts_m, t0, Fs, Qs));
    }
    *Fs_out = Fs;
    *Qs_out = Qs;
    return PGPS_OK;
}

static int lti_dev_wc(pgps_ctx* ctx, size_t m, int d, const double* model, const double* F_host, const double* P_host,
                      double R, const double* ts_m, const double* ys_m, double t0, const int* qslot, double* mean,
                      double* var, double* ll) {
    const size_t dd = (size_t)d * d;
    double *Fs, *Qs, *fms, *fPs;
    TRY(lti_disc_wc(ctx, m, d, model, F_host, P_host, ts_m, t0, &Fs, &Qs));
    ScanArgs<double> a{};
    a.N = (long)m; a.seg_first = 1; a.seg_last = 1;
    a.P0 = model + dd; a.H = model + 2 * dd; a.R = R; a.Fs = Fs; a.Qs = Qs; a.ys = ys_m;
    a.ll = ll;
    TRY(stage_in<double>(ctx, ctx->lti[6], nullptr, m * dd, &fPs));
    TRY(stage_in<double>(ctx, ctx->lti[7], nullptr, m * d, &fms));
    a.fms = fms; a.fPs = fPs;
    if (!qslot) return launch_scan_wc<double>(ctx, a, d, MODE_PKF);
    // smoothed moments in place of the filtered ones is not possible (the smoother reads both): two more buffers
    double *sms, *sPs;
    TRY(stage_in<double>(ctx, ctx->lti[8], nullptr, m * dd, &sPs));
    TRY(stage_in<double>(ctx, ctx->lti[9], nullptr, m * d, &sms));
    a.sms = sms; a.sPs = sPs;
    TRY(launch_scan_wc<double>(ctx, a, d, MODE_PKFS));
    const int block = 256;
    hipLaunchKernelGGL(k_project_rows, dim3((unsigned)((m + block - 1) / block)), dim3(block), 0, ctx->stream, (long)m, d,
                       qslot, a.H, (const double*)sms, (const double*)sPs, mean, var);
    HIPCHK(ctx, hipGetLastError());
    return PGPS_OK;
}

// the small model [F | Pinf | H] from host memory, in ONE copy (three pageable copies were 15 us of a short series' call;
// the source is staged by the runtime before the call returns, so it may change afterwards)
static int lti_model_in(pgps_ctx* ctx, int d, const double* F, const double* Pinf, const double* H, double** model) {
    const size_t dd = (size_t)d * d;
    TRY(ensure(ctx, ctx->lti[0], (2 * dd + d) * sizeof(double)));
    *model = (double*)ctx->lti[0].p;
    double host[2 * PGPS_MAX_DIM * PGPS_MAX_DIM + PGPS_MAX_DIM];
    std::memcpy(host, F, dd * sizeof(double));
    std::memcpy(host + dd, Pinf, dd * sizeof(double));
    std::memcpy(host + 2 * dd, H, (size_t)d * sizeof(double));
    HIPCHK(ctx, hipMemcpyAsync(*model, host, (2 * dd + d) * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    return PGPS_OK;
}

// m steps (training and query rows merged: qslot != nullptr marks the query rows and asks for the posterior there),
// everything on the device except the model
static int lti_core(pgps_ctx* ctx, size_t m, int d, const double* F, const double* Pinf, const double* H, double R,
                    const double* ts_m, const double* ys_m, double t0, const int* qslot, double* mean, double* var,
                    double* ll) {
    const size_t dd = (size_t)d * d;
    double* model;
    TRY(lti_model_in(ctx, d, F, Pinf, H, &model));
    double *Fs, *Qs = nullptr, *dll;
    TRY(stage_in<double>(ctx, ctx->st[11], nullptr, 2, &dll));
    if (d > rc::kDimMax) return lti_dev_wc(ctx, m, d, model, F, Pinf, R, ts_m, ys_m, t0, qslot, mean, var, ll ? ll : dll);
    TRY(stage_in<double>(ctx, ctx->lti[4], nullptr, m * dd, &Fs));
    // the process noise stays implicit in both calls (Q_k = Pinf - F_k Pinf F_k^T inside the predict step): Qs is
    // never formed
    TRY(launch_disc_rc(ctx, (long)m, d, model, model + dd, ts_m, t0, Fs, Qs));
    ScanArgs<double> a{};
    a.N = (long)m; a.seg_first = 1; a.seg_last = 1;
    a.P0 = model + dd; a.H = model + 2 * dd; a.R = R; a.Fs = Fs; a.Qs = Qs; a.ys = ys_m;
    a.ll = ll ? ll : dll;
    if (!qslot) return launch_scan_rc_proj(ctx, a, d, MODE_PKF, nullptr, nullptr, nullptr);
    // scratch for the smoothing elements (E, g) where pkfs would have sPs, sms
    TRY(stage_in<double>(ctx, ctx->lti[6], nullptr, m * dd, &a.sPs));
    TRY(stage_in<double>(ctx, ctx->lti[7], nullptr, m * d, &a.sms));
    return launch_scan_rc_proj(ctx, a, d, MODE_PKFS, qslot, mean, var);
}

static int lti_dev(pgps_ctx* ctx, long N, long K, int d, const double* F, const double* Pinf, const double* H, double R,
                   const double* ts, const double* ys, double t0, const double* tq, double* mean, double* var,
                   double* ll) {
    if (!ctx || N < 1 || K < 0 || !F || !Pinf || !H || !ts || !ys) return PGPS_E_INVALID;
    if (K > 0 && (!tq || !mean || !var)) return PGPS_E_INVALID;
    if (K == 0 && !ll) return PGPS_E_INVALID;
    if (d < rc::kDimMin || d > PGPS_MAX_DIM) return PGPS_E_UNSUPPORTED_DIM;
    if (N + K > 0x7fffffffL) return PGPS_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t m = (size_t)(N + K);
    const double *ts_m = ts, *ys_m = ys;
    int* qslot = nullptr;
    if (K > 0) {
        double *tsm, *ysm;
        TRY(stage_in<double>(ctx, ctx->lti[1], nullptr, m, &tsm));
        TRY(stage_in<double>(ctx, ctx->lti[2], nullptr, m, &ysm));
        TRY(stage_in<int>(ctx, ctx->lti[3], nullptr, m, &qslot));
        TRY(launch_merge<double>(ctx, N, K, ts, ys, tq, tsm, ysm, qslot));
        ts_m = tsm; ys_m = ysm;
    }
    return lti_core(ctx, m, d, F, Pinf, H, R, ts_m, ys_m, t0, qslot, mean, var, ll);
}

static int lti_host(pgps_ctx* ctx, long N, long K, int d, const double* F, const double* Pinf, const double* H, double R,
                    const double* ts, const double* ys, double t0, const double* tq, double* mean, double* var,
                    double* ll) {
    if (!ctx || N < 1 || K < 0 || !ts || !ys) return PGPS_E_INVALID;
    if (K > 0 && (!tq || !mean || !var)) return PGPS_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    {
        SmallStage st(ctx, 2 * SmallStage::up((size_t)N * 8) + SmallStage::up((size_t)K * 8), 2 * SmallStage::up((size_t)K * 8) + 16);
        if (st.ok) {
            double llh = 0.0;
            double* dts_ = st.in(ts, (size_t)N);
            double* dys_ = st.in(ys, (size_t)N);
            double* dtq_ = K > 0 ? st.in(tq, (size_t)K) : nullptr;
            double* dll_ = st.out(&llh, 1);
            double* dmean_ = K > 0 ? st.out(mean, (size_t)K) : nullptr;
            double* dvar_ = K > 0 ? st.out(var, (size_t)K) : nullptr;
            TRY(st.send());
            TRY(lti_dev(ctx, N, K, d, F, Pinf, H, R, dts_, dys_, t0, dtq_, dmean_, dvar_, dll_));
            TRY(st.finish());
            if (ll) *ll = llh;
            return std::isfinite(llh) ? PGPS_OK : PGPS_E_NUMERIC;
        }
    }
    double *dts, *dys, *dtq = nullptr, *dmean = nullptr, *dvar = nullptr, *dll;
    TRY(stage_in(ctx, ctx->st[10], ts, (size_t)N, &dts));
    TRY(stage_in(ctx, ctx->st[4], ys, (size_t)N, &dys));
    if (K > 0) {
        TRY(stage_in(ctx, ctx->st[3], tq, (size_t)K, &dtq));
        TRY(stage_in<double>(ctx, ctx->st[7], nullptr, (size_t)K, &dmean));
        TRY(stage_in<double>(ctx, ctx->st[8], nullptr, (size_t)K, &dvar));
    }
    TRY(stage_in<double>(ctx, ctx->st[9], nullptr, 2, &dll));
    TRY(lti_dev(ctx, N, K, d, F, Pinf, H, R, dts, dys, t0, dtq, dmean, dvar, dll));
    if (K > 0) {
        TRY(stage_out(ctx, mean, dmean, (size_t)K));
        TRY(stage_out(ctx, var, dvar, (size_t)K));
    }
    double llh = 0.0;
    TRY(stage_out(ctx, &llh, dll, 1));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (ll) *ll = llh;
    if (!std::isfinite(llh)) return PGPS_E_NUMERIC;
    return PGPS_OK;
}

// B models over one series: table = B x [F | Pinf | H | R] from host memory
static int lti_ll_batch_dev(pgps_ctx* ctx, int B, long N, int d, const double* models, const double* ts, const double* ys,
                            double t0, double* ll) {
    if (!ctx || B < 1 || N < 1 || !models || !ts || !ys || !ll) return PGPS_E_INVALID;
    if (d < rc::kDimMin || d > rc::kDimMax) return PGPS_E_UNSUPPORTED_DIM;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t dd = (size_t)d * d, ms = 2 * dd + d + 1;
    double *table, *Fs;
    TRY(stage_in<double>(ctx, ctx->lti[0], models, (size_t)B * ms, &table));
    TRY(stage_in<double>(ctx, ctx->lti[4], nullptr, (size_t)B * (size_t)N * dd, &Fs));
    TRY(launch_disc_rc(ctx, N, d, table, table + dd, ts, t0, Fs, nullptr, B, (long)ms));     // implicit process noise
    return launch_ll_batch_rc(ctx, N, d, B, table, (long)ms, Fs, nullptr, ys, ll);
}

static int lti_ll_batch_host(pgps_ctx* ctx, int B, long N, int d, const double* models, const double* ts, const double* ys,
                             double t0, double* ll) {
    if (!ctx || B < 1 || N < 1 || !models || !ts || !ys || !ll) return PGPS_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    double *dts, *dys, *dll;
    TRY(stage_in(ctx, ctx->st[10], ts, (size_t)N, &dts));
    TRY(stage_in(ctx, ctx->st[4], ys, (size_t)N, &dys));
    TRY(stage_in<double>(ctx, ctx->st[9], nullptr, (size_t)B, &dll));
    TRY(lti_ll_batch_dev(ctx, B, N, d, models, dts, dys, t0, dll));
    TRY(stage_out(ctx, ll, dll, (size_t)B));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return PGPS_OK;
}

extern "C" int pgps_lti_ll_batch_f64(pgps_ctx* c, int B, long N, int d, const double* models, const double* ts,
                                     const double* ys, double t0, double* ll) {
    return lti_ll_batch_host(c, B, N, d, models, ts, ys, t0, ll);
}
extern "C" int pgps_lti_ll_batch_dev_f64(pgps_ctx* c, int B, long N, int d, const double* models, const double* ts,
                                         const double* ys, double t0, double* ll) {
    return lti_ll_batch_dev(c, B, N, d, models, ts, ys, t0, ll);
}

// the same three calls for ANY kernel's LTI model (F, Pinf, H from the host; fp64, 2 <= d <= 32): pgps_lti_ll_* /
// pgps_lti_predict_* / pgps_lti_ll_batch_* on the resident series and its merged query grid
extern "C" int pgps_series_lti_ll_f64(pgps_series* s, int d, const double* F, const double* Pinf, const double* H, double R,
                                      double* ll) {
    if (!s || !ll || !F || !Pinf || !H) return PGPS_E_INVALID;
    if (d < rc::kDimMin || d > PGPS_MAX_DIM) return PGPS_E_UNSUPPORTED_DIM;
    pgps_ctx* ctx = s->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    double* const res = s->zero_copy ? s->hdev : s->res;
    TRY(lti_core(ctx, (size_t)s->N, d, F, Pinf, H, R, s->ts, s->ys, s->t0, nullptr, nullptr, nullptr, res));
    if (!s->zero_copy) HIPCHK(ctx, hipMemcpyAsync(s->host, s->res, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *ll = s->host[0];
    return std::isfinite(*ll) ? PGPS_OK : PGPS_E_NUMERIC;
}

extern "C" int pgps_series_lti_predict_f64(pgps_series* s, int d, const double* F, const double* Pinf, const double* H, double R,
                                           double* mean, double* var, double* ll) {
    if (!s || s->K < 1 || !mean || !var || !F || !Pinf || !H) return PGPS_E_INVALID;
    if (d < rc::kDimMin || d > PGPS_MAX_DIM) return PGPS_E_UNSUPPORTED_DIM;
    pgps_ctx* ctx = s->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t K = (size_t)s->K;
    double* const dm = s->zero_copy ? s->hdev : s->pm;
    double* const dv = s->zero_copy ? s->hdev + K : s->pv;
    double* const dl = s->zero_copy ? s->hdev + 2 * K : s->res;
    TRY(lti_core(ctx, (size_t)(s->N + s->K), d, F, Pinf, H, R, s->ts_m, s->ys_m, s->t0, s->qslot, dm, dv, dl));
    if (!s->zero_copy) {
        HIPCHK(ctx, hipMemcpyAsync(s->host, s->pm, K * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(s->host + K, s->pv, K * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(s->host + 2 * K, s->res, 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    std::memcpy(mean, s->host, K * 8);
    std::memcpy(var, s->host + K, K * 8);
    if (ll) *ll = s->host[2 * K];
    return std::isfinite(s->host[2 * K]) ? PGPS_OK : PGPS_E_NUMERIC;
}

extern "C" int pgps_series_lti_ll_batch_f64(pgps_series* s, int B, int d, const double* models, double* ll) {
    if (!s || B < 1 || !models || !ll) return PGPS_E_INVALID;
    pgps_ctx* ctx = s->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    double* dll;
    if (s->zero_copy && (size_t)B <= s->host_cap) dll = s->hdev;
    else TRY(stage_in<double>(ctx, ctx->st[9], nullptr, (size_t)B, &dll));
    TRY(lti_ll_batch_dev(ctx, B, s->N, d, models, s->ts, s->ys, s->t0, dll));
    if (dll != s->hdev) {
        TRY(stage_out(ctx, ll, dll, (size_t)B));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        return PGPS_OK;
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    std::memcpy(ll, s->host, (size_t)B * sizeof(double));
    return PGPS_OK;
}

// log-likelihood and the model's adjoints (pgps_gradlti.h): [ll | Abar | Ubar | Hbar | Rbar]
static int lti_grad_dev(pgps_ctx* ctx, long N, int d, const double* F, const double* Pinf, const double* H, double R,
                        const double* ts, const double* ys, double t0, double* out) {
    if (!ctx || N < 1 || !F || !Pinf || !H || !ts || !ys || !out) return PGPS_E_INVALID;
    if (d < rc::kDimMin || d > PGPS_MAX_DIM) return PGPS_E_UNSUPPORTED_DIM;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    double* model;
    TRY(lti_model_in(ctx, d, F, Pinf, H, &model));
    // row-cooperative kernels up to d = 16; above that (and wherever the wave-cooperative family is forced: the tests'
    // cross-check) the wave-cooperative ones, from discretised arrays
    if (d <= rc::kDimMax && ctx->family != 2) return launch_ll_grad_lti(ctx, N, d, model, R, ts, t0, ys, out);
    double *Fs, *Qs;
    TRY(lti_disc_wc(ctx, (size_t)N, d, model, F, Pinf, ts, t0, &Fs, &Qs));
    return launch_ll_grad_lti_wc(ctx, N, d, model, R, Fs, Qs, ts, t0, ys, out);
}
extern "C" int pgps_lti_ll_grad_dev_f64(pgps_ctx* c, long N, int d, const double* F, const double* Pinf, const double* H,
                                        double R, const double* ts, const double* ys, double t0, double* out) {
    return lti_grad_dev(c, N, d, F, Pinf, H, R, ts, ys, t0, out);
}
extern "C" int pgps_lti_ll_grad_f64(pgps_ctx* ctx, long N, int d, const double* F, const double* Pinf, const double* H,
                                    double R, const double* ts, const double* ys, double t0, double* out) {
    if (!ctx || N < 1 || !ts || !ys || !out) return PGPS_E_INVALID;
    if (d < rc::kDimMin || d > PGPS_MAX_DIM) return PGPS_E_UNSUPPORTED_DIM;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t nout = 1 + (size_t)grad_lti_nstat(d);
    double *dts, *dys, *dout;
    TRY(stage_in(ctx, ctx->st[10], ts, (size_t)N, &dts));
    TRY(stage_in(ctx, ctx->st[4], ys, (size_t)N, &dys));
    TRY(stage_in<double>(ctx, ctx->st[9], nullptr, nout, &dout));
    TRY(lti_grad_dev(ctx, N, d, F, Pinf, H, R, dts, dys, t0, dout));
    TRY(stage_out(ctx, out, dout, nout));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return std::isfinite(out[0]) ? PGPS_OK : PGPS_E_NUMERIC;
}
extern "C" int pgps_series_lti_ll_grad_f64(pgps_series* s, int d, const double* F, const double* Pinf, const double* H, double R,
                                           double* out) {
    if (!s || !out || !F || !Pinf || !H) return PGPS_E_INVALID;
    if (d < rc::kDimMin || d > PGPS_MAX_DIM) return PGPS_E_UNSUPPORTED_DIM;
    pgps_ctx* ctx = s->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t nout = 1 + (size_t)grad_lti_nstat(d);
    TRY(series_host(s, std::max<size_t>(nout, s->host_cap)));
    double* dout;
    if (s->zero_copy) dout = s->hdev;
    else TRY(stage_in<double>(ctx, ctx->st[9], nullptr, nout, &dout));
    TRY(lti_grad_dev(ctx, s->N, d, F, Pinf, H, R, s->ts, s->ys, s->t0, dout));
    if (!s->zero_copy) HIPCHK(ctx, hipMemcpyAsync(s->host, dout, nout * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    std::memcpy(out, s->host, nout * sizeof(double));
    return std::isfinite(out[0]) ? PGPS_OK : PGPS_E_NUMERIC;
}

extern "C" int pgps_lti_ll_f64(pgps_ctx* c, long N, int d, const double* F, const double* Pinf, const double* H, double R,
                               const double* ts, const double* ys, double t0, double* ll) {
    if (!ll) return PGPS_E_INVALID;
    return lti_host(c, N, 0, d, F, Pinf, H, R, ts, ys, t0, nullptr, nullptr, nullptr, ll);
}
extern "C" int pgps_lti_ll_dev_f64(pgps_ctx* c, long N, int d, const double* F, const double* Pinf, const double* H,
                                   double R, const double* ts, const double* ys, double t0, double* ll) {
    return lti_dev(c, N, 0, d, F, Pinf, H, R, ts, ys, t0, nullptr, nullptr, nullptr, ll);
}
extern "C" int pgps_lti_predict_f64(pgps_ctx* c, long N, long K, int d, const double* F, const double* Pinf,
                                    const double* H, double R, const double* ts, const double* ys, double t0,
                                    const double* tq, double* mean, double* var, double* ll) {
    if (K < 1) return PGPS_E_INVALID;
    return lti_host(c, N, K, d, F, Pinf, H, R, ts, ys, t0, tq, mean, var, ll);
}
extern "C" int pgps_lti_predict_dev_f64(pgps_ctx* c, long N, long K, int d, const double* F, const double* Pinf,
                                        const double* H, double R, const double* ts, const double* ys, double t0,
                                        const double* tq, double* mean, double* var, double* ll) {
    if (K < 1) return PGPS_E_INVALID;
    return lti_dev(c, N, K, d, F, Pinf, H, R, ts, ys, t0, tq, mean, var, ll);
}

// ---------------------------------------------------------------------------------------------
// batched log-likelihood: B hyper-parameter settings over one series
// ---------------------------------------------------------------------------------------------
template <typename T>
static int gp_ll_batch_dev(pgps_ctx* ctx, int B, long N, int d, const double* models_host, const T* ts, double t0,
                           const T* ys, double* ll) {
    if (!ctx || B < 1 || B > 65535 || N < 1 || !models_host || !ts || !ys || !ll) return PGPS_E_INVALID;
    if (d < 1 || d > 3) return PGPS_E_UNSUPPORTED_DIM;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // models: B blocks [lam | N1 (d*d) | N2 (d*d) | Pinf (d*d) | H (d) | R] from the caller, re-packed
    // to the fixed device stride
    std::vector<double> packed((size_t)B * kGpModelStride, 0.0);
    const int in_stride = 1 + 3 * d * d + d + 1;
    for (int m = 0; m < B; ++m) {
        const double* p = models_host + (size_t)m * in_stride;
        double* q = packed.data() + (size_t)m * kGpModelStride;
        q[0] = p[0];
        for (int i = 0; i < d * d; ++i) { q[1 + i] = p[1 + i]; q[10 + i] = p[1 + d * d + i]; q[19 + i] = p[1 + 2 * d * d + i]; }
        for (int i = 0; i < d; ++i) q[28 + i] = p[1 + 3 * d * d + i];
        q[31] = p[1 + 3 * d * d + d];
        if (!(q[0] > 0.0) || !(q[31] > 0.0)) return PGPS_E_INVALID;
    }
    double* dmodels;
    TRY(stage_in<double>(ctx, ctx->st[0], nullptr, packed.size(), &dmodels));
    // the packed vector dies with this frame: synchronous copy (pageable memory, so hipMemcpyAsync would
    // stage it anyway)
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipMemcpy(dmodels, packed.data(), packed.size() * sizeof(double), hipMemcpyHostToDevice));
    GpBatchArgs<T> b{};
    b.N = N;
    b.ts = ts;
    b.ys = ys;
    b.t_prev = (T)t0;
    b.models = dmodels;
    b.ll = ll;
    switch (d) {
        case 1: return launch_gp_batch<T, 1>(ctx, B, b);
        case 2: return launch_gp_batch<T, 2>(ctx, B, b);
        default: return launch_gp_batch<T, 3>(ctx, B, b);
    }
}

template <typename T>
static int gp_ll_batch_host(pgps_ctx* ctx, int B, long N, int d, const double* models, const T* ts, double t0,
                            const T* ys, double* ll) {
    if (!ctx || B < 1 || N < 1 || !ts || !ys || !ll) return PGPS_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    T *dts, *dys;
    double* dll;
    TRY(stage_in(ctx, ctx->st[10], ts, (size_t)N, &dts));
    TRY(stage_in(ctx, ctx->st[4], ys, (size_t)N, &dys));
    TRY(stage_in<double>(ctx, ctx->st[9], nullptr, (size_t)B, &dll));
    TRY(gp_ll_batch_dev<T>(ctx, B, N, d, models, dts, t0, dys, dll));
    TRY(stage_out(ctx, ll, dll, (size_t)B));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return PGPS_OK;
}

#define PGPS_DEFINE_LL_BATCH(SUF, T)                                                                                \
    extern "C" int pgps_gp_ll_batch_dev_##SUF(pgps_ctx* c, int B, long N, int d, const double* models, const T* ts, \
                                              double t0, const T* ys, double* ll) {                                \
        return gp_ll_batch_dev<T>(c, B, N, d, models, ts, t0, ys, ll);                                              \
    }                                                                                                               \
    extern "C" int pgps_gp_ll_batch_##SUF(pgps_ctx* c, int B, long N, int d, const double* models, const T* ts,     \
                                          double t0, const T* ys, double* ll) {                                    \
        return gp_ll_batch_host<T>(c, B, N, d, models, ts, t0, ys, ll);                                             \
    }

PGPS_DEFINE_LL_BATCH(f64, double)
PGPS_DEFINE_LL_BATCH(f32, float)

// ---------------------------------------------------------------------------------------------
// log-likelihood and its gradient (fused path, forward-mode duals through the scan)
// ---------------------------------------------------------------------------------------------
extern "C" int pgps_gp_ll_grad_dev_f64(pgps_ctx* ctx, long N, int d, int np, const double* model, const double* ts,
                                       double t0, const double* ys, double* out) {
    if (!ctx || N < 1 || !model || !ts || !ys || !out) return PGPS_E_INVALID;
    return launch_grad(ctx, N, d, np, model, ts, t0, ys, out);
}

static int gradb_dispatch(pgps_ctx* ctx, long N, int d, int nblk, const int* bsize, int np, const double* model,
                          const double* ts, double t0, const double* ys, double* out) {
    if (!ctx || N < 1 || !bsize || !model || !ts || !ys || !out) return PGPS_E_INVALID;
    if (nblk < 1 || nblk > 4 || np < 1 || np > 16) return PGPS_E_INVALID;
    int sum = 0;
    for (int b = 0; b < nblk; ++b) { if (bsize[b] < 1) return PGPS_E_INVALID; sum += bsize[b]; }
    if (sum != d) return PGPS_E_INVALID;
    RoctxRange range_("parallel_filter");
    switch (d) {
        case 2: return launch_gradb<2>(ctx, N, nblk, bsize, np, model, ts, t0, ys, out);
        case 3: return launch_gradb<3>(ctx, N, nblk, bsize, np, model, ts, t0, ys, out);
        case 4: return launch_gradb<4>(ctx, N, nblk, bsize, np, model, ts, t0, ys, out);
        case 5: return launch_gradb<5>(ctx, N, nblk, bsize, np, model, ts, t0, ys, out);
        case 6: return launch_gradb<6>(ctx, N, nblk, bsize, np, model, ts, t0, ys, out);
        default: return PGPS_E_UNSUPPORTED_DIM;
    }
}

extern "C" int pgps_gp_ll_grad_blocks_dev_f64(pgps_ctx* ctx, long N, int d, int nblk, const int* bsize, int np,
                                              const double* model, const double* ts, double t0, const double* ys,
                                              double* out) {
    return gradb_dispatch(ctx, N, d, nblk, bsize, np, model, ts, t0, ys, out);
}

extern "C" int pgps_gp_ll_grad_blocks_f64(pgps_ctx* ctx, long N, int d, int nblk, const int* bsize, int np,
                                          const double* model, const double* ts, double t0, const double* ys, double* out) {
    if (!ctx || N < 1 || !model || !ts || !ys || !out || np < 1 || np > 16) return PGPS_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    double *dts, *dys, *dout;
    TRY(stage_in(ctx, ctx->st[10], ts, (size_t)N, &dts));
    TRY(stage_in(ctx, ctx->st[4], ys, (size_t)N, &dys));
    TRY(stage_in<double>(ctx, ctx->st[9], nullptr, (size_t)(1 + 3 * np), &dout));
    TRY(gradb_dispatch(ctx, N, d, nblk, bsize, np, model, dts, t0, dys, dout));
    TRY(stage_out(ctx, out, dout, (size_t)(1 + np)));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (!std::isfinite(out[0])) return PGPS_E_NUMERIC;
    return PGPS_OK;
}

extern "C" int pgps_gp_ll_grad_f64(pgps_ctx* ctx, long N, int d, int np, const double* model, const double* ts,
                                   double t0, const double* ys, double* out) {
    if (!ctx || N < 1 || !model || !ts || !ys || !out) return PGPS_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    double *dts, *dys, *dout;
    TRY(stage_in(ctx, ctx->st[10], ts, (size_t)N, &dts));
    TRY(stage_in(ctx, ctx->st[4], ys, (size_t)N, &dys));
    TRY(stage_in<double>(ctx, ctx->st[9], nullptr, 16, &dout));
    TRY(launch_grad(ctx, N, d, np, model, dts, t0, dys, dout));
    TRY(stage_out(ctx, out, dout, (size_t)(1 + np)));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (!std::isfinite(out[0])) return PGPS_E_NUMERIC;
    return PGPS_OK;
}
