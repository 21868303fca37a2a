// pgps_internal.h -- context, scratch and launch plumbing shared by the C-ABI translation
// unit (pgps_core.hip) and the per-(dtype, d) kernel translation units (pgps_inst.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <string>
#include <vector>

#include "../../include/pgps.h"
#include "pgps_dyn.h"

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

struct pgps_ctx {
    int device = 0;
    int n_cu = 0;                       // compute units of the device (hipDeviceAttributeMultiprocessorCount)
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    int chunk = 0;                      // 0 = auto
    int stage_g = -1;                   // LDS staging: -1 = auto, 0 = off, 2 / 4 = steps per sub-tile
    int single_pass = -1;               // single-pass filter kernel: -1 = auto, 0 = off, 1 = on
    int lookback_window = 256;          // tiles per look-back window (<= 256; small values are for tests)
    int block = 0;                      // lane-chunk workgroups: 0 = auto, 128 / 256 lanes (pgps_set_block)
    int one_launch = -1;                // fused (pgps_gp_*) calls of short series in ONE launch: -1 = auto (N <= kOneLaunchAuto), 0 = never, n > 0 = up to n steps
    long grad_pack = -1;                // gradient at d <= 2: one direction per model up to this many steps (-1 = automatic, 0 = never)
    int rc_scan = -1;                   // scans of the chain totals (row- / quad-cooperative families): -1 = auto, 0 = one launch per Kogge-Stone level, 1 = blocked (pgps_set_rc_scan)
    int dma = -1;                       // LDS-DMA ring in the Kalman pass (d = 2 fp64, 128-lane build): -1 = auto, 0 = off, 1 = on
    int shortcut = 1;                   // forgetting shortcut for the carry across workgroups (pgps_set_shortcut): 1 = where it applies, 0 = never
    int resident = -1;                  // one resident launch for filter + smoother (pgps_resident.hip.h): -1 = auto, 0 = off, 1 = on where it fits, 2 = on + phase stamps
    unsigned res_epoch = 0;             // launches of the resident kernel so far: picks the barrier's counter set
    DevBuf res_stamps;                  // diagnostics: (workgroups, 16) cycle stamps of the last resident launch
    int res_stamp_blocks = 0;
    int family = 0;                     // 0 = auto (lane-chunk d <= 6; row-cooperative fp64 d <= 16; else wave-cooperative), 1 = lane, 2 = wave, 3 = row
    std::string hip_err;
    DevBuf ws;                          // scratch of the scan kernels
    DevBuf st[12];                      // staging buffers of the host entry points
    bool wc_attr_done[2][17] = {};      // wave-cooperative kernels: dynamic-LDS attributes set (by dtype, DP / 2)
    int wc_rows2 = 15;                  // d = 17..32: two-rows level-1 kernels (pgps_rc2.hip.h); 0 = the LDS-tile ones (env PGPS_WC_ROWS2)
    int wc_serial3 = 0;                 // wave-cooperative family: serial level 3 instead of Kogge-Stone (env PGPS_WC_SERIAL3)
    DevBuf lti[12];                     // general-LTI entry points: model, merged series, Fs, Qs, E, g (d > 16: moments)
    DevBuf stamps;                      // diagnostic build only
    int* status_word = nullptr;         // device word kernels raise flags in (pgps_status)
    int host_flags = 0;                 // flags raised on the host side (bit 2: a float32 call ran in fp64 arithmetic), reported and cleared by pgps_status
    int f32_policy = 0;                 // float32 series with a smoother: 0 = automatic promotion on dense grids, 1 = always native, 2 = always fp64 arithmetic (pgps_set_f32_policy)
    int* probe_host = nullptr;          // pinned word pair the dense-grid probe of a float32 call writes (and its device alias)
    int* probe_dev = nullptr;
    int probe_seq = 0;                  // sequence number of the last probe: its last workgroup writes it to probe_host[0] behind the verdict
    int f32_last_promoted = 0;          // which way the last probed call went: decides the ORDER of the next one (see pgps_core.hip)
    // small host-array calls (pgps_gp_predict_*, pgps_lti_predict_f64, ...): one pinned arena, ONE copy in and ONE copy out
    char* pin_h = nullptr;              // hipHostMalloc, kPinArena bytes, made at the first small call
    DevBuf pin_d;                       // its device twin
    DevBuf gadj;                        // fused-path adjoint gradient (pgps_gpadj.hip.h): kept states of the forward pass, workgroup partials
    DevBuf wide[9];                     // fp64 copies of a promoted float32 call's arrays: P0, H, Fs, Qs, ys, fms, fPs, sms, sPs
    void* comm = nullptr;               // ncclComm_t (RCCL) of a series sharded over GPUs: pgps_comm_init (pgps_comm.hip)
    int comm_rank = 0, comm_nranks = 0;
    DevBuf comm_buf;                    // [rec_f | rec_s | gathered_f | gathered_s] of pgps_pkfs_seg_dev_*
    // the three-phase segment protocol keeps state in `ws` between its calls: what the last phase left, and the
    // workspace epoch (bumped by every call that carves `ws`) it left it at
    struct SegTag { int phase = 0; long N = 0; int d = 0, rank = 0, nranks = 0, chunk = 0, block = 0, family = 0, stage_g = 0, dma = 0, rc_scan = 0; unsigned long epoch = 0; } seg_tag;
    unsigned long ws_epoch = 0;
    unsigned profiling = 0;             // bit i = time launches of slot PGPS_K_* i
    int prof_every = 1;                 // time every n-th launch of an enabled slot
    long prof_seen[PGPS_K_COUNT] = {0};
    struct EvPair { hipEvent_t a, b; int slot; };
    std::vector<EvPair> ev_pool;
    size_t ev_used = 0;
    double prof_ms[PGPS_K_COUNT] = {0};
    long prof_n[PGPS_K_COUNT] = {0};
};

// roctx range named after the reference's tf.name_scope of the same work (pssgp/kalman/parallel.py:122 "parallel_filter",
// pssgp/model.py:35 "merge_sorted", model.py:87 "make_model"): visible in `rocprofv3 --marker-trace`, a no-op otherwise
struct RoctxRange {
    explicit RoctxRange(const char* name) { pgps::dyn::range_push(name); }
    ~RoctxRange() { pgps::dyn::range_pop(); }
    RoctxRange(const RoctxRange&) = delete;
    RoctxRange& operator=(const RoctxRange&) = delete;
};

#define HIPCHK(ctx, expr)                                                        \
    do {                                                                         \
        hipError_t e_ = (expr);                                                  \
        if (e_ != hipSuccess) {                                                  \
            (ctx)->hip_err = std::string(#expr) + ": " + hipGetErrorString(e_);  \
            return PGPS_E_HIP;                                                   \
        }                                                                        \
    } while (0)

namespace pgps {

#ifndef PGPS_BLOCK
#define PGPS_BLOCK 256
#endif
constexpr int kBlock = PGPS_BLOCK;      // lanes per workgroup (experiments: make EXTRA=-DPGPS_BLOCK=128 ...)
constexpr int kWave = 64;
constexpr int kWaves = kBlock / kWave;
// fused path: series up to this length run as one workgroup, one launch (pgps_set_one_launch).  c1 (4096 + 1024 steps):
// one launch 82 us per predict_f against 61 us for the three launches, 40 against 38 us for the log-likelihood; N = 1000:
// 57 against 89 us -- profiles/r03_small_n_latency.txt
constexpr int kOneLaunchAuto = 2048;

int ensure(pgps_ctx* ctx, DevBuf& b, size_t bytes);
// all-gather of `bytes` per rank on the context's stream through the context's RCCL communicator (pgps_comm.hip)
int comm_allgather(pgps_ctx* ctx, const void* send, void* recv, size_t bytes);
int prof_flush(pgps_ctx* ctx);
void geometry(const pgps_ctx* ctx, long N, int* Lc, int* nblocks, int d = 0);

// Per-kernel timing: when the launch is sampled (pgps_profile_enable / _sample) the kernel goes out
// through hipExtLaunchKernelGGL, which stamps the two events with the dispatch's own start / end
// timestamps (the same clock rocprofv3's kernel trace reads), not with separate event packets.
pgps_ctx::EvPair* prof_acquire(pgps_ctx* ctx, int slot);

template <typename T>
struct ScanArgs {
    long N;                 // steps in this segment
    int Lc;                 // steps per lane
    int nblocks;            // workgroups
    long nlanes;            // nblocks * kBlock (stride of the field-major workspaces)
    // segment position in the whole series (single GPU: first = last = 1)
    int seg_first;          // step 0 of this segment is the first step of the series
    int seg_last;           // step N-1 of this segment is the last step of the series
    // model
    const T* P0;            // (d, d) prior covariance              [device]
    const T* H;             // (d,)   observation row               [device]
    T R;                    // observation noise variance
    // series
    const T* Fs;            // (N, d, d)
    const T* Qs;            // (N, d, d)
    const T* ys;            // (N,)   NaN = missing
    // stitching across segments (multi-GPU); unused when seg_first / seg_last
    const T* carry_in;      // (d + d*d) filtered (m, P full) entering this segment
    const T* halo_FQ;       // (2, d, d) F, Q of the first step of the NEXT segment
    const T* carry_back;    // (d + d*d) smoothed (m, P full) of the first step of the NEXT segment
    // outputs
    T* fms; T* fPs; T* sms; T* sPs;
    double* ll;             // scalar log-likelihood (of this segment)
    // workspace
    T* spine;               // (nblocks, NFILT)
    T* lpre;                // (NFILT, nlanes)
    T* sspine;              // (nblocks, NSMTH)
    T* lsuf;                // (NSMTH, nlanes)
    double* llpart;         // (nblocks,)
    int* status;            // != 0: a non-positive innovation variance was met
    // multi-GPU stitching (pgps_seg_*): records exchanged between ranks
    int rank, nranks;
    T* rec_f;               // out: this segment's filter record  [NFILT total | F_0 | Q_0]
    const T* gathered_f;    // in : (nranks, REC_F) all ranks' filter records
    T* rec_s;               // out: this segment's smoother record [NSMTH total | pad | ll partial (double)]
    const T* gathered_s;    // in : (nranks, REC_S)
    T* seg_ws;              // scratch: carry_in (d+d*d) | halo_FQ (2 d*d) | carry_back (d+d*d)
    int nt;                 // streaming stores for the outputs (pass larger than the Infinity Cache)
    // single-pass filter (k_filter_single): inter-workgroup look-back state, zeroed before every launch
    int* ticket;            // dynamic tile counter
    int* flags;             // (nblocks,) 0 = nothing, 1 = aggregate published, 2 = inclusive prefix published
    int ll_in_apply;        // filter-only calls: the apply kernel's last workgroup sums the log-likelihood (no finalize launch)
    T* incl;                // (nblocks, d + d(d+1)/2) inclusive (m, P) of the window-closing tiles
    int win;                // look-back window (tiles); <= kBlock
    long long* stamps;      // diagnostic build only (-DPGPS_STAMPS): (3 kernels, nblocks, 8) s_memtime stamps
    int shortcut;           // try the forgetting shortcut for the carry across workgroups (pgps_kernels.hip.h): set by the launch code where a
                            // workgroup spans >= 2048 steps and pgps_set_shortcut has not turned it off
    int dform;              // lsuf / sspine hold smoothing totals in innovation form (pgps_math.h smth_extend_u): the smoother adds the
                            // filtered moments of the step a total was applied at.  Whole-series pkfs of the lane-chunk kernels only
};

// record lengths (in elements of T) of the segment exchange
inline int seg_rec_f_len(int d) { return d * d + d + d * (d + 1) + d + 2 * d * d; }
inline int seg_rec_s_pad(int d) { int n = d * d + d + d * (d + 1) / 2; return n + (n & 1); }
inline int seg_rec_s_len(int d) { return seg_rec_s_pad(d) + 2; }

template <typename... KArgs, typename... Args>
inline void timed_launch(pgps_ctx* ctx, int slot, void (*kernel)(KArgs...), dim3 grid, dim3 block, unsigned shmem,
                         Args... args) {
    pgps_ctx::EvPair* ev = prof_acquire(ctx, slot);
    if (ev) hipExtLaunchKernelGGL(kernel, grid, block, shmem, ctx->stream, ev->a, ev->b, 0, KArgs(args)...);
    else hipLaunchKernelGGL(kernel, grid, block, shmem, ctx->stream, KArgs(args)...);
}

template <typename T>
struct GpModel {
    double lam;             // F = -lam I + N
    double N1[9];           // N            (d x d, row-major, d <= 3)
    double N2[9];           // N^2 / 2      (zero for d <= 2)
    double Pinf[9];         // stationary covariance = P0
    T H[3];
    const T* ts;            // (N,) time stamps                     [device]
    T t_prev;               // time before the first step (t0)
};

template <typename T>
struct GpArgs {
    ScanArgs<T> s;          // N, geometry, ys, R, outputs, scratch (Fs, Qs, P0, H unused)
    GpModel<T> m;
    // projected-posterior mode of the smoother (pgps_gp_predict_*): step k writes H sm_k and H sP_k H^T
    // to slot qslot[k] when that is >= 0 and nothing otherwise; sms / sPs are not written
    const int* qslot;       // (N,) or null                         [device]
    T* pmean;               // (K,)                                 [device]
    T* pvar;                // (K,)                                 [device]
};

template <typename T, int D>
int launch_gp(pgps_ctx* ctx, GpArgs<T> g, int want_filtered, int want_smoothed);

// filter + log-likelihood + smoother of a whole series in one resident launch (pgps_resident.hip.h, pgps_res_inst.hip)
template <typename T>
struct ResArgs {
    ScanArgs<T> s;          // N, nblocks, model, series, outputs, spine / sspine / llpart, status, ll
    GpModel<T> m;           // fused form: the Matern model (F = -lam I + N) and the time stamps
    int* bar;               // this launch's 8 counter shards (32 ints apart), zero on entry
    int* bar_next;          // the next launch's: zeroed by this one
    int* flags1;            // per workgroup: `epoch` once its filtering total is published (the neighbour hand-off)
    int* flags2;            // ... once its smoothing total and log-likelihood partial are
    int epoch;              // this launch's (never 0)
    long long* stamps;      // diagnostics (pgps_set_resident(ctx, 2)): (nblocks, 16) cycle stamps, else null
};
constexpr int kResLc = 16;                  // steps per lane: the chunk lives in registers
constexpr int kStatusBytes = 8192;          // the context's status buffer: word 0 flags, 16.. tickets, 512.. the resident kernel's barrier counters
constexpr int kResBarWord = 512;            // two sets of 8 shards x 32 ints
constexpr int kResFlagWord = 1024;          // two arrays of 256 per-workgroup hand-off flags (words 1024 .. 1535)
// does a whole-series filter + smoother call of N steps at dimension d (fp64 when !f32) take the resident launch?
bool resident_fits(const pgps_ctx* ctx, long N, int d, bool f32);
template <typename T, int D>
int launch_resident(pgps_ctx* ctx, ResArgs<T> ra, bool fused, bool smooth);
// log-likelihood and the model's adjoints on the fused path (pgps_gpadj.hip.h), fp64, d <= 3: out = 1 + d^2 + 2 d + 1 doubles [device]
// (T names the unit that holds the instantiation: call it with T = double)
template <typename T, int D>
int launch_gp_adj(pgps_ctx* ctx, GpArgs<double> g, double* out);

template <typename T>
struct GpBatchArgs {
    long N;
    int Lc, nblocks;
    long nlanes;
    const T* ts;
    const T* ys;
    T t_prev;
    const double* models;       // (B, kGpModelStride): lam | N1[9] | N2[9] | Pinf[9] | H[3] | R
    T* spine;                   // (B, nblocks, NFILT)
    T* lpre;                    // (B, NFILT, nlanes)
    double* llpart;             // (B, nblocks)
    double* ll;                 // (B,)
};
constexpr int kGpModelStride = 32;

template <typename T, int D>
int launch_gp_batch(pgps_ctx* ctx, int B, GpBatchArgs<T> b);

// merge of two sorted time arrays on the device + NaN marking of the query rows (pgps_core.hip)
template <typename T>
int launch_merge(pgps_ctx* ctx, long N, long K, const T* ts, const T* ys, const T* tq, T* ts_m, T* ys_m, int* qslot);

// log-likelihood + gradient on the fused path (pgps_grad.hip); all array pointers device
int launch_grad(pgps_ctx* ctx, long N, int d, int np, const double* model, const double* ts, double t0,
                const double* ys, double* out_dev);

// the same for block-nilpotent composite models (sums / products of Matern kernels), d = 2..6 (pgps_gradb.hip, one
// instantiation per d): model rows [lam (4) | N (d*d) | Pinf (d*d) | H (d) | R], bsize = the nblk block sizes
template <int D>
int launch_gradb(pgps_ctx* ctx, long N, int nblk, const int* bsize, int np, const double* model, const double* ts, double t0,
                 const double* ys, double* out_dev);

enum Mode { MODE_PKF, MODE_PKS, MODE_PKFS, MODE_SEG_REDUCE, MODE_SEG_FILTER, MODE_SEG_SMOOTHER };

// defined in pgps_inst.hip, one explicit instantiation per compiled (T, D)
template <typename T, int D>
int launch_scan(pgps_ctx* ctx, ScanArgs<T> a, Mode mode);
// the same scan built a second time with 128-lane workgroups (pgps_inst.hip with -DPGPS_NARROW -DPGPS_BLOCK=128)
template <typename T, int D>
int launch_scan_narrow(pgps_ctx* ctx, ScanArgs<T> a, Mode mode);
constexpr int kBlockNarrow = 128;
void geometry_narrow(const pgps_ctx* ctx, long N, int* Lc, int* nblocks, int d);      // pgps_core.hip
// wave-cooperative family (pgps_wc.hip): runtime state dimension, 1 <= d <= 32
template <typename T>
int launch_scan_wc(pgps_ctx* ctx, ScanArgs<T> a, int d, Mode mode);
// row-cooperative family (pgps_rc.hip.h): fp64 and fp32, 2 <= d <= 16
template <typename Real>
int launch_scan_rc(pgps_ctx* ctx, ScanArgs<Real> a, int d, Mode mode);
// the same with nothing written per step: MODE_PKF = log-likelihood only; MODE_PKFS = H sm, H sP H^T at the steps
// qslot marks (a.sPs / a.sms are then scratch of N d^2 / N d doubles for the smoothing elements)
// (MODE_PKF with a.Qs == nullptr: implicit process noise Q_k = P0 - F_k P0 F_k^T, P0 must be stationary)
int launch_scan_rc_proj(pgps_ctx* ctx, ScanArgs<double> a, int d, Mode mode, const int* qslot, double* pmean,
                        double* pvar);
int launch_disc_rc(pgps_ctx* ctx, long N, int d, const double* F, const double* Pinf, const double* ts, double t0,
                   double* Fs, double* Qs, int batch = 1, long bs_model = 0);
// log-likelihoods of `batch` models over one series: table = batch x [F | Pinf | H | R] (stride bs_model, device),
// Fs / Qs = (batch, N, d, d) discretised arrays, ll = (batch,) device
int launch_ll_batch_rc(pgps_ctx* ctx, long N, int d, int batch, const double* table, long bs_model, const double* Fs,
                       const double* Qs, const double* ys, double* ll);
namespace rc {
constexpr int kDimMin = 2, kDimMax = 16;
constexpr int kScanBlockedDimMax = 15;     // blocked scans of the chain totals: the dimensions whose kernels need no scratch
template <typename Real>
struct RcArgsT {
    long N;
    int Lw;                     // steps per chain
    long nchunk;                // chains
    long wfast;                 // waves [1, wfast) lie inside the series, halo step included: predicate-free body
    const Real *P0, *H;
    Real R;
    const Real *Fs, *Qs, *ys;
    Real *fms, *fPs, *sms, *sPs;
    Real* agg1;               // (nchunk, nfilt) chunk totals
    const Real* pre;          // (nchunk, nfilt) inclusive prefixes of agg1
    Real* sagg1;              // (nchunk, nsmth) smoothing totals
    const Real* suf;          // (nchunk, nsmth) inclusive suffixes of sagg1
    Real *Es, *gs;            // (N, d, d), (N, d) the smoothing elements' E and g: sPs / sms themselves (overwritten
                                // in place by the smoother) unless those are not there yet (segments, projections)
    Real* Lws;                // (N, d, d) the smoothing elements' L
    double* llpart;             // (nchunk,)
    int seg_first, seg_last;    // this launch covers the first / last step of the whole series (multi-GPU segments)
    const Real* carry;        // !seg_first: compact filter record of everything before the segment
    const Real *halo_F, *halo_Q;  // !seg_last: F, Q of the first step of the next segment
    const Real* carry_back;   // !seg_last: compact smoother record of everything after the segment
    int batch;                  // models evaluated over the same series (blockIdx.y); 0 / 1 = one
    long bs_F, bs_agg, bs_model;    // per-model strides of Fs / Qs, of agg1 / pre, of the model table
    const Real* Rs;           // batch entry point: observation noise of model b at Rs[b * bs_model] (else null)
    int implicit_q;             // Qs is not there: Q_k = P0 - F_k P0 F_k^T is folded into the predict (P0 stationary)
    int store_f;                // write fms / fPs (0: log-likelihood-only and projected-posterior calls)
    const int* qslot;           // projected-posterior mode: (N,) slot of step k in pmean / pvar, or -1
    Real *pmean, *pvar;       // (K,) H sm and H sP H^T at the query steps
    int dform;                  // the chains' smoothing totals are in innovation form (rc_apply1 DFORM): rc_smooth1 adds the filtered moments
    int quad;                   // level-1 kernels of the quad-cooperative family (pgps_qc.hip.h: fp32, 5 <= d <= 8, 16 chains per wave)
};
using RcArgs = RcArgsT<double>;
// defined in pgps_rc_inst.hip, one explicit instantiation per (scalar type, d)
template <typename Real, int D>
int launch_rc_level1(pgps_ctx* ctx, const RcArgsT<Real>& a, int phase);
template <typename Real, int D>
int launch_rc_ks(pgps_ctx* ctx, int which, long n, long stride, const Real* in, Real* out, int batch, long bstride,
                 const Real* fixed);
template <typename Real, int D>
int launch_rc_scan_blocked(pgps_ctx* ctx, int which, long n, Real* data, Real* scratch);
template <typename Real, int D>
int launch_rc_seg_carry(pgps_ctx* ctx, int which, const Real* gathered, int rank, int nranks, int reclen, Real* out);
template <int D>
int launch_rc_disc(pgps_ctx* ctx, long N, const double* F, const double* Pinf, const double* ts, double t0, double* Fs,
                   double* Qs, int batch, long bs_model);
}  // namespace rc
// quad-cooperative level-1 kernels (pgps_qc.hip.h): fp32, 5 <= d <= 8; they speak the row-cooperative protocol
namespace qc {
constexpr int kDimMin = 5, kDimMax = 8;
template <int D>
int launch_qc_level1(pgps_ctx* ctx, const rc::RcArgsT<float>& a, int phase);
}  // namespace qc
template <typename T>
int launch_disc_wc(pgps_ctx* ctx, long N, int d, const T* F, const T* Pinf, const T* ts, T t0, T* Fs, T* Qs);

template <typename T, int D>
int launch_disc(pgps_ctx* ctx, long N, const T* F, const T* Pinf, const T* ts, T t0, T* Fs, T* Qs);

}  // namespace pgps
