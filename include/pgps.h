/* pgps.h -- C ABI of libpgps.so: the parallel Kalman filter / RTS smoother scan of
 * EEA-sensors/parallel-gps (`pssgp`) as hand-written HIP kernels for MI355X (gfx950).
 *
 * The reference has no native boundary (it is pure Python on TensorFlow); the boundary it
 * does have is the Python signature of the functions below, so every entry point names the
 * reference function whose arguments and results it carries (paths relative to the
 * reference checkout).  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions
 *   - plain pointers and sizes only; row-major, time-major arrays: Fs[k][i][j].
 *   - `_f64` / `_f32` = the dtype every array of that call has (ll is always double).
 *   - host entry points take HOST pointers: the library stages to the device, runs, copies
 *     back and returns when the results are in the caller's buffers.
 *   - `_dev` entry points take DEVICE pointers (16-byte aligned) and are asynchronous on the
 *     context's stream (pgps_set_stream / pgps_synchronize).
 *   - the caller owns every buffer; the library owns only the context (stream, scratch).
 *   - return value 0 = PGPS_OK, negative = error (pgps_strerror); nothing throws across the
 *     ABI; outputs are undefined after an error.
 *   - one call in flight per context; contexts are independent (one per GPU / per thread).
 *   - state dimension d: 1..PGPS_MAX_DIM_LANE use the lane-chunk kernels (one lane owns whole d x d operands), fp64
 *     series with 5 <= d <= 16 and fp32 series with 7 <= d <= 16 the row-cooperative ones (built natively in both
 *     precisions for 2 <= d <= 16), fp32 series at d = 8 -- and at d = 6 away from 2^19 .. 2^20 steps -- the
 *     quad-cooperative level-1 kernels under the row-cooperative driver, up to PGPS_MAX_DIM the wave-cooperative ones
 *     (from d = 17 on with the two-rows level-1 kernels: a chain on two DPP rows, the products in registers);
 *     pgps_set_family overrides.
 *     Every call takes every d <= PGPS_MAX_DIM, the segment calls (pgps_seg_*, pgps_pkfs_seg_*) included.
 *   - NaN in `ys` marks a missing observation (parallel.py:42,86-95).
 */
#ifndef PGPS_H_
#define PGPS_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PGPS_OK 0
#define PGPS_E_INVALID (-1)         /* bad argument (null pointer, N < 1, misaligned device pointer) */
#define PGPS_E_UNSUPPORTED_DIM (-2) /* state dimension outside the compiled kernels */
#define PGPS_E_HIP (-3)             /* a HIP runtime call failed (pgps_last_hip_error) */
#define PGPS_E_NOMEM (-4)           /* device or host allocation failed */
#define PGPS_E_NUMERIC (-5)         /* non-finite result (reference: TF raises on CPU, NaNs on GPU) */
#define PGPS_E_NO_DEVICE (-6)       /* no HIP device visible */
#define PGPS_E_COMM (-7)            /* an RCCL call failed (pgps_last_hip_error carries RCCL's message) */

#define PGPS_MAX_DIM_LANE 6   /* lane-chunk kernels: one lane holds whole d x d operands */
#define PGPS_MAX_DIM 32       /* wave-cooperative kernels: operands in LDS, 64 lanes share each operation (levels 2, 3);
                                 level 1 from d = 17: two-rows kernels, a row of every operand per lane */

typedef struct pgps_ctx pgps_ctx;

/* ---- library / context ---------------------------------------------------------------- */
int pgps_version(void);
const char* pgps_strerror(int code);
int pgps_device_count(int* n);
/* Create a context on HIP device `device` (own non-blocking stream + scratch). */
int pgps_create(int device, pgps_ctx** out);
int pgps_destroy(pgps_ctx* ctx);
/* Launch on an external hipStream_t (e.g. the caller's framework stream).  NULL is the HIP null
 * (default) stream, exactly as in HIP; pgps_use_own_stream goes back to the context's own
 * non-blocking stream (the state after pgps_create). */
int pgps_set_stream(pgps_ctx* ctx, void* hip_stream);
int pgps_use_own_stream(pgps_ctx* ctx);
int pgps_synchronize(pgps_ctx* ctx);
/* Steps per lane of the scan kernels; 0 = automatic. */
int pgps_set_chunk(pgps_ctx* ctx, int steps_per_lane);
/* Single-pass filter kernel (d <= 2, 16 steps per lane, whole series on one GPU): mode -1 = automatic,
 * 0 = off (three-launch reduce-then-scan), 1 = on; window = tiles per look-back window (1..256, 0 = keep). */
int pgps_set_single_pass(pgps_ctx* ctx, int mode, int window);
/* Filter + log-likelihood + smoother of a whole series in ONE resident launch (csrc/pgps_resident.hip.h): fp64, d = 2,
 * series of up to 4096 steps per compute unit (2^20 on MI355X); taken by pgps_pkfs_dev_f64 / pgps_pkfs_f64 and by
 * pgps_gp_dev_f64 / pgps_gp_f64, and -- in its filter-only form, without the smoothing phase -- by pgps_pkf_dev_f64 /
 * pgps_pkf_f64 and by pgps_gp_* calls that ask for no smoothed moments.  Fs, Qs, ys are read once and every output is
 * written once (the reference's pkf + pks contract, pssgp/kalman/parallel.py:121-201); the launch needs every workgroup
 * resident, so another stream's kernel holding compute units makes it give up (bounded spins) with bit 1 of
 * pgps_status set and undefined outputs.  A call made while the stream is being captured into a hipGraph takes the three
 * launches (the resident launch's barrier set and hand-off epoch are per-launch host state: a replay would reuse them).
 * Series of up to 2048 steps per compute unit run it with 8 steps per lane (twice the workgroups), longer ones with 16;
 * pgps_set_chunk(ctx, 8 or 16) under mode >= 1 pins that choice (any other pinned chunk selects the three launches).
 * mode -1 = automatic (from 2^17 steps), 0 = never (three launches),
 * 1 = wherever the series fits, 2 = as 1 with in-kernel phase stamps kept for pgps_resident_stamps (diagnostics). */
int pgps_set_resident(pgps_ctx* ctx, int mode);
/* The forgetting shortcut of the lane-chunk and resident kernels (csrc/pgps_kernels.hip.h): a workgroup whose neighbour's total
 * has |A| (smoother: |E|) <= 2^-120 (fp64) / 2^-60 (fp32) -- a filter that has forgotten what came before those thousands of
 * steps -- takes the neighbour's (b, C) ((g, L)) as its carry instead of folding every total on that side: same bits, ten
 * combine levels less.  Decided from the data per workgroup; 1 (default) = where a workgroup spans >= 2048 steps, 0 = never
 * (every carry by the general fold: what the tests compare it with). */
int pgps_set_shortcut(pgps_ctx* ctx, int on);
/* Diagnostics: cycle stamps of the last resident launch made under mode 2: out = (n_blocks, 16) long long (host), at most
 * max_blocks rows copied; out may be NULL to ask for n_blocks only. */
int pgps_resident_stamps(pgps_ctx* ctx, long long* out, int max_blocks, int* n_blocks);
/* Kernel family: 0 = automatic (lane-chunk for d <= 4 and for fp32 up to PGPS_MAX_DIM_LANE; row-cooperative for
 * fp64 with 5 <= d <= 16 and fp32 with 7 <= d <= 16, segments use it above d = 6; quad-cooperative for whole fp32
 * series at d = 8 and, up to 3 * 2^17 and from 3 * 2^19 steps, at d = 6; wave-cooperative otherwise, d <= 32),
 * 1 = lane-chunk (d <= PGPS_MAX_DIM_LANE), 2 = wave-cooperative, 3 = row-cooperative (fp64 and fp32, 2 <= d <= 16),
 * 4 = quad-cooperative level-1 kernels under the row-cooperative driver (fp32 only, 5 <= d <= 8; pkf / pkfs / segments:
 * outside that range, for fp64 and for pks a forced family 4 returns PGPS_E_UNSUPPORTED_DIM -- until round 3 it fell
 * back to the row-cooperative kernels there). */
int pgps_set_family(pgps_ctx* ctx, int family);
/* Lanes per workgroup of the lane-chunk kernels (d <= PGPS_MAX_DIM_LANE): 0 = automatic (128 -- half the scan tree per
 * step at the same number of workgroups -- except d <= 3 from 2^22 steps of this call / this rank's segment), 128, 256.
 * The library carries both builds; the fused (pgps_gp_*) kernels always use 256.  Do not change it between the phases
 * of a segment pass (refused).  pgps_get_chunk reports the geometry of the 256-lane build. */
int pgps_set_block(pgps_ctx* ctx, int lanes);
/* Kalman pass of the 128-lane lane-chunk kernels at d = 2, fp64 (16 or 32 steps per lane): inputs through a ring of
 * LDS-DMA slots (buffer_load ... lds) requested before the workgroup folds the spine, instead of register staging.
 * -1 = automatic (= off: measured slower at 2^20 steps, profiles/r03_experiments.txt), 0 = off, 1 = on (the ring takes
 * 128 KiB of LDS: one workgroup per CU). */
int pgps_set_dma(pgps_ctx* ctx, int mode);
/* Scans over the chain totals of the row- and quad-cooperative families: -1 = automatic (blocked from 64 chains: log2(B)
 * levels per launch through LDS, B <= 64 records per workgroup), 0 = one launch per Kogge-Stone level, 1 = blocked. */
int pgps_set_rc_scan(pgps_ctx* ctx, int mode);
/* Fused (pgps_gp_*) calls of short series -- the reference's own lengths, N = 200 .. 10^4 per evaluation
 * (pssgp/experiments/toy_models/mcmc.py:55) -- run as ONE workgroup in ONE launch (reduce, scan, Kalman pass, smoother,
 * projection) up to max_steps steps: -1 = automatic (2048: at 4096 + 1024 steps the three launches are ahead again), 0 = never,
 * n > 0 = up to n steps. */
int pgps_set_one_launch(pgps_ctx* ctx, int max_steps);
/* pgps_gp_ll_grad_* at d <= 2: series up to max_steps take ONE derivative direction per model, the directions side by side
 * in the same launches (a Dual<1> scan tree is less than half of a Dual<3> one, and for a short series the tree's latency
 * is the whole cost); longer ones carry all directions in one dual number (the primal arithmetic is shared).
 * -1 = automatic (2^18 steps), 0 = never.  d = 3 always runs one direction per model. */
int pgps_set_grad_pack(pgps_ctx* ctx, long max_steps);
/* What a lane-chunk call (or one rank's segment) of N steps at state dimension d <= PGPS_MAX_DIM_LANE runs with: lanes per
 * workgroup (128 / 256), steps per lane, workgroups -- after pgps_set_block / pgps_set_chunk. */
int pgps_get_geometry(pgps_ctx* ctx, long N, int d, int* lanes, int* steps_per_lane, int* workgroups);
/* (Diagnostic, environment: PGPS_WC_SERIAL3=1 when a context is created makes the wave-cooperative family walk its
 * group totals with one wave instead of the Kogge-Stone scan -- the cross-check of tests/test_gpu_wavecoop.py.
 * PGPS_WC_ROWS2=<mask> selects, per level-1 kernel of d >= 17, the two-rows kernels (bit set) or the LDS-tile kernels
 * they replaced: 1 reduce, 2 Kalman pass without and 4 with the smoothing total, 8 smoother; default 15, 0 = the
 * LDS-tile kernels throughout -- the cross-check of tests/test_gpu_tworows.py.  The Kogge-Stone levels of the filter scan
 * follow bit 1; PGPS_WC_KS2=0 keeps the LDS-tile level on its own.) */
/* LDS staging of the lane-chunk kernels: -1 = automatic, 0 = off (direct global accesses),
 * 2 or 4 = steps per lane per staged sub-tile (2: fp64 only).  Tuning / A-B knob. */
int pgps_set_stage(pgps_ctx* ctx, int steps_per_subtile);
int pgps_get_chunk(pgps_ctx* ctx, long n_steps, int* steps_per_lane, int* n_workgroups);
/* Which kernel family a pkf (what = 0) / pks (1) / pkfs (2) call -- or a phase of the segment protocol (3) -- of n_steps at
 * state dimension d will run on under the context's settings (float32 arithmetic if f32 != 0; a float32 smoother call that
 * is promoted to fp64 arithmetic, pgps_set_f32_policy, takes the fp64 answer): what a benchmark needs to name the kernels it
 * timed without repeating the library's dispatch rule. */
#define PGPS_FAMILY_LANE 1          /* lane-chunk kernels, 256-lane workgroups (d <= 6) */
#define PGPS_FAMILY_WAVE 2          /* wave-cooperative LDS-tile kernels (d <= 32) */
#define PGPS_FAMILY_ROW 3           /* row-cooperative kernels (2 <= d <= 16) */
#define PGPS_FAMILY_QUAD 4          /* quad-cooperative level-1 kernels under the row-cooperative driver (fp32, 5 <= d <= 8) */
#define PGPS_FAMILY_TWO_ROWS 5      /* two-rows level-1 kernels under the wave-cooperative driver (17 <= d <= 23 fp64, <= 31 fp32) */
#define PGPS_FAMILY_LANE_NARROW 11  /* lane-chunk kernels, 128-lane workgroups */
#define PGPS_FAMILY_RESIDENT 12     /* filter + smoother in one resident launch (pkfs, fp64, d = 2, up to 4096 steps per CU: pgps_set_resident) */
int pgps_get_family(pgps_ctx* ctx, long n_steps, int d, int f32, int what, int* family);
const char* pgps_last_hip_error(pgps_ctx* ctx);
/* Diagnostic flags raised since the last call (synchronises; 0 = none; bit 1: a bounded spin of the single-pass filter
 * or of the resident launch's grid barrier gave up -- results of that pass are invalid; PGPS_STATUS_F32_PROMOTED: a float32 call ran in fp64
 * arithmetic, see pgps_set_f32_policy). */
#define PGPS_STATUS_F32_PROMOTED 4
int pgps_status(pgps_ctx* ctx, int* flags);
/* float32 series whose call runs a smoother (pgps_pkfs_f32, pgps_pks_f32 and their _dev forms).  The reference's own
 * benchmark grid -- np.linspace(0, 4, N), pssgp/experiments/toy_models/common.py:31-32, with --dtype float32,
 * speed_and_stability.py:68 -- is so dense at N >= 2^15 that float32 ARITHMETIC cannot hold 1e-3 on the smoothed moments
 * (pssgp/kalman/parallel.py:159-166 and sequential.py:57-61 alike rebuild a covariance from a cancellation behind an
 * ill-conditioned solve).  policy 0 (default): such calls probe a sample of the transition matrices on the device and,
 * where the grid is that dense, run in fp64 arithmetic on the float32 arrays (widened into scratch, results rounded
 * back; state dimensions above 16 always do) -- the arrays, the entry points and the tolerances stay the float32 ones;
 * 1: float32 arithmetic whatever the grid; 2: always fp64 arithmetic.  pgps_status tells which way the calls went.
 * Under policy 0 a float32 smoother call is NOT asynchronous: the host waits for the probe's verdict (an event on a
 * stream of the library's own, ordered behind everything already queued on the context's stream), so such a call must
 * not be made while the caller's stream is being captured into a hipGraph -- use policy 1 or 2 there -- and the outputs
 * of pgps_pks_* must not alias its inputs (a promoted call reads fms / fPs after the float32 pass may have written
 * sms / sPs). */
int pgps_set_f32_policy(pgps_ctx* ctx, int policy);

/* ---- device memory helpers (for hosts without a device-array library) ----------------- */
int pgps_malloc(pgps_ctx* ctx, size_t bytes, void** dptr);
int pgps_free(pgps_ctx* ctx, void* dptr);
int pgps_memcpy_h2d(pgps_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int pgps_memcpy_d2h(pgps_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);

/* ---- per-kernel timing (hipEvents on the context's stream) ----------------------------- */
#define PGPS_K_FILTER_REDUCE 0
#define PGPS_K_FILTER_APPLY 1
#define PGPS_K_SMOOTHER_REDUCE 2
#define PGPS_K_SMOOTHER_APPLY 3
#define PGPS_K_LL_FINALIZE 4
#define PGPS_K_DISCRETISE 5
#define PGPS_K_RESIDENT 6           /* the one-launch filter + smoother pass (pgps_set_resident) */
#define PGPS_K_COUNT 7
/* mask: bit i set = record a hipEvent pair around every launch of slot i; 0 = off. */
int pgps_profile_enable(pgps_ctx* ctx, int mask);
/* Time only every n-th launch of an enabled slot (default 1): keeps the events' own cost
 * (a few microseconds per pair) out of a throughput measurement. */
int pgps_profile_sample(pgps_ctx* ctx, int every_n);
/* Mean elapsed milliseconds of an empty hipEvent pair on the context's stream. */
int pgps_profile_calibrate(pgps_ctx* ctx, double* empty_pair_ms);
/* Synchronises, then returns accumulated milliseconds and launch counts per PGPS_K_* slot
 * since the last reset (arrays of PGPS_K_COUNT). */
int pgps_profile_read(pgps_ctx* ctx, double* total_ms, long* launches, int reset);
const char* pgps_kernel_name(int slot);

/* ---- LTI discretisation: pssgp/kernels/base.py:29-47 (_get_ssm) ------------------------
 * ts (N,), F (d,d), Pinf (d,d) -> Fs (N,d,d), Qs (N,d,d);  dt_0 = ts[0] - t0.
 * Qs = Pinf - Fs Pinf Fs^T (Pinf must be the stationary covariance of the SDE). */
int pgps_discretise_f64(pgps_ctx*, long N, int d, const double* F, const double* Pinf, const double* ts,
                        double t0, double* Fs, double* Qs);
int pgps_discretise_f32(pgps_ctx*, long N, int d, const float* F, const float* Pinf, const float* ts,
                        float t0, float* Fs, float* Qs);
int pgps_discretise_dev_f64(pgps_ctx*, long N, int d, const double* F, const double* Pinf, const double* ts,
                            double t0, double* Fs, double* Qs);
int pgps_discretise_dev_f32(pgps_ctx*, long N, int d, const float* F, const float* Pinf, const float* ts,
                            float t0, float* Fs, float* Qs);

/* ---- parallel filter: pssgp/kalman/parallel.py:121-152 (pkf) ---------------------------
 * LGSSM (P0 (d,d), Fs (N,d,d), Qs (N,d,d), H (1,d), R scalar) + observations ys (N,)
 *   -> fms (N,d), fPs (N,d,d), and the log-likelihood if `ll` != NULL
 * (return_loglikelihood=True, parallel.py:135-151).  m0 = 0 (parallel.py:125). */
int pgps_pkf_f64(pgps_ctx*, long N, int d, const double* P0, const double* Fs, const double* Qs,
                 const double* H, double R, const double* ys, double* fms, double* fPs, double* ll);
int pgps_pkf_f32(pgps_ctx*, long N, int d, const float* P0, const float* Fs, const float* Qs,
                 const float* H, float R, const float* ys, float* fms, float* fPs, double* ll);
int pgps_pkf_dev_f64(pgps_ctx*, long N, int d, const double* P0, const double* Fs, const double* Qs,
                     const double* H, double R, const double* ys, double* fms, double* fPs, double* ll);
int pgps_pkf_dev_f32(pgps_ctx*, long N, int d, const float* P0, const float* Fs, const float* Qs,
                     const float* H, float R, const float* ys, float* fms, float* fPs, double* ll);

/* ---- parallel smoother: pssgp/kalman/parallel.py:187-196 (pks) -------------------------
 * Fs, Qs + filtered fms (N,d), fPs (N,d,d) -> sms (N,d), sPs (N,d,d). */
int pgps_pks_f64(pgps_ctx*, long N, int d, const double* Fs, const double* Qs, const double* fms,
                 const double* fPs, double* sms, double* sPs);
int pgps_pks_f32(pgps_ctx*, long N, int d, const float* Fs, const float* Qs, const float* fms,
                 const float* fPs, float* sms, float* sPs);
int pgps_pks_dev_f64(pgps_ctx*, long N, int d, const double* Fs, const double* Qs, const double* fms,
                     const double* fPs, double* sms, double* sPs);
int pgps_pks_dev_f32(pgps_ctx*, long N, int d, const float* Fs, const float* Qs, const float* fms,
                     const float* fPs, float* sms, float* sPs);

/* ---- filter + smoother: pssgp/kalman/parallel.py:199-201 (pkfs) ------------------------
 * One fused three-launch pass.  Also returns the filtered moments and the log-likelihood
 * (the reference's pkfs drops them; StateSpaceGP runs the filter a second time for ll,
 * pssgp/model.py:113-117).  fms / fPs / ll may be NULL for the host entry points; the
 * device entry points need fms and fPs (the smoother reads them back). */
int pgps_pkfs_f64(pgps_ctx*, long N, int d, const double* P0, const double* Fs, const double* Qs,
                  const double* H, double R, const double* ys, double* fms, double* fPs, double* sms,
                  double* sPs, double* ll);
int pgps_pkfs_f32(pgps_ctx*, long N, int d, const float* P0, const float* Fs, const float* Qs,
                  const float* H, float R, const float* ys, float* fms, float* fPs, float* sms,
                  float* sPs, double* ll);
int pgps_pkfs_dev_f64(pgps_ctx*, long N, int d, const double* P0, const double* Fs, const double* Qs,
                      const double* H, double R, const double* ys, double* fms, double* fPs, double* sms,
                      double* sPs, double* ll);
int pgps_pkfs_dev_f32(pgps_ctx*, long N, int d, const float* P0, const float* Fs, const float* Qs,
                      const float* H, float R, const float* ys, float* fms, float* fPs, float* sms,
                      float* sPs, double* ll);

/* ---- fused path: times and observations in, posterior and log-likelihood out ------------------
 * Replaces the chain  _get_ssm (pssgp/kernels/base.py:29-47) -> pkf / pkfs (parallel.py:121-201) that
 * StateSpaceGP runs (pssgp/model.py:92-117) for SDEs whose drift is  F = -lam I + N  with N
 * nilpotent (every Matern kernel; d <= 3):  Fs[k] = exp(-lam dt)(I + dt N1 + dt^2 N2), N1 = N,
 * N2 = N^2/2, and Qs[k] = Pinf - Fs[k] Pinf Fs[k]^T are formed in registers inside the scan kernels,
 * so a pass reads (t, y) instead of the (N, d, d) arrays.  The small model (lam, N1, N2 (may be NULL
 * for d <= 2), Pinf (d,d), H (d)) is always passed from HOST memory; ts, ys and the outputs are
 * device pointers for pgps_gp_dev_* and host pointers for pgps_gp_*.
 *   fms/fPs == NULL and sms/sPs == NULL : log-likelihood only (nothing is written per step)
 *   sms/sPs == NULL                     : filter (pkf with return_loglikelihood=True)
 *   all given                           : filter + smoother (pkfs) + log-likelihood. */
int pgps_gp_dev_f64(pgps_ctx*, long N, int d, double lam, const double* N1, const double* N2, const double* Pinf,
                    const double* H, double R, const double* ts, double t0, const double* ys, double* fms,
                    double* fPs, double* sms, double* sPs, double* ll);
int pgps_gp_dev_f32(pgps_ctx*, long N, int d, double lam, const double* N1, const double* N2, const double* Pinf,
                    const double* H, double R, const float* ts, double t0, const float* ys, float* fms, float* fPs,
                    float* sms, float* sPs, double* ll);
int pgps_gp_f64(pgps_ctx*, long N, int d, double lam, const double* N1, const double* N2, const double* Pinf,
                const double* H, double R, const double* ts, double t0, const double* ys, double* fms, double* fPs,
                double* sms, double* sPs, double* ll);
int pgps_gp_f32(pgps_ctx*, long N, int d, double lam, const double* N1, const double* N2, const double* Pinf,
                const double* H, double R, const float* ts, double t0, const float* ys, float* fms, float* fPs,
                float* sms, float* sPs, double* ll);

/* ---- general LTI models on the device: any kernel, fp64, 2 <= d <= 32 ------------------------------
 * (row-cooperative kernels up to d = 16; 17..32 on the wave-cooperative kernels, with whole moments in device scratch
 * and the projection at the query rows by a separate kernel; the batch entry points cover d <= 16)
 * The chain  _get_ssm (pssgp/kernels/base.py:29-47) -> pkf / pkfs (parallel.py:121-201) that StateSpaceGP runs
 * (pssgp/model.py:92-117) for kernels without the closed-form discretisation (RBF, Periodic, sums, products),
 * with only the results leaving the GPU.  The model F (d,d), Pinf (d,d) (= P0, the stationary covariance: Qs =
 * Pinf - Fs Pinf Fs^T), H (d) is always passed from HOST memory; ts, ys, tq, mean, var, ll are host pointers for
 * the plain entry points and device pointers for the _dev ones.
 *   pgps_lti_ll_*      : discretise, filter, log-likelihood -- nothing is written per step
 *   pgps_lti_predict_* : merge of the sorted ts (N) and tq (K) exactly as _merge_sorted (model.py:15-55), missing
 *                        observations at the query rows, discretise, filter + smoother over the N + K steps,
 *                        mean[j] = H sm, var[j] = H sP H^T of query j; ll (nullable) = log-likelihood of the
 *                        training series.  N + K < 2^31. */
int pgps_lti_ll_f64(pgps_ctx*, long N, int d, const double* F, const double* Pinf, const double* H, double R,
                    const double* ts, const double* ys, double t0, double* ll);
int pgps_lti_ll_dev_f64(pgps_ctx*, long N, int d, const double* F, const double* Pinf, const double* H, double R,
                        const double* ts, const double* ys, double t0, double* ll);
int pgps_lti_predict_f64(pgps_ctx*, long N, long K, int d, const double* F, const double* Pinf, const double* H,
                         double R, const double* ts, const double* ys, double t0, const double* tq, double* mean,
                         double* var, double* ll);
int pgps_lti_predict_dev_f64(pgps_ctx*, long N, long K, int d, const double* F, const double* Pinf, const double* H,
                             double R, const double* ts, const double* ys, double t0, const double* tq, double* mean,
                             double* var, double* ll);

/* B models over the same series in one set of launches (hyper-parameter grids, HMC chains, multi-start optimisation
 * -- the realistic series, N = 1e3..1e5, are launch-latency bound one model at a time).  `models` is HOST memory,
 * B rows of [F (d*d) | Pinf (d*d) | H (d) | R]; ll receives B log-likelihoods (host pointer; device for _dev). */
int pgps_lti_ll_batch_f64(pgps_ctx*, int B, long N, int d, const double* models, const double* ts, const double* ys,
                          double t0, double* ll);
int pgps_lti_ll_batch_dev_f64(pgps_ctx*, int B, long N, int d, const double* models, const double* ts, const double* ys,
                              double t0, double* ll);

/* ---- predict_f on the device (fused path, d <= 3) --------------------------------------------------
 * StateSpaceGP.predict_f (pssgp/model.py:92-111) in one call: the sorted training times `ts` (N) and
 * sorted query times `tq` (K) are merged on the device exactly as _merge_sorted does (model.py:15-55:
 * on equal times the shorter array's point comes first), query rows carry missing observations, the
 * fused filter + smoother runs over the N + K steps and only  mean[j] = H sm,  var[j] = H sP H^T  of
 * query j are written -- 2 scalars per query instead of (d^2 + d) per merged step.  ll (nullable)
 * receives the log-likelihood of the training series (query rows contribute nothing).
 * Host pointers; the _dev form takes device pointers for ts, ys, tq, mean, var, ll.  N + K < 2^31. */
int pgps_gp_predict_f64(pgps_ctx*, long N, long K, int d, double lam, const double* N1, const double* N2,
                        const double* Pinf, const double* H, double R, const double* ts, const double* ys, double t0,
                        const double* tq, double* mean, double* var, double* ll);
int pgps_gp_predict_f32(pgps_ctx*, long N, long K, int d, double lam, const double* N1, const double* N2,
                        const double* Pinf, const double* H, double R, const float* ts, const float* ys, double t0,
                        const float* tq, float* mean, float* var, double* ll);
int pgps_gp_predict_dev_f64(pgps_ctx*, long N, long K, int d, double lam, const double* N1, const double* N2,
                            const double* Pinf, const double* H, double R, const double* ts, const double* ys,
                            double t0, const double* tq, double* mean, double* var, double* ll);
int pgps_gp_predict_dev_f32(pgps_ctx*, long N, long K, int d, double lam, const double* N1, const double* N2,
                            const double* Pinf, const double* H, double R, const float* ts, const float* ys,
                            double t0, const float* tq, float* mean, float* var, double* ll);

/* ---- batched log-likelihood (fused path, d <= 3) ---------------------------------------------------
 * B hyper-parameter settings evaluated over the SAME series in one pair of launches (HMC leapfrogs,
 * grid search, multi-start optimisation at the reference's typical N of 1e3..1e5: SURVEY.md section
 * 8f rank 3).  `models` (HOST memory): B consecutive blocks [lam | N1 (d*d) | N2 (d*d) | Pinf (d*d) |
 * H (d) | R].  ll: B doubles.  Each ll[m] is bit-identical to what pgps_gp_* returns for model m at
 * the same steps-per-lane geometry.  ts, ys, ll: host pointers; device pointers for the _dev form.
 * B <= 65535. */
int pgps_gp_ll_batch_f64(pgps_ctx*, int B, long N, int d, const double* models, const double* ts, double t0,
                         const double* ys, double* ll);
int pgps_gp_ll_batch_f32(pgps_ctx*, int B, long N, int d, const double* models, const float* ts, double t0,
                         const float* ys, double* ll);
int pgps_gp_ll_batch_dev_f64(pgps_ctx*, int B, long N, int d, const double* models, const double* ts, double t0,
                             const double* ys, double* ll);
int pgps_gp_ll_batch_dev_f32(pgps_ctx*, int B, long N, int d, const double* models, const float* ts, double t0,
                             const float* ys, double* ll);

/* ---- log-likelihood and its gradient (fused path, d <= 3, fp64) --------------------------------
 * What the reference gets from TensorFlow autodiff through the scan (tests/test_gp_vs_kfs.py:53-78;
 * SURVEY.md section 8f, rank 1): forward-mode dual numbers carried through every filtering element and
 * every application of the associative operator.  `model` (HOST memory) holds 1 + np blocks of
 * [lam | N1 (d*d) | Pinf (d*d) | H (d) | R]: block 0 the values, block p the partial derivatives with
 * respect to hyper-parameter p (np <= 3).  out[0] = log-likelihood, out[1..np] = its gradient.
 * ts, ys, out: host pointers for pgps_gp_ll_grad_f64, device pointers for the _dev form, whose `out`
 * must hold 1 + 3 np doubles (the tail is scratch for the d = 3 one-direction-per-pass schedule). */
int pgps_gp_ll_grad_f64(pgps_ctx*, long N, int d, int np, const double* model, const double* ts, double t0,
                        const double* ys, double* out);
int pgps_gp_ll_grad_dev_f64(pgps_ctx*, long N, int d, int np, const double* model, const double* ts, double t0,
                            const double* ys, double* out);

/* The same for composite kernels whose drift is block diagonal with blocks  F_b = -lam_b I + N_b,  N_b nilpotent --
 * sums and products of Matern kernels, balanced or not (kernels/base.py:130-244; a product of Matern kernels is one
 * block: lam = the sum of the factors', N = the Kronecker sum of theirs) -- state dimension 2 <= d <= 6, nblk <= 4
 * blocks of sizes bsize[] (sum d), np <= 16 hyper-parameters: the reference's gradient test kernels `Matern32 +
 * Matern52` and `Matern32 * Matern52` (tests/test_gp_vs_kfs.py:40-41,53-78) differentiated exactly, by dual numbers
 * through the scan, one direction per pass.  `model` (HOST memory): 1 + np rows of
 * [lam (4, unused ones 0) | N (d*d) | Pinf (d*d) | H (d) | R], row 0 the values, row p the partial derivatives with
 * respect to hyper-parameter p.  out[0] = log-likelihood, out[1..np] = gradient; the _dev form takes device pointers
 * for ts, ys, out, and its `out` must hold 1 + 3 np doubles (the tail is scratch). */
int pgps_gp_ll_grad_blocks_f64(pgps_ctx*, long N, int d, int nblk, const int* bsize, int np, const double* model,
                               const double* ts, double t0, const double* ys, double* out);
int pgps_gp_ll_grad_blocks_dev_f64(pgps_ctx*, long N, int d, int nblk, const int* bsize, int np, const double* model,
                                   const double* ts, double t0, const double* ys, double* out);

/* ---- one series sharded over several GPUs (contiguous time segments) -------------------
 * No reference equivalent (the reference is single-device).  Rank r of
 * `nranks` owns steps [r*N, (r+1)*N) -- every rank passes its own N -- and calls, in order:
 *   1. pgps_seg_filter_reduce_dev   -> rec_f  (pgps_seg_record_len: rec_filter elements)
 *      caller all-gathers rec_f over the ranks -> gathered_f (nranks, rec_filter)
 *   2. pgps_seg_filter_apply_dev    -> fms, fPs, rec_s (rec_smoother elements)
 *      caller all-gathers rec_s -> gathered_s (nranks, rec_smoother)
 *   3. pgps_seg_smoother_apply_dev  -> sms, sPs, ll (log-likelihood of the WHOLE series)
 * The three calls of one pass must use the same context, N and d (they share its scratch).
 * rec_f = [segment total (A, b, C, J, eta) | F, Q of the segment's first step];
 * rec_s = [segment total (E, g, L) | pad | the segment's log-likelihood as a double]. */
int pgps_seg_record_len(int d, int* rec_filter, int* rec_smoother);
int pgps_seg_filter_reduce_dev_f64(pgps_ctx*, long N, int d, int rank, int nranks, const double* P0,
                                   const double* Fs, const double* Qs, const double* H, double R,
                                   const double* ys, double* rec_f);
int pgps_seg_filter_reduce_dev_f32(pgps_ctx*, long N, int d, int rank, int nranks, const float* P0,
                                   const float* Fs, const float* Qs, const float* H, float R, const float* ys,
                                   float* rec_f);
int pgps_seg_filter_apply_dev_f64(pgps_ctx*, long N, int d, int rank, int nranks, const double* P0,
                                  const double* Fs, const double* Qs, const double* H, double R,
                                  const double* ys, const double* gathered_f, double* fms, double* fPs,
                                  double* rec_s);
int pgps_seg_filter_apply_dev_f32(pgps_ctx*, long N, int d, int rank, int nranks, const float* P0,
                                  const float* Fs, const float* Qs, const float* H, float R, const float* ys,
                                  const float* gathered_f, float* fms, float* fPs, float* rec_s);
int pgps_seg_smoother_apply_dev_f64(pgps_ctx*, long N, int d, int rank, int nranks, const double* Fs,
                                    const double* Qs, const double* fms, const double* fPs,
                                    const double* gathered_s, double* sms, double* sPs, double* ll);
int pgps_seg_smoother_apply_dev_f32(pgps_ctx*, long N, int d, int rank, int nranks, const float* Fs,
                                    const float* Qs, const float* fms, const float* fPs, const float* gathered_s,
                                    float* sms, float* sPs, double* ll);

/* The three calls of a pass keep state in the context's scratch between them (chain totals, their scans, for
 * d > 6 the stored smoothing elements): no other call on the same context, and no pgps_set_chunk, may come between
 * phase 1 and phase 3 of a pass.  The library checks it -- a phase whose predecessor was not the previous call on the
 * context with the same (N, d, rank, nranks) returns PGPS_E_INVALID instead of reading stale scratch. */

/* ---- the same pass with the exchange inside the library: RCCL over xGMI ----------------------------
 * The context owns the communicator (one process per GPU, one context per process).
 *   rank 0:      pgps_comm_get_unique_id(id)   -> hand the PGPS_COMM_ID_BYTES bytes to every rank (any transport)
 *   every rank:  pgps_comm_init(ctx, id, rank, nranks)        (collective: returns when all ranks have joined)
 *   every rank:  pgps_pkfs_seg_dev_*(ctx, N_local, d, ...)    per pass, as often as wanted
 *   every rank:  pgps_comm_destroy(ctx)                       (pgps_destroy does it too)
 * pgps_pkfs_seg_dev_* = pkfs (pssgp/kalman/parallel.py:199-201) + log-likelihood for the segment [rank's steps] of a
 * series sharded over the communicator's ranks in rank order: reduce -> ncclAllGather (segment totals + first-step
 * halo, (3d^2+3d+d(d+1)) scalars per rank) -> filter -> ncclAllGather (smoothing totals + log-likelihood partial) ->
 * smoother, ALL enqueued on the context's stream: no host synchronisation, no framework.  Arguments as
 * pgps_pkfs_dev_* with this rank's arrays; P0 is the prior of the WHOLE series (used by rank 0), ll receives the
 * log-likelihood of the whole series on every rank.  d <= PGPS_MAX_DIM, fp64 and fp32 natively.  nranks = 1 works
 * (RCCL copies locally).  pgps_comm_allgather_dev is the bare collective on the context's stream (bytes per rank). */
#define PGPS_COMM_ID_BYTES 128
int pgps_comm_get_unique_id(void* id);
int pgps_comm_init(pgps_ctx* ctx, const void* id, int rank, int nranks);
int pgps_comm_destroy(pgps_ctx* ctx);
int pgps_comm_info(pgps_ctx* ctx, int* rank, int* nranks);      /* nranks = 0: no communicator */
int pgps_comm_count(pgps_ctx* ctx, int* nranks, int* rank);     /* as RCCL reports them (ncclCommCount / ncclCommUserRank); rank may be NULL */
/* RCCL is loaded on first use of a pgps_comm_* call (the copy already in the process, else the loader's librccl.so.1):
 * buf <- what was loaded, or why nothing was (PGPS_E_COMM).  A context without a communicator never needs RCCL. */
int pgps_comm_library(char* buf, size_t n);
int pgps_comm_allgather_dev(pgps_ctx* ctx, const void* send, void* recv, size_t bytes_per_rank);
int pgps_pkfs_seg_dev_f64(pgps_ctx*, long N, int d, const double* P0, const double* Fs, const double* Qs, const double* H,
                          double R, const double* ys, double* fms, double* fPs, double* sms, double* sPs, double* ll);
int pgps_pkfs_seg_dev_f32(pgps_ctx*, long N, int d, const float* P0, const float* Fs, const float* Qs, const float* H,
                          float R, const float* ys, float* fms, float* fPs, float* sms, float* sPs, double* ll);

/* ---- sequential mode: pssgp/kalman/sequential.py:11-73 (kf, ks) ------------------------
 * StateSpaceGP(parallel=False).  Host arithmetic on HOST pointers, as in the reference (its
 * sequential mode is the CPU `tf.scan`).  No context needed.  mps / Pps (predicted moments,
 * return_predicted=True) and ll may be NULL in kf. */
int pgps_seq_kf_f64(long N, int d, const double* P0, const double* Fs, const double* Qs, const double* H,
                    double R, const double* ys, double* fms, double* fPs, double* ll, double* mps, double* Pps);
int pgps_seq_kf_f32(long N, int d, const float* P0, const float* Fs, const float* Qs, const float* H, float R,
                    const float* ys, float* fms, float* fPs, double* ll, float* mps, float* Pps);
int pgps_seq_ks_f64(long N, int d, const double* Fs, const double* ms, const double* Ps, const double* mps,
                    const double* Pps, double* sms, double* sPs);
int pgps_seq_ks_f32(long N, int d, const float* Fs, const float* ms, const float* Ps, const float* mps,
                    const float* Pps, float* sms, float* sPs);

/* ---- host helper: the balancing sweep of balance_ss (pssgp/kernels/math_utils.py:10-29, numba in the reference) ----
 * scale[d] = accumulated diagonal scaling after n_iter sweeps over F (d,d).  Host pointers, no context. */
int pgps_host_balance_f64(int d, const double* F, int n_iter, double* scale);

/* ---- a series kept on the device across calls (round 3) ------------------------------------------------------------
 * What the reference's drivers do thousands of times per run is evaluate ONE (ts, ys) at changing hyper-parameters
 * (StateSpaceGP.maximum_log_likelihood_objective / training_loss under L-BFGS and HMC: pssgp/model.py:113-117 called from
 * pssgp/experiments/sunspot/map.py:74-82, experiments/common.py:95-133) and predict on a fixed grid (predict_f,
 * pssgp/model.py:92-111 from toy_models/speed_and_stability.py:73-87).  A pgps_series holds ts, ys -- and the query grid
 * merged with them by _merge_sorted's rule (pssgp/model.py:15-55) -- on the device, so a call moves the model's scalars
 * in and the results out, nothing else.  fp64; the fused (Matern-family, d <= 3) model of pgps_gp_*: F = -lam I + N1,
 * N2 = N1^2 / 2, Pinf, H, R.  The handle belongs to its context (one in-flight call per context); destroy it before it. */
typedef struct pgps_series pgps_series;
int pgps_series_create_f64(pgps_ctx* ctx, long N, const double* ts, const double* ys, double t0, pgps_series** out);
int pgps_series_set_queries_f64(pgps_series* s, long K, const double* tq);     /* sorted query times; K = 0 drops them */
int pgps_series_info(pgps_series* s, long* N, long* K);
int pgps_series_destroy(pgps_series* s);
/* results are on the HOST when these return (they synchronise the context's stream) */
int pgps_series_gp_ll_f64(pgps_series* s, int d, double lam, const double* N1, const double* N2, const double* Pinf,
                          const double* H, double R, double* ll);
int pgps_series_gp_ll_grad_f64(pgps_series* s, int d, int np, const double* model, double* out /* 1 + np */);
int pgps_series_gp_predict_f64(pgps_series* s, int d, double lam, const double* N1, const double* N2, const double* Pinf,
                               const double* H, double R, double* mean /* K */, double* var /* K */, double* ll /* or NULL */);
/* Log-likelihood and the ADJOINTS of the fused model by one filter pass and one reverse pass (csrc/pgps_gpadj.hip.h; the
 * lane-chunk twin of pgps_lti_ll_grad_*): out = [ll | Abar (d d, row-major) | Ubar (d) | Hbar (d) | Rbar], 1 + d d + 2 d + 1
 * doubles, with  d ll / d theta = <Abar, dF> + Ubar^T dPinf H^T + Hbar . dH + Rbar dR  for every dF that commutes with F -- for
 * the Matern family: dF = -F / lengthscale, dPinf = Pinf / variance.  What the reference takes from tf.GradientTape over
 * maximum_log_likelihood_objective (tests/test_gp_vs_kfs.py:53-78; pssgp/kalman/parallel.py:121-152 differentiated), at
 * the cost of about two likelihoods whatever the number of hyper-parameters.  fp64, d <= 3.  The _dev form takes device
 * pointers ts, ys, out and is asynchronous on the context's stream. */
int pgps_series_gp_ll_grad_adj_f64(pgps_series* s, int d, double lam, const double* N1, const double* N2, const double* Pinf,
                                   const double* H, double R, double* out /* 1 + d d + 2 d + 1 */);
int pgps_gp_ll_grad_adj_dev_f64(pgps_ctx* ctx, long N, int d, double lam, const double* N1, const double* N2,
                                const double* Pinf, const double* H, double R, const double* ts, double t0, const double* ys,
                                double* out);
/* ... and for ANY kernel's LTI model (F, Pinf, H: host pointers; fp64, 2 <= d <= PGPS_MAX_DIM): pgps_lti_ll_f64 /
 * pgps_lti_predict_f64 / pgps_lti_ll_batch_f64 (models: B rows [F | Pinf | H | R], d <= 16) on the resident series and its
 * merged query grid -- the evaluation loop of an optimiser or sampler over an RBF / Periodic / composite kernel. */
int pgps_series_lti_ll_f64(pgps_series* s, int d, const double* F, const double* Pinf, const double* H, double R, double* ll);
int pgps_series_lti_predict_f64(pgps_series* s, int d, const double* F, const double* Pinf, const double* H, double R,
                                double* mean /* K */, double* var /* K */, double* ll /* or NULL */);
int pgps_series_lti_ll_batch_f64(pgps_series* s, int B, int d, const double* models, double* ll /* B */);

/* ---- log-likelihood AND its gradient for any kernel's LTI model, in two passes (round 4) ---------------------------
 * What the reference obtains from tf.GradientTape over maximum_log_likelihood_objective (tests/test_gp_vs_kfs.py:53-78;
 * pssgp/model.py:113-117 through pssgp/kalman/parallel.py:121-152 and pssgp/kernels/base.py:29-47), consumed by its
 * L-BFGS and HMC drivers (pssgp/experiments/sunspot/map.py:74-82, experiments/common.py:95-133): one filter pass and
 * one reverse (adjoint) pass return the adjoints of the MODEL, whatever the number of hyper-parameters:
 *   out[0]                    ll
 *   out[1 .. d d]             Abar (d, d) row-major = sum_k dt_k Fbar_k F_k^T : d ll / d theta gets <Abar, dF> for every
 *                             dF = dF/dtheta that commutes with F (d expm(dt F) = dt dF expm(dt F); true of every
 *                             hyper-parameter of every kernel of the reference in a frozen state basis -- time scalings of
 *                             block / Kronecker factors: pssgp/kernels/sde_grads.py)
 *   out[1 + d d ..]           Ubar (d):  d ll / d Pinf = sym(Ubar H), i.e. + Ubar^T dPinf H^T
 *   out[1 + d d + d ..]       Hbar (d):  + Hbar . dH
 *   out[1 + d d + 2 d]        Rbar    :  + Rbar dR
 * (1 + d d + 2 d + 1 doubles).  F, Pinf (stationary: F Pinf + Pinf F^T + L Q L^T = 0), H from HOST memory; ts, ys host
 * pointers for the plain entry point, device pointers for _dev (out device too: asynchronous on the context's stream);
 * the series entry point leaves `out` on the host.  fp64, 2 <= d <= PGPS_MAX_DIM. */
int pgps_lti_ll_grad_f64(pgps_ctx*, long N, int d, const double* F, const double* Pinf, const double* H, double R,
                         const double* ts, const double* ys, double t0, double* out);
int pgps_lti_ll_grad_dev_f64(pgps_ctx*, long N, int d, const double* F, const double* Pinf, const double* H, double R,
                             const double* ts, const double* ys, double t0, double* out);
int pgps_series_lti_ll_grad_f64(pgps_series* s, int d, const double* F, const double* Pinf, const double* H, double R,
                                double* out);

#ifdef __cplusplus
}
#endif
#endif /* PGPS_H_ */
