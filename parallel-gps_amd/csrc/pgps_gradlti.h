// pgps_gradlti.h -- adjoint gradient of the log-likelihood for general LTI models: the argument
// block of the two level-1 kernels it adds to each cooperative family and the launch functions the host driver
// (pgps_wc.hip: launch_ll_grad_lti) calls.  fp64 only.
//
// What the reference gets from TensorFlow autodiff through tfp.math.scan_associative (tests/test_gp_vs_kfs.py:53-78;
// pssgp/kalman/parallel.py:121-152 differentiated) is here one filter pass and one backward pass:
//   forward   Kalman pass over every chain from the scanned prefixes (as the log-likelihood call), filtered moments kept
//             in scratch; per step the ADJOINT ELEMENT  (E, g, L) = (A^T, v r / s, -v v^T / (2 s)),  A = (I - K H) F,
//             v = (H F)^T, is folded into the chain's total -- under the SMOOTHING operator (parallel.py:176-184): with
//             a = d ll / d m_k and W = d ll / d P_k - a a^T / 2 the reverse sweep is  a' = E a + g,  W' = E W E^T + L.
//   scan      suffix scan of the chain totals: the smoother's own scan kernels.
//   backward  per chain, from the suffix entering it: the model's adjoints
//               Abar = sum_k dt_k [mpbar mp^T + 2 Ppbar (Pp - Pinf)]    (contracts with every dF that commutes with F)
//               Ubar = sum_k ubar_k     (d ll / d Pinf = sym(Ubar H))
//               Hbar = sum_k sbar u + Pp ubar - rbar mp,    Rbar = sum_k sbar
//             as chain partials, summed in a fixed order by the finalize kernel.
// The host contracts them with d(F, Pinf, H, R)/d(theta) (pssgp/kernels/sde_grads.py): any number of hyper-parameters
// for the price of two passes.
#pragma once

#include "pgps_internal.h"

namespace pgps {

struct GradLtiArgs {
    long N;
    int d;
    int Lw;                     // steps per chain
    long nchunk;                // chains
    const double* Pinf;         // (d, d) stationary covariance = P0        [device]
    const double* H;            // (d,)
    double R;
    const double* Fs;           // (N, d, d) transition matrices (the process noise is implicit: Q_k = Pinf - F_k Pinf F_k^T)
    const double* ys;           // (N,) NaN = missing
    const double* ts;           // (N,)
    double t0;
    double* fms;                // (N, d)    filtered means        (scratch, written by the forward pass)
    double* fPs;                // (N, d, d) filtered covariances
    const double* pre;          // (nchunk, nfilt) inclusive prefixes of the filter totals
    double* sagg;               // (nchunk, nsmth) adjoint totals of the chains [E | L | g]
    const double* suf;          // (nchunk, nsmth) their inclusive suffixes
    double* llpart;             // (nchunk,)
    double* gpart;              // (nchunk, d d + 2 d + 1) chain partials [Abar | Ubar | Hbar | Rbar]
    double* out;                // [ll | Abar (d d, row-major) | Ubar (d) | Hbar (d) | Rbar]
};
__host__ __device__ inline int grad_lti_nstat(int d) { return d * d + 2 * d + 1; }

// out[0] = sum of the chains' log-likelihood partials, out[1 + e] = sum over the chains of entry e of their partials:
// one workgroup per entry, a fixed order of additions (bit-reproducible).  Shared by the families (a HIP translation unit each).
static __global__ __launch_bounds__(256) void k_grad_lti_finalize(long nchunk, int nst, const double* llpart, const double* gpart,
                                                                   double* out) {
    __shared__ double part[256];
    const int e = blockIdx.x;                   // 0: ll, 1 + e: statistic e
    const double* src = e == 0 ? llpart : gpart + (e - 1);
    const long stride = e == 0 ? 1 : nst;
    double t = 0.0;
    for (long c = threadIdx.x; c < nchunk; c += 256) t += src[c * stride];
    part[threadIdx.x] = t;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[e] = part[0];
}

namespace rc {
// phase 0: forward (rc_gapply1), 1: backward (rc_gback1), 2: finalize.  Defined in pgps_rc_inst.hip (fp64 units).
template <int D>
int launch_rc_grad(pgps_ctx* ctx, const GradLtiArgs& a, int phase);
}  // namespace rc

// ll and the model adjoints of an LTI model on the device (model = [F | Pinf | H] device, ts / ys device, out device:
// 1 + grad_lti_nstat(d) doubles).  2 <= d <= 32.  Defined in pgps_wc.hip.
int launch_ll_grad_lti(pgps_ctx* ctx, long N, int d, const double* model, double R, const double* ts, double t0,
                       const double* ys, double* out);
// The same on the wave-cooperative family (any d <= 32; the road of d = 17..32) from DISCRETISED arrays Fs, Qs (N, d, d)
// [device]: the caller runs the discretisation (pgps_core.hip: block-wise for block-diagonal models).
int launch_ll_grad_lti_wc(pgps_ctx* ctx, long N, int d, const double* model, double R, const double* Fs, const double* Qs,
                          const double* ts, double t0, const double* ys, double* out);

}  // namespace pgps
