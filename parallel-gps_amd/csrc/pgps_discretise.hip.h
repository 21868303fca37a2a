// pgps_discretise.hip.h -- batched LTI discretisation (replaces pssgp/kernels/base.py:29-47).
//
// One lane per time step:  dt_k = t_k - t_{k-1} (t_{-1} = t0, base.py:34-35),
//   Fs[k] = expm(dt_k F)                       (base.py:36; Pade-13 scaling and squaring, the
//                                               algorithm behind tf.linalg.expm / scipy expm)
//   Qs[k] = Pinf - Fs[k] Pinf Fs[k]^T          (stationary form of base.py:39-46, valid because
//                                               F Pinf + Pinf F^T + L Q L^T = 0; checked on the host)
// The arithmetic is fp64 whatever the storage type T, so fp32 series get correctly rounded
// Fs / Qs (the difference Pinf - F Pinf F^T cancels badly in fp32).
#pragma once

#include <hip/hip_runtime.h>

#include "pgps_math.h"

namespace pgps {

template <int D>
__device__ __forceinline__ void expm_pade13(const double* A, double* R) {
    constexpr int MAT = D * D;
    const double b[14] = {64764752532480000., 32382376266240000., 7771770303897600., 1187353796428800.,
                          129060195264000., 10559470521600., 670442572800., 33522128640.,
                          1323241920., 40840800., 960960., 16380., 182., 1.};
    // 1-norm and scaling
    double nrm = 0.0;
#pragma unroll
    for (int j = 0; j < D; ++j) {
        double c = 0.0;
#pragma unroll
        for (int i = 0; i < D; ++i) c += fabs(A[i * D + j]);
        nrm = fmax(nrm, c);
    }
    int s = 0;
    if (nrm > 5.371920351148152) {
        s = (int)ceil(log2(nrm / 5.371920351148152));
        if (s < 0) s = 0;
        if (s > 60) s = 60;
    }
    const double sc = ldexp(1.0, -s);
    double As[MAT], A2[MAT], A4[MAT], A6[MAT], W[MAT], U[MAT], V[MAT];
#pragma unroll
    for (int i = 0; i < MAT; ++i) As[i] = A[i] * sc;
    mat_mul<double, D>(As, As, A2);
    mat_mul<double, D>(A2, A2, A4);
    mat_mul<double, D>(A4, A2, A6);
#pragma unroll
    for (int i = 0; i < MAT; ++i) W[i] = b[13] * A6[i] + b[11] * A4[i] + b[9] * A2[i];
    mat_mul<double, D>(A6, W, V);       // V used as scratch
#pragma unroll
    for (int i = 0; i < MAT; ++i) W[i] = V[i] + b[7] * A6[i] + b[5] * A4[i] + b[3] * A2[i];
#pragma unroll
    for (int i = 0; i < D; ++i) W[i * D + i] += b[1];
    mat_mul<double, D>(As, W, U);
#pragma unroll
    for (int i = 0; i < MAT; ++i) W[i] = b[12] * A6[i] + b[10] * A4[i] + b[8] * A2[i];
    mat_mul<double, D>(A6, W, V);
#pragma unroll
    for (int i = 0; i < MAT; ++i) V[i] += b[6] * A6[i] + b[4] * A4[i] + b[2] * A2[i];
#pragma unroll
    for (int i = 0; i < D; ++i) V[i * D + i] += b[0];
    // R = (V - U)^-1 (V + U)
    double M[MAT];
#pragma unroll
    for (int i = 0; i < MAT; ++i) { M[i] = V[i] - U[i]; R[i] = V[i] + U[i]; }
    gj_solve<double, D, D, true>(M, R);
    for (int q = 0; q < s; ++q) {
        mat_mul<double, D>(R, R, W);
#pragma unroll
        for (int i = 0; i < MAT; ++i) R[i] = W[i];
    }
}

template <typename T, int D>
__global__ __launch_bounds__(256) void k_discretise(long N, const T* __restrict__ F, const T* __restrict__ Pinf,
                                                    const T* __restrict__ ts, T t_prev, T* __restrict__ Fs,
                                                    T* __restrict__ Qs) {
    constexpr int MAT = D * D;
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= N) return;
    const T tk = ts[k];
    const T tp = (k > 0) ? ts[k - 1] : t_prev;
    const double dt = double(tk - tp);
    double A[MAT], P[MAT], E[MAT], X[MAT];
#pragma unroll
    for (int i = 0; i < MAT; ++i) { A[i] = dt * double(F[i]); P[i] = double(Pinf[i]); }
    expm_pade13<D>(A, E);
    mat_mul<double, D>(E, P, X);
    T Fo[MAT], Qo[MAT];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) {
            double acc = 0.0, acct = 0.0;
#pragma unroll
            for (int l = 0; l < D; ++l) { acc += X[i * D + l] * E[j * D + l]; acct += X[j * D + l] * E[i * D + l]; }
            const double q = 0.5 * (P[i * D + j] + P[j * D + i]) - 0.5 * (acc + acct);
            Qo[i * D + j] = T(q);
            Fo[i * D + j] = T(E[i * D + j]);
        }
#pragma unroll
    for (int i = 0; i < MAT; ++i) { Fs[k * MAT + i] = Fo[i]; Qs[k * MAT + i] = Qo[i]; }
}

}  // namespace pgps
