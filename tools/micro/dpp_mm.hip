// Micro-benchmark: 16-lane-row cooperative D x D fp64 product with DPP row_newbcast operands.
// Lane j of a row owns column j of every matrix (D registers); Z = X Y is D*D fmacs per lane:
//   Z_i += bcast_k(X_i) * Y_k.   Variant A: v_mov_b64_dpp + v_fma_f64 (compiler-managed hazards),
// variant B: v_fmac_f64_dpp in inline asm (one s_nop 1 at the head of each block).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

template <int K> __device__ __forceinline__ double bc(double x) {
    return __builtin_amdgcn_update_dpp(x, x, 0x150 + K, 0xf, 0xf, true);
}

template <int D, int K> struct RowA {
    static __device__ __forceinline__ void run(double& acc, double xi, const double* y) {
        acc = __builtin_fma(bc<K>(xi), y[K], acc);
        if constexpr (K + 1 < D) RowA<D, K + 1>::run(acc, xi, y);
    }
};
template <int D> __device__ __forceinline__ void mmA(const double* x, const double* y, double* z) {
#pragma unroll
    for (int i = 0; i < D; ++i) { double acc = 0.0; RowA<D, 0>::run(acc, x[i], y); z[i] = acc; }
}

// 4 rows x D columns in ONE asm statement (nothing can be scheduled into it): s_nop 4 covers the
// VALU-write -> DPP-read (2) and VALU-EXEC-write -> DPP (5) wait states at entry; inside, DPP sources are inputs only.
__device__ __forceinline__ void rows4_8(double& a0, double& a1, double& a2, double& a3, double x0, double x1, double x2, double x3, const double* y) {
    asm("s_nop 4\n"
        "v_fmac_f64_dpp %0, %4, %8 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %1, %5, %8 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %2, %6, %8 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %3, %7, %8 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %0, %4, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %1, %5, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %2, %6, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %3, %7, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %0, %4, %10 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %1, %5, %10 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %2, %6, %10 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %3, %7, %10 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %0, %4, %11 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %1, %5, %11 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %2, %6, %11 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %3, %7, %11 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %0, %4, %12 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %1, %5, %12 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %2, %6, %12 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %3, %7, %12 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %0, %4, %13 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %1, %5, %13 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %2, %6, %13 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %3, %7, %13 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %0, %4, %14 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %1, %5, %14 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %2, %6, %14 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %3, %7, %14 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %0, %4, %15 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %1, %5, %15 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %2, %6, %15 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %3, %7, %15 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]), "v"(y[6]), "v"(y[7]));
}
__device__ __forceinline__ void rows4_11(double& a0, double& a1, double& a2, double& a3, double x0, double x1, double x2, double x3, const double* y) {
    asm("s_nop 4\n"
        "v_fmac_f64_dpp %0, %4, %8 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %1, %5, %8 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %2, %6, %8 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %3, %7, %8 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %0, %4, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %1, %5, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %2, %6, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %3, %7, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %0, %4, %10 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %1, %5, %10 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %2, %6, %10 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %3, %7, %10 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %0, %4, %11 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %1, %5, %11 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %2, %6, %11 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %3, %7, %11 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %0, %4, %12 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %1, %5, %12 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %2, %6, %12 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %3, %7, %12 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %0, %4, %13 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %1, %5, %13 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %2, %6, %13 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %3, %7, %13 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %0, %4, %14 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %1, %5, %14 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %2, %6, %14 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %3, %7, %14 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %0, %4, %15 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %1, %5, %15 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %2, %6, %15 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %3, %7, %15 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %0, %4, %16 row_newbcast:8 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %1, %5, %16 row_newbcast:8 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %2, %6, %16 row_newbcast:8 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %3, %7, %16 row_newbcast:8 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %0, %4, %17 row_newbcast:9 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %1, %5, %17 row_newbcast:9 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %2, %6, %17 row_newbcast:9 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %3, %7, %17 row_newbcast:9 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %0, %4, %18 row_newbcast:10 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %1, %5, %18 row_newbcast:10 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %2, %6, %18 row_newbcast:10 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %3, %7, %18 row_newbcast:10 row_mask:0xf bank_mask:0xf\n"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]), "v"(y[6]), "v"(y[7]), "v"(y[8]), "v"(y[9]), "v"(y[10]));
}
__device__ __forceinline__ void rows4_12(double& a0, double& a1, double& a2, double& a3, double x0, double x1, double x2, double x3, const double* y) {
    asm("s_nop 4\n"
        "v_fmac_f64_dpp %0, %4, %8 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %1, %5, %8 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %2, %6, %8 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %3, %7, %8 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %0, %4, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %1, %5, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %2, %6, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %3, %7, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %0, %4, %10 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %1, %5, %10 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %2, %6, %10 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %3, %7, %10 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %0, %4, %11 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %1, %5, %11 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %2, %6, %11 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %3, %7, %11 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %0, %4, %12 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %1, %5, %12 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %2, %6, %12 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %3, %7, %12 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %0, %4, %13 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %1, %5, %13 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %2, %6, %13 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %3, %7, %13 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %0, %4, %14 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %1, %5, %14 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %2, %6, %14 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %3, %7, %14 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %0, %4, %15 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %1, %5, %15 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %2, %6, %15 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %3, %7, %15 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %0, %4, %16 row_newbcast:8 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %1, %5, %16 row_newbcast:8 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %2, %6, %16 row_newbcast:8 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %3, %7, %16 row_newbcast:8 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %0, %4, %17 row_newbcast:9 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %1, %5, %17 row_newbcast:9 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %2, %6, %17 row_newbcast:9 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %3, %7, %17 row_newbcast:9 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %0, %4, %18 row_newbcast:10 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %1, %5, %18 row_newbcast:10 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %2, %6, %18 row_newbcast:10 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %3, %7, %18 row_newbcast:10 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %0, %4, %19 row_newbcast:11 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %1, %5, %19 row_newbcast:11 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %2, %6, %19 row_newbcast:11 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %3, %7, %19 row_newbcast:11 row_mask:0xf bank_mask:0xf\n"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]), "v"(y[6]), "v"(y[7]), "v"(y[8]), "v"(y[9]), "v"(y[10]), "v"(y[11]));
}
__device__ __forceinline__ void rows4_16(double& a0, double& a1, double& a2, double& a3, double x0, double x1, double x2, double x3, const double* y) {
    asm("s_nop 4\n"
        "v_fmac_f64_dpp %0, %4, %8 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %1, %5, %8 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %2, %6, %8 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %3, %7, %8 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %0, %4, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %1, %5, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %2, %6, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %3, %7, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %0, %4, %10 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %1, %5, %10 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %2, %6, %10 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %3, %7, %10 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %0, %4, %11 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %1, %5, %11 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %2, %6, %11 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %3, %7, %11 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %0, %4, %12 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %1, %5, %12 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %2, %6, %12 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %3, %7, %12 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %0, %4, %13 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %1, %5, %13 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %2, %6, %13 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %3, %7, %13 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %0, %4, %14 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %1, %5, %14 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %2, %6, %14 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %3, %7, %14 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %0, %4, %15 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %1, %5, %15 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %2, %6, %15 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %3, %7, %15 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %0, %4, %16 row_newbcast:8 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %1, %5, %16 row_newbcast:8 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %2, %6, %16 row_newbcast:8 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %3, %7, %16 row_newbcast:8 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %0, %4, %17 row_newbcast:9 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %1, %5, %17 row_newbcast:9 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %2, %6, %17 row_newbcast:9 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %3, %7, %17 row_newbcast:9 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %0, %4, %18 row_newbcast:10 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %1, %5, %18 row_newbcast:10 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %2, %6, %18 row_newbcast:10 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %3, %7, %18 row_newbcast:10 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %0, %4, %19 row_newbcast:11 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %1, %5, %19 row_newbcast:11 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %2, %6, %19 row_newbcast:11 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %3, %7, %19 row_newbcast:11 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %0, %4, %20 row_newbcast:12 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %1, %5, %20 row_newbcast:12 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %2, %6, %20 row_newbcast:12 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %3, %7, %20 row_newbcast:12 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %0, %4, %21 row_newbcast:13 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %1, %5, %21 row_newbcast:13 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %2, %6, %21 row_newbcast:13 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %3, %7, %21 row_newbcast:13 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %0, %4, %22 row_newbcast:14 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %1, %5, %22 row_newbcast:14 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %2, %6, %22 row_newbcast:14 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %3, %7, %22 row_newbcast:14 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %0, %4, %23 row_newbcast:15 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %1, %5, %23 row_newbcast:15 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %2, %6, %23 row_newbcast:15 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %3, %7, %23 row_newbcast:15 row_mask:0xf bank_mask:0xf\n"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]), "v"(y[6]), "v"(y[7]), "v"(y[8]), "v"(y[9]), "v"(y[10]), "v"(y[11]), "v"(y[12]), "v"(y[13]), "v"(y[14]), "v"(y[15]));
}
template <int D> __device__ __forceinline__ void mmB(const double* x, const double* y, double* z) {
    constexpr int DR = (D + 3) / 4 * 4;
    double xx[DR], zz[DR];
#pragma unroll
    for (int i = 0; i < DR; ++i) { xx[i] = i < D ? x[i] : 0.0; zz[i] = 0.0; }
#pragma unroll
    for (int i = 0; i < DR; i += 4) {
        if constexpr (D == 8) rows4_8(zz[i], zz[i + 1], zz[i + 2], zz[i + 3], xx[i], xx[i + 1], xx[i + 2], xx[i + 3], y);
        if constexpr (D == 11) rows4_11(zz[i], zz[i + 1], zz[i + 2], zz[i + 3], xx[i], xx[i + 1], xx[i + 2], xx[i + 3], y);
        if constexpr (D == 12) rows4_12(zz[i], zz[i + 1], zz[i + 2], zz[i + 3], xx[i], xx[i + 1], xx[i + 2], xx[i + 3], y);
        if constexpr (D == 16) rows4_16(zz[i], zz[i + 1], zz[i + 2], zz[i + 3], xx[i], xx[i + 1], xx[i + 2], xx[i + 3], y);
    }
#pragma unroll
    for (int i = 0; i < D; ++i) z[i] = zz[i];
}

__device__ __forceinline__ void rows4p_11(double& a0, double& a1, double& a2, double& a3, double x0, double x1, double x2, double x3, const double* y) {
    asm("s_nop 4\n"
        "v_fmac_f64_e32 %0, %4, %8\n"
        "v_fmac_f64_e32 %1, %5, %8\n"
        "v_fmac_f64_e32 %2, %6, %8\n"
        "v_fmac_f64_e32 %3, %7, %8\n"
        "v_fmac_f64_e32 %0, %4, %9\n"
        "v_fmac_f64_e32 %1, %5, %9\n"
        "v_fmac_f64_e32 %2, %6, %9\n"
        "v_fmac_f64_e32 %3, %7, %9\n"
        "v_fmac_f64_e32 %0, %4, %10\n"
        "v_fmac_f64_e32 %1, %5, %10\n"
        "v_fmac_f64_e32 %2, %6, %10\n"
        "v_fmac_f64_e32 %3, %7, %10\n"
        "v_fmac_f64_e32 %0, %4, %11\n"
        "v_fmac_f64_e32 %1, %5, %11\n"
        "v_fmac_f64_e32 %2, %6, %11\n"
        "v_fmac_f64_e32 %3, %7, %11\n"
        "v_fmac_f64_e32 %0, %4, %12\n"
        "v_fmac_f64_e32 %1, %5, %12\n"
        "v_fmac_f64_e32 %2, %6, %12\n"
        "v_fmac_f64_e32 %3, %7, %12\n"
        "v_fmac_f64_e32 %0, %4, %13\n"
        "v_fmac_f64_e32 %1, %5, %13\n"
        "v_fmac_f64_e32 %2, %6, %13\n"
        "v_fmac_f64_e32 %3, %7, %13\n"
        "v_fmac_f64_e32 %0, %4, %14\n"
        "v_fmac_f64_e32 %1, %5, %14\n"
        "v_fmac_f64_e32 %2, %6, %14\n"
        "v_fmac_f64_e32 %3, %7, %14\n"
        "v_fmac_f64_e32 %0, %4, %15\n"
        "v_fmac_f64_e32 %1, %5, %15\n"
        "v_fmac_f64_e32 %2, %6, %15\n"
        "v_fmac_f64_e32 %3, %7, %15\n"
        "v_fmac_f64_e32 %0, %4, %16\n"
        "v_fmac_f64_e32 %1, %5, %16\n"
        "v_fmac_f64_e32 %2, %6, %16\n"
        "v_fmac_f64_e32 %3, %7, %16\n"
        "v_fmac_f64_e32 %0, %4, %17\n"
        "v_fmac_f64_e32 %1, %5, %17\n"
        "v_fmac_f64_e32 %2, %6, %17\n"
        "v_fmac_f64_e32 %3, %7, %17\n"
        "v_fmac_f64_e32 %0, %4, %18\n"
        "v_fmac_f64_e32 %1, %5, %18\n"
        "v_fmac_f64_e32 %2, %6, %18\n"
        "v_fmac_f64_e32 %3, %7, %18\n"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]), "v"(y[6]), "v"(y[7]), "v"(y[8]), "v"(y[9]), "v"(y[10]));
}
__device__ __forceinline__ void rows4p_12(double& a0, double& a1, double& a2, double& a3, double x0, double x1, double x2, double x3, const double* y) {
    asm("s_nop 4\n"
        "v_fmac_f64_e32 %0, %4, %8\n"
        "v_fmac_f64_e32 %1, %5, %8\n"
        "v_fmac_f64_e32 %2, %6, %8\n"
        "v_fmac_f64_e32 %3, %7, %8\n"
        "v_fmac_f64_e32 %0, %4, %9\n"
        "v_fmac_f64_e32 %1, %5, %9\n"
        "v_fmac_f64_e32 %2, %6, %9\n"
        "v_fmac_f64_e32 %3, %7, %9\n"
        "v_fmac_f64_e32 %0, %4, %10\n"
        "v_fmac_f64_e32 %1, %5, %10\n"
        "v_fmac_f64_e32 %2, %6, %10\n"
        "v_fmac_f64_e32 %3, %7, %10\n"
        "v_fmac_f64_e32 %0, %4, %11\n"
        "v_fmac_f64_e32 %1, %5, %11\n"
        "v_fmac_f64_e32 %2, %6, %11\n"
        "v_fmac_f64_e32 %3, %7, %11\n"
        "v_fmac_f64_e32 %0, %4, %12\n"
        "v_fmac_f64_e32 %1, %5, %12\n"
        "v_fmac_f64_e32 %2, %6, %12\n"
        "v_fmac_f64_e32 %3, %7, %12\n"
        "v_fmac_f64_e32 %0, %4, %13\n"
        "v_fmac_f64_e32 %1, %5, %13\n"
        "v_fmac_f64_e32 %2, %6, %13\n"
        "v_fmac_f64_e32 %3, %7, %13\n"
        "v_fmac_f64_e32 %0, %4, %14\n"
        "v_fmac_f64_e32 %1, %5, %14\n"
        "v_fmac_f64_e32 %2, %6, %14\n"
        "v_fmac_f64_e32 %3, %7, %14\n"
        "v_fmac_f64_e32 %0, %4, %15\n"
        "v_fmac_f64_e32 %1, %5, %15\n"
        "v_fmac_f64_e32 %2, %6, %15\n"
        "v_fmac_f64_e32 %3, %7, %15\n"
        "v_fmac_f64_e32 %0, %4, %16\n"
        "v_fmac_f64_e32 %1, %5, %16\n"
        "v_fmac_f64_e32 %2, %6, %16\n"
        "v_fmac_f64_e32 %3, %7, %16\n"
        "v_fmac_f64_e32 %0, %4, %17\n"
        "v_fmac_f64_e32 %1, %5, %17\n"
        "v_fmac_f64_e32 %2, %6, %17\n"
        "v_fmac_f64_e32 %3, %7, %17\n"
        "v_fmac_f64_e32 %0, %4, %18\n"
        "v_fmac_f64_e32 %1, %5, %18\n"
        "v_fmac_f64_e32 %2, %6, %18\n"
        "v_fmac_f64_e32 %3, %7, %18\n"
        "v_fmac_f64_e32 %0, %4, %19\n"
        "v_fmac_f64_e32 %1, %5, %19\n"
        "v_fmac_f64_e32 %2, %6, %19\n"
        "v_fmac_f64_e32 %3, %7, %19\n"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]), "v"(y[6]), "v"(y[7]), "v"(y[8]), "v"(y[9]), "v"(y[10]), "v"(y[11]));
}
__device__ __forceinline__ void rows4p_16(double& a0, double& a1, double& a2, double& a3, double x0, double x1, double x2, double x3, const double* y) {
    asm("s_nop 4\n"
        "v_fmac_f64_e32 %0, %4, %8\n"
        "v_fmac_f64_e32 %1, %5, %8\n"
        "v_fmac_f64_e32 %2, %6, %8\n"
        "v_fmac_f64_e32 %3, %7, %8\n"
        "v_fmac_f64_e32 %0, %4, %9\n"
        "v_fmac_f64_e32 %1, %5, %9\n"
        "v_fmac_f64_e32 %2, %6, %9\n"
        "v_fmac_f64_e32 %3, %7, %9\n"
        "v_fmac_f64_e32 %0, %4, %10\n"
        "v_fmac_f64_e32 %1, %5, %10\n"
        "v_fmac_f64_e32 %2, %6, %10\n"
        "v_fmac_f64_e32 %3, %7, %10\n"
        "v_fmac_f64_e32 %0, %4, %11\n"
        "v_fmac_f64_e32 %1, %5, %11\n"
        "v_fmac_f64_e32 %2, %6, %11\n"
        "v_fmac_f64_e32 %3, %7, %11\n"
        "v_fmac_f64_e32 %0, %4, %12\n"
        "v_fmac_f64_e32 %1, %5, %12\n"
        "v_fmac_f64_e32 %2, %6, %12\n"
        "v_fmac_f64_e32 %3, %7, %12\n"
        "v_fmac_f64_e32 %0, %4, %13\n"
        "v_fmac_f64_e32 %1, %5, %13\n"
        "v_fmac_f64_e32 %2, %6, %13\n"
        "v_fmac_f64_e32 %3, %7, %13\n"
        "v_fmac_f64_e32 %0, %4, %14\n"
        "v_fmac_f64_e32 %1, %5, %14\n"
        "v_fmac_f64_e32 %2, %6, %14\n"
        "v_fmac_f64_e32 %3, %7, %14\n"
        "v_fmac_f64_e32 %0, %4, %15\n"
        "v_fmac_f64_e32 %1, %5, %15\n"
        "v_fmac_f64_e32 %2, %6, %15\n"
        "v_fmac_f64_e32 %3, %7, %15\n"
        "v_fmac_f64_e32 %0, %4, %16\n"
        "v_fmac_f64_e32 %1, %5, %16\n"
        "v_fmac_f64_e32 %2, %6, %16\n"
        "v_fmac_f64_e32 %3, %7, %16\n"
        "v_fmac_f64_e32 %0, %4, %17\n"
        "v_fmac_f64_e32 %1, %5, %17\n"
        "v_fmac_f64_e32 %2, %6, %17\n"
        "v_fmac_f64_e32 %3, %7, %17\n"
        "v_fmac_f64_e32 %0, %4, %18\n"
        "v_fmac_f64_e32 %1, %5, %18\n"
        "v_fmac_f64_e32 %2, %6, %18\n"
        "v_fmac_f64_e32 %3, %7, %18\n"
        "v_fmac_f64_e32 %0, %4, %19\n"
        "v_fmac_f64_e32 %1, %5, %19\n"
        "v_fmac_f64_e32 %2, %6, %19\n"
        "v_fmac_f64_e32 %3, %7, %19\n"
        "v_fmac_f64_e32 %0, %4, %20\n"
        "v_fmac_f64_e32 %1, %5, %20\n"
        "v_fmac_f64_e32 %2, %6, %20\n"
        "v_fmac_f64_e32 %3, %7, %20\n"
        "v_fmac_f64_e32 %0, %4, %21\n"
        "v_fmac_f64_e32 %1, %5, %21\n"
        "v_fmac_f64_e32 %2, %6, %21\n"
        "v_fmac_f64_e32 %3, %7, %21\n"
        "v_fmac_f64_e32 %0, %4, %22\n"
        "v_fmac_f64_e32 %1, %5, %22\n"
        "v_fmac_f64_e32 %2, %6, %22\n"
        "v_fmac_f64_e32 %3, %7, %22\n"
        "v_fmac_f64_e32 %0, %4, %23\n"
        "v_fmac_f64_e32 %1, %5, %23\n"
        "v_fmac_f64_e32 %2, %6, %23\n"
        "v_fmac_f64_e32 %3, %7, %23\n"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]), "v"(y[6]), "v"(y[7]), "v"(y[8]), "v"(y[9]), "v"(y[10]), "v"(y[11]), "v"(y[12]), "v"(y[13]), "v"(y[14]), "v"(y[15]));
}
template <int D> __device__ __forceinline__ void mmC(const double* x, const double* y, double* z) {
    constexpr int DR = (D + 3) / 4 * 4;
    double xx[DR], zz[DR];
#pragma unroll
    for (int i = 0; i < DR; ++i) { xx[i] = i < D ? x[i] : 0.0; zz[i] = 0.0; }
#pragma unroll
    for (int i = 0; i < DR; i += 4) {
        if constexpr (D == 11) rows4p_11(zz[i], zz[i + 1], zz[i + 2], zz[i + 3], xx[i], xx[i + 1], xx[i + 2], xx[i + 3], y);
        if constexpr (D == 12) rows4p_12(zz[i], zz[i + 1], zz[i + 2], zz[i + 3], xx[i], xx[i + 1], xx[i + 2], xx[i + 3], y);
        if constexpr (D == 16) rows4p_16(zz[i], zz[i + 1], zz[i + 2], zz[i + 3], xx[i], xx[i + 1], xx[i + 2], xx[i + 3], y);
    }
#pragma unroll
    for (int i = 0; i < D; ++i) z[i] = zz[i];
}

template <int D, int V>
__global__ __launch_bounds__(256) void kern(const double* X, const double* Y, double* Z, int iters) {
    const int lane = threadIdx.x & 15;
    const long row = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 4;
    double x[D], y[D], z[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        x[i] = lane < D ? X[row * D * D + i * D + lane] : 0.0;
        y[i] = lane < D ? Y[row * D * D + i * D + lane] : 0.0;
    }
    for (int it = 0; it < iters; ++it) {
        if (V == 0) mmA<D>(x, y, z); else if (V == 1) mmB<D>(x, y, z); else mmC<D>(x, y, z);
        if (it + 1 < iters) {
#pragma unroll
            for (int i = 0; i < D; ++i) { y[i] = z[i] * 0.25; }
        }
    }
#pragma unroll
    for (int i = 0; i < D; ++i) if (lane < D) Z[row * D * D + i * D + lane] = z[i];
}

template <int D, int V>
static void run(const char* name, int blocks = 1024, int threads = 256) {
    const int rows = blocks * threads / 16;
    const int iters = 200;
    std::vector<double> X((size_t)rows * D * D), Y(X.size()), Z(X.size()), Zr(X.size());
    srand(1);
    for (auto& v : X) v = (rand() / (double)RAND_MAX - 0.5) * 0.5;
    for (auto& v : Y) v = rand() / (double)RAND_MAX - 0.5;
    double *dX, *dY, *dZ;
    hipMalloc(&dX, X.size() * 8); hipMalloc(&dY, X.size() * 8); hipMalloc(&dZ, X.size() * 8);
    hipMemcpy(dX, X.data(), X.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dY, Y.data(), X.size() * 8, hipMemcpyHostToDevice);
    // correctness with 3 chained products on a few rows
    kern<D, V><<<blocks, threads>>>(dX, dY, dZ, 3);
    hipMemcpy(Z.data(), dZ, X.size() * 8, hipMemcpyDeviceToHost);
    double maxerr = 0;
    for (int r = 0; r < 64 && V != 2; ++r) {
        std::vector<double> y(Y.begin() + (size_t)r * D * D, Y.begin() + (size_t)(r + 1) * D * D), z(D * D);
        const double* x = &X[(size_t)r * D * D];
        for (int it = 0; it < 3; ++it) {
            for (int i = 0; i < D; ++i) for (int j = 0; j < D; ++j) {
                double acc = 0; for (int k = 0; k < D; ++k) acc = fma(x[i * D + k], y[k * D + j], acc); z[i * D + j] = acc; }
            if (it < 2) for (int e = 0; e < D * D; ++e) y[e] = z[e] * 0.25;
        }
        for (int e = 0; e < D * D; ++e) maxerr = fmax(maxerr, fabs(z[e] - Z[(size_t)r * D * D + e]));
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kern<D, V><<<blocks, threads>>>(dX, dY, dZ, iters);
    hipEventRecord(e0);
    kern<D, V><<<blocks, threads>>>(dX, dY, dZ, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fma_lane = (double)blocks * threads * iters * D * D;      // lane-FMAs issued (incl. idle lanes)
    const double useful = (double)rows * iters * D * D * D;
    printf("[%d x %d] %s D=%d: maxerr %.3e  %.3f ms  issued %.2f T lane-FMA/s (peak 39.3)  useful %.2f TFLOP/s\n", blocks, threads, name, D, maxerr, ms,
           fma_lane / ms * 1e-9, 2 * useful / ms * 1e-9);
    hipFree(dX); hipFree(dY); hipFree(dZ);
}

int main() {
    for (int wps = 1; wps <= 4; wps *= 2) {
        run<12, 0>("mov_dpp+fma ", 256 * wps, 256);
        run<12, 1>("fmac_dpp asm", 256 * wps, 256);
        run<12, 2>("plain fmac  ", 256 * wps, 256);
    }
    run<11, 1>("fmac_dpp asm", 1024, 64);
    run<16, 1>("fmac_dpp asm", 1024, 64);
    run<16, 2>("plain fmac  ", 1024, 64);
    return 0;
}
