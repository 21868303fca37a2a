// Micro-benchmark: the memory access pattern of rc_smooth1 without its arithmetic.  A 16-lane row walks a chain
// of Lw records of D*D doubles backwards: per step it reads two records (E, L) in column layout (lane j reads
// element (i, j), i = 0..D-1: D loads of 8 bytes, 16 lanes contiguous) and writes one; four rows per wave, chains
// Lw records apart.  Reports the achieved bytes/s -- the ceiling the level-1 kernels of the row-cooperative
// family have with this layout -- next to a plain coalesced copy of the same volume.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int D>
__global__ __launch_bounds__(64) void rows(long N, int Lw, const double* E, const double* L, double* out) {
    const int lane = threadIdx.x & 15, row = threadIdx.x >> 4;
    const long c = (long)blockIdx.x * 4 + row;
    const long k0 = c * Lw;
    const bool lv = lane < D;
    double acc[D];
#pragma unroll
    for (int i = 0; i < D; ++i) acc[i] = 0.0;
    double e[D], l[D];
    auto load = [&](long k) {
        if (lv) {
#pragma unroll
            for (int i = 0; i < D; ++i) { e[i] = E[k * D * D + i * D + lane]; l[i] = L[k * D * D + i * D + lane]; }
        }
    };
    load(k0 + Lw - 1);
    for (int s = Lw - 1; s >= 0; --s) {
        const long k = k0 + s;
#pragma unroll
        for (int i = 0; i < D; ++i) acc[i] = acc[i] * 0.5 + e[i] + l[i];
        if (s > 0) load(k - 1);
        if (lv) {
#pragma unroll
            for (int i = 0; i < D; ++i) out[k * D * D + i * D + lane] = acc[i];
        }
    }
}

__global__ void copy3(long n, const double* a, const double* b, double* o) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) o[i] = a[i] + b[i];
}

template <int D>
static void run(long N, int Lw) {
    const size_t n = (size_t)N * D * D;
    double *E, *L, *O;
    hipMalloc(&E, n * 8); hipMalloc(&L, n * 8); hipMalloc(&O, n * 8);
    hipMemset(E, 0, n * 8); hipMemset(L, 0, n * 8);
    const long chains = N / Lw;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        rows<D><<<dim3((unsigned)(chains / 4)), 64>>>(N, Lw, E, L, O);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        copy3<<<4096, 256>>>((long)n, E, L, O);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms2; hipEventElapsedTime(&ms2, e0, e1);
    printf("D=%d N=%ld Lw=%d chains=%ld: row pattern %.3f ms = %.2f TB/s;  coalesced copy %.3f ms = %.2f TB/s\n", D, N, Lw, chains,
           ms, 3.0 * n * 8 / ms * 1e-9, ms2, 3.0 * n * 8 / ms2 * 1e-9);
    hipFree(E); hipFree(L); hipFree(O);
}

int main() {
    run<11>(1L << 20, 128);
    run<11>(1L << 20, 256);
    run<11>(1L << 20, 32);
    run<16>(1L << 19, 128);
    return 0;
}
