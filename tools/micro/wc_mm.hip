// Micro-benchmark: the LDS-tile product of the wave-cooperative family (pgps_wc.hip mm_acc) in isolation.
// One wave per workgroup, a GR x GR lane grid of TS x TS register tiles, operands in LDS (leading dimension LD).
// Reports clocks per product for a number of waves per CU (set through the dynamic LDS size).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/wc_mm.hip -o tools/micro/wc_mm && tools/micro/wc_mm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ void wsync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int DP, int GR, int LD, int KU, int MODE, int VAR>
__global__ __launch_bounds__(64) void k_mm(int reps, double* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TS = DP / GR;
    double* A = reinterpret_cast<double*>(smem);
    double* B = A + DP * LD;
    double* B2 = B + DP * LD;
    double* C = B2 + DP * LD;
    const int lane = threadIdx.x;
    const int lr = (VAR == 3) ? lane % GR : lane / GR, lc = (VAR == 3) ? lane / GR : lane % GR;
    const bool act = lane < GR * GR;
    const int r0 = lr * TS, c0 = lc * TS;
    for (int e = lane; e < DP * LD; e += 64) { A[e] = 1.0 / (1 + e); B[e] = 0.5 / (2 + e); B2[e] = 0.25; C[e] = 0; }
    wsync();
    double t[TS][TS], t2[TS][TS];
#pragma unroll
    for (int i = 0; i < TS; ++i)
#pragma unroll
        for (int j = 0; j < TS; ++j) { t[i][j] = 0; t2[i][j] = 0; }
    for (int r = 0; r < reps; ++r) {
        if (act) {
            if (VAR == 5 || VAR == 6) {
                // software pipeline: the operands of the next group of KU inner indices are requested before the
                // multiply-adds of the current one
                double av[2][KU][TS], bv[2][KU][TS];
                auto load = [&](int buf, int k0) {
#pragma unroll
                    for (int kk = 0; kk < KU; ++kk) {
                        const int k = k0 + kk;
#pragma unroll
                        for (int i = 0; i < TS; ++i) av[buf][kk][i] = (MODE == 2) ? A[k * LD + r0 + i] : A[(r0 + i) * LD + k];
#pragma unroll
                        for (int j = 0; j < TS; ++j) bv[buf][kk][j] = (MODE == 1) ? B[(c0 + j) * LD + k] : B[k * LD + c0 + j];
                    }
                };
                load(0, 0);
#pragma unroll
                for (int g = 0; g < DP / KU; ++g) {
                    if (g + 1 < DP / KU) load((g + 1) & 1, (g + 1) * KU);
                    if (VAR == 6) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int kk = 0; kk < KU; ++kk)
#pragma unroll
                        for (int i = 0; i < TS; ++i)
#pragma unroll
                            for (int j = 0; j < TS; ++j) t[i][j] += av[g & 1][kk][i] * bv[g & 1][kk][j];
                    if (VAR == 6) __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
            for (int k0 = 0; k0 < DP; k0 += KU) {
#pragma unroll
                for (int kk = 0; kk < KU; ++kk) {
                    const int k = k0 + kk;
                    double av[TS], bv[TS], b2[TS];
#pragma unroll
                    for (int i = 0; i < TS; ++i) av[i] = (MODE == 2) ? A[k * LD + r0 + i] : A[(r0 + i) * LD + k];
#pragma unroll
                    for (int j = 0; j < TS; ++j) bv[j] = (MODE == 1) ? B[(c0 + j) * LD + k] : B[k * LD + c0 + j];
                    if (VAR == 2) {
#pragma unroll
                        for (int j = 0; j < TS; ++j) b2[j] = B2[k * LD + c0 + j];
                    }
#pragma unroll
                    for (int i = 0; i < TS; ++i)
#pragma unroll
                        for (int j = 0; j < TS; ++j) {
                            t[i][j] += av[i] * bv[j];
                            if (VAR == 2) t2[i][j] += av[i] * b2[j];
                        }
                }
            }
            }
            if (VAR != 4) {
#pragma unroll
                for (int i = 0; i < TS; ++i)
#pragma unroll
                    for (int j = 0; j < TS; ++j) C[(r0 + i) * LD + c0 + j] = t[i][j] + t2[i][j];
            }
        }
        wsync();
        double* s = A; A = C; C = s;            // the next product reads what this one wrote
    }
    if (act) out[blockIdx.x * 64 + lane] = t[0][0] + t2[TS - 1][TS - 1] + A[r0 * LD + c0];
}

template <int DP, int GR, int LD, int KU, int MODE, int VAR>
static void run(const char* name, int waves_per_cu) {
    constexpr int TS = DP / GR;
    const int reps = 400, cus = 256;
    const size_t lds_floor = (size_t)4 * DP * LD * 8;
    size_t lds = (size_t)160 * 1024 / waves_per_cu;
    lds = lds / 256 * 256;
    if (lds < lds_floor) { printf("%-34s %d waves/CU: needs %zu B\n", name, waves_per_cu, lds_floor); return; }
    if (waves_per_cu > 1 && lds * (waves_per_cu + 1) <= 160 * 1024) lds += 0;
    auto kern = k_mm<DP, GR, LD, KU, MODE, VAR>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    double* out;
    hipMalloc(&out, (size_t)cus * waves_per_cu * 64 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<cus * waves_per_cu, 64, lds>>>(10, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<<<cus * waves_per_cu, 64, lds>>>(reps, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double us_per = ms * 1e3 / reps;
    const int macs = (VAR == 2 ? 2 : 1);
    printf("%-34s %d waves/CU  %.3f us per product-step (%.0f clk at 2.4 GHz) per %d product(s); fma floor %d clk\n", name,
           waves_per_cu, us_per, us_per * 2400.0, macs, macs * TS * TS * DP * 4);
    hipFree(out);
}

int main() {
    for (int w : {1, 4, 8}) {
        run<18, 6, 20, 3, 0, 0>("d18 A B      LD 20", w);
        run<18, 6, 20, 3, 1, 0>("d18 A B^T    LD 20", w);
        run<18, 6, 21, 3, 0, 0>("d18 A B      LD 21", w);
        run<18, 6, 21, 3, 1, 0>("d18 A B^T    LD 21", w);
        run<18, 6, 20, 3, 0, 2>("d18 A [B B2] LD 20", w);
        run<18, 6, 20, 3, 0, 3>("d18 A B col-major lanes", w);
        run<18, 6, 20, 3, 0, 4>("d18 A B no store", w);
        run<18, 6, 20, 3, 0, 5>("d18 A B pipelined 3", w);
        run<18, 6, 20, 6, 0, 5>("d18 A B pipelined 6", w);
        run<18, 6, 20, 3, 0, 6>("d18 A B pipelined 3 sched", w);
        run<18, 6, 20, 6, 0, 6>("d18 A B pipelined 6 sched", w);
        run<18, 6, 20, 6, 1, 6>("d18 A B^T pipelined 6 sched", w);
        run<24, 8, 26, 4, 0, 6>("d24 A B pipelined 4 sched", w);
        run<32, 8, 34, 4, 0, 6>("d32 A B pipelined 4 sched", w);
        run<24, 8, 26, 4, 0, 0>("d24 A B      LD 26", w);
        run<24, 8, 26, 4, 1, 0>("d24 A B^T    LD 26", w);
        run<32, 8, 34, 4, 0, 0>("d32 A B      LD 34", w);
    }
    return 0;
}
