// Issue cost of packed float32 arithmetic from ONE wave per SIMD (the occupancy of the d = 6 float32 lane-chunk kernels,
// 512 registers): v_fma_f32 against v_pk_fma_f32 (plain, and with a broadcast operand through op_sel), v_pk_mul_f32,
// v_pk_add_f32, independent streams of 32 accumulators; cycles per instruction from s_memtime around 131072 instructions.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/pk_f32.hip -o /tmp/pk_f32 && /tmp/pk_f32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, long long* cyc, float s0, float s1) {
    float a[32];
    v2f p[16];
#pragma unroll
    for (int i = 0; i < 32; ++i) a[i] = threadIdx.x * 0.001f + i;
#pragma unroll
    for (int i = 0; i < 16; ++i) p[i] = v2f{a[2 * i], a[2 * i + 1]};
    v2f m = v2f{s0, s1};
    float x = s0, y = s1;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 4096; ++it) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(p[i]) : "v"(m));
        } else if constexpr (MODE == 2) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %1, %1, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(p[i]) : "v"(m));
        } else if constexpr (MODE == 3) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(m));
        } else if constexpr (MODE == 4) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(m));
        } else if constexpr (MODE == 5) {            // dependent chain of v_fma_f32
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[0]) : "v"(x), "v"(y));
        } else if constexpr (MODE == 6) {            // dependent chain of v_pk_fma_f32
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(p[0]) : "v"(m));
        } else if constexpr (MODE == 7) {            // v_mov_b32 stream
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(x));
        } else if constexpr (MODE == 8) {            // v_pk_mov_b32
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_pk_mov_b32 %0, %1, %1" : "+v"(p[i]) : "v"(m));
        } else if constexpr (MODE == 9) {            // v_cndmask stream
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(x));
        } else if constexpr (MODE == 10) {           // v_fma_f64
            double* dd = reinterpret_cast<double*>(p);
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(dd[i]) : "v"(*reinterpret_cast<double*>(&m)));
        } else if constexpr (MODE == 11) {           // v_mov_b32 dpp row_shr
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
        } else if constexpr (MODE == 12) {           // v_fma_f32 with dpp? (v_fmac_f32_dpp)
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(x), "v"(y));
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) acc += a[i];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int MODE>
void run(const char* name, int threads, float* out, long long* cyc) {
    const int blocks = 256;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks * threads / 64);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += (double)v;
    s /= h.size();
    // s_memtime counts at 100 MHz: convert with the event time instead
    printf("%-34s %4d lanes/WG  memtime ticks/instr %6.2f  kernel %.1f us  -> %.2f ns per instruction per wave\n", name, threads, s / 131072.0, ms * 1e3, ms * 1e6 / 131072.0);
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 16 * 8);
    for (int threads : {256, 512, 1024}) {
        run<0>("v_fma_f32 x32 independent", threads, out, cyc);
        run<1>("v_pk_fma_f32 x16 independent", threads, out, cyc);
        run<2>("v_pk_fma_f32 op_sel broadcast", threads, out, cyc);
        run<3>("v_pk_mul_f32", threads, out, cyc);
        run<4>("v_pk_add_f32", threads, out, cyc);
        run<5>("v_fma_f32 dependent chain", threads, out, cyc);
        run<6>("v_pk_fma_f32 dependent chain", threads, out, cyc);
        run<7>("v_mov_b32", threads, out, cyc);
        run<8>("v_pk_mov_b32", threads, out, cyc);
        run<9>("v_cndmask_b32", threads, out, cyc);
        run<10>("v_fma_f64", threads, out, cyc);
        run<11>("v_mov_b32_dpp row_shr", threads, out, cyc);
        run<12>("v_fmac_f32_dpp", threads, out, cyc);
    }
    return 0;
}
