// What the CU's memory path delivers when a wave streams RECORDS the way the lane-chunk kernels do: 64 owners (lanes), each
// with a contiguous chunk of Lc records of RB bytes; per step the wave fetches record s of every owner as 16-byte pieces
// (piece q = v * 64 + lane -> owner q / NV, piece q % NV: consecutive lanes read consecutive pieces of one record, then the
// next owner's, `pitch` = Lc * RB bytes further).  Two arrays (F and Q), 128-lane workgroups, one wave per SIMD, one
// sub-tile of prefetch -- the float32 d = 6 geometry of config c3 is RB = 144, Lc = 16.  Reports useful bytes per second for
//   RB = 144 (a record straddles 128-byte lines: each line is fetched for two consecutive steps),
//   RB = 128 (line-exact), RB = 256, RB = 288 (two records per sub-tile), and a plain contiguous stream.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/rec_stream.hip -o tools/micro/rec_stream && tools/micro/rec_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float V4 __attribute__((ext_vector_type(4)));

template <int RB, bool CONTIG>
__global__ __launch_bounds__(128) void k(const char* __restrict__ F, const char* __restrict__ Q, float* out, int Lc) {
    constexpr int NV = RB / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long wbase = ((long)blockIdx.x * 2 + wave) * 64 * (long)Lc * RB;
    const long pitch = (long)Lc * RB;
    V4 acc = V4{0, 0, 0, 0};
    V4 r[2][NV], n[2][NV];
    auto issue = [&](int s, V4 (&d)[2][NV]) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int q = v * 64 + lane;
            const long off = CONTIG ? ((long)s * 64 * RB + (long)q * 16) : ((long)(q / NV) * pitch + (long)s * RB + (q % NV) * 16);
            d[0][v] = *reinterpret_cast<const V4*>(F + wbase + off);
            d[1][v] = *reinterpret_cast<const V4*>(Q + wbase + off);
        }
    };
    issue(0, r);
    for (int s = 0; s < Lc; ++s) {
        issue(s + 1 < Lc ? s + 1 : s, n);
#pragma unroll
        for (int v = 0; v < NV; ++v) { acc += r[0][v] * r[1][v]; }
#pragma unroll
        for (int v = 0; v < NV; ++v) { r[0][v] = n[0][v]; r[1][v] = n[1][v]; }
    }
    out[blockIdx.x * 128 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

template <int RB, bool CONTIG>
void run(const char* name, const char* F, const char* Q, float* out, int Lc, int blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<RB, CONTIG>), dim3(blocks), dim3(128), 0, 0, F, Q, out, Lc);
    hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<RB, CONTIG>), dim3(blocks), dim3(128), 0, 0, F, Q, out, Lc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double bytes = 2.0 * blocks * 128.0 * Lc * RB;
    printf("%-44s Lc %2d  %7.1f us  %6.2f TB/s useful (%.1f MB)\n", name, Lc, ms * 1e3, bytes / ms * 1e-9, bytes * 1e-6);
}

int main() {
    const size_t cap = (size_t)512 * 128 * 32 * 288 + 4096;
    char *F, *Q; float* out;
    hipMalloc(&F, cap); hipMalloc(&Q, cap); hipMalloc(&out, 512 * 128 * 4);
    hipMemset(F, 0, cap); hipMemset(Q, 0, cap);
    for (int blocks : {512, 1024}) {
        printf("-- %d workgroups of 128 lanes\n", blocks);
        run<144, false>("records of 144 B (c3), lane-owned chunks", F, Q, out, 16, blocks);
        run<128, false>("records of 128 B, lane-owned chunks", F, Q, out, 16, blocks);
        run<256, false>("records of 256 B, lane-owned chunks", F, Q, out, 16, blocks);
        run<288, false>("records of 288 B (two per sub-tile)", F, Q, out, 8, blocks);
        run<144, true>("the same bytes, contiguous per step", F, Q, out, 16, blocks);
        run<64, false>("records of 64 B (d = 4 float32)", F, Q, out, 16, blocks);
        run<32, false>("records of 32 B (d = 2 fp64)", F, Q, out, 16, blocks);
        run<32, false>("records of 32 B, Lc 32", F, Q, out, 32, blocks);
    }
    return 0;
}
