// Micro-benchmark + functional check of the LDS-DMA ring that feeds k_filter_apply<double, 2> (DMA build).
//
// A wave owns 64 lanes x Lc consecutive steps of (Fs, Qs, ys) -> (P, m), exactly the traffic of the Kalman pass at
// d = 2 fp64: per 4-step sub-tile 16 x 1 KiB `buffer_load_dwordx4 ... lds` pieces (F, Q) into a ring of K slots, the
// lane-owned records read back with ds_read_b128 through an XOR swizzle applied on the SOURCE address (the DMA's LDS
// side is lane-linear), results written into the consumed slot and drained with 16-byte stores.  `work` dependent
// FMAs per step stand in for the Kalman arithmetic, `fold` for the spine fold in front of the loop (the prefetch is
// issued before it).  Output is checked on the host, so the addressing (M0 beyond 64 KiB included) is verified, and
// the achieved bytes/s for K = 1 .. 3 say what ring depth buys.
//
//   hipcc --offload-arch=gfx950 -O3 -o dma_ring dma_ring.hip && ./dma_ring
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kWave = 64, G = 4, SEG = 128, ARR = kWave * SEG, SLOT = 2 * ARR;
using V4 = __attribute__((ext_vector_type(4))) unsigned int;

__device__ __forceinline__ void wait_vm(int n) {
    switch (n) {
#define C(i) case i: asm volatile("s_waitcnt vmcnt(" #i ")" ::: "memory"); break;
        C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15) C(16) C(17) C(18) C(19) C(20)
        C(21) C(22) C(23) C(24) C(25) C(26) C(27) C(28) C(29) C(30) C(31) C(32) C(33) C(34) C(35) C(36) C(37) C(38) C(39)
        C(40) C(41) C(42) C(43) C(44) C(45) C(46) C(47) C(48) C(49) C(50) C(51) C(52) C(53) C(54) C(55) C(56) C(57) C(58)
        C(59) C(60) C(61) C(62)
#undef C
        default: asm volatile("s_waitcnt vmcnt(63)" ::: "memory"); break;
    }
}

// eight 1 KiB pieces of one array of one sub-tile: piece v covers owners 8v .. 8v+7
__device__ __forceinline__ void dma8(__amdgpu_buffer_rsrc_t rs, unsigned voff0, unsigned voff1, unsigned soff, unsigned stride,
                                     unsigned lds) {
    unsigned keep, so;
    asm volatile(
        "s_mov_b32 %[keep], m0\n\t"
        "s_mov_b32 m0, %[lds]\n\t"
        "s_mov_b32 %[so], %[soff]\n\t"
        "s_nop 4\n\t"
        "buffer_load_dwordx4 %[v0], %[rs], %[so] offen lds\n\t"
        "s_add_u32 %[so], %[so], %[stride]\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
        "buffer_load_dwordx4 %[v1], %[rs], %[so] offen lds\n\t"
        "s_add_u32 %[so], %[so], %[stride]\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
        "buffer_load_dwordx4 %[v0], %[rs], %[so] offen lds\n\t"
        "s_add_u32 %[so], %[so], %[stride]\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
        "buffer_load_dwordx4 %[v1], %[rs], %[so] offen lds\n\t"
        "s_add_u32 %[so], %[so], %[stride]\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
        "buffer_load_dwordx4 %[v0], %[rs], %[so] offen lds\n\t"
        "s_add_u32 %[so], %[so], %[stride]\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
        "buffer_load_dwordx4 %[v1], %[rs], %[so] offen lds\n\t"
        "s_add_u32 %[so], %[so], %[stride]\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
        "buffer_load_dwordx4 %[v0], %[rs], %[so] offen lds\n\t"
        "s_add_u32 %[so], %[so], %[stride]\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
        "buffer_load_dwordx4 %[v1], %[rs], %[so] offen lds\n\t"
        "s_mov_b32 m0, %[keep]"
        : [keep] "=&s"(keep), [so] "=&s"(so)
        : [lds] "s"(lds), [soff] "s"(soff), [stride] "s"(stride), [v0] "v"(voff0), [v1] "v"(voff1), [rs] "s"(rs)
        : "memory");
}
// one 1 KiB piece
__device__ __forceinline__ void dma1(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, unsigned lds) {
    unsigned keep;
    asm volatile("s_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[lds]\n\ts_nop 4\n\t"
                 "buffer_load_dwordx4 %[v0], %[rs], %[so] offen lds\n\ts_mov_b32 m0, %[keep]"
                 : [keep] "=&s"(keep) : [lds] "s"(lds), [so] "s"(soff), [v0] "v"(voff), [rs] "s"(rs) : "memory");
}

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

struct Args {
    long N;
    int Lc, work, fold;
    const double *Fs, *Qs, *ys;
    double *P, *m, *sink;
};

template <int K, int WAVES, bool WRITE, int LCMAX>
__global__ __launch_bounds__(WAVES * 64) void apply_like(Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int YB = LCMAX * 8 * kWave;
    constexpr int PERW = K * SLOT + YB;
    char* base = smem + wave * PERW;
    const unsigned lbase = (unsigned)(size_t)base;
    const int Lc = a.Lc, S = Lc / G;
    const long wbase = ((long)blockIdx.x * WAVES * 64 + (long)wave * 64) * Lc;
    const unsigned pitch = (unsigned)Lc * 32u;
    const unsigned span = 64u * pitch;
    const __amdgpu_buffer_rsrc_t rF = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(a.Fs + wbase * 4), 0, (int)span, 0x00020000);
    const __amdgpu_buffer_rsrc_t rQ = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(a.Qs + wbase * 4), 0, (int)span, 0x00020000);
    const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(a.ys + wbase), 0, (int)(64 * Lc * 8), 0x00020000);
    const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc(a.P + wbase * 4, 0, (int)span, 0x00020000);
    const __amdgpu_buffer_rsrc_t rM = __builtin_amdgcn_make_buffer_rsrc(a.m + wbase * 2, 0, (int)(span / 2), 0x00020000);
    // DMA piece p of instruction v lands at LDS v*1024 + p*16 = owner (8v + p/8), physical piece p%8; it must hold the
    // owner's LOGICAL piece (p%8) ^ f(owner), f(o) = (o & 7) ^ ((o >> 3) & 1): conflict-free ds_read_b128 / ds_write_b128
    const unsigned po = lane >> 3, pp = lane & 7;
    const unsigned voff0 = po * pitch + ((pp ^ po) << 4), voff1 = po * pitch + ((pp ^ po ^ 1u) << 4);
    const unsigned fo = (lane & 7) ^ ((lane >> 3) & 1);           // f(owner = this lane)
    // y: 64*Lc doubles = Lc*32 granules of 16 B, Lc/2 per owner; swizzled by h(o)
    const int NG = Lc / 2;                                        // granules per owner (16 or 8)
    const unsigned hmask = NG - 1;

    auto issue = [&](int sb) {
        const unsigned slot = lbase + (unsigned)(sb % K) * SLOT;
        dma8(rF, voff0, voff1, (unsigned)sb * SEG, 8u * pitch, slot);
        dma8(rQ, voff0, voff1, (unsigned)sb * SEG, 8u * pitch, slot + ARR);
    };
    constexpr int nD = 16, nS = WRITE ? 12 : 0;
    const int nY = Lc / 2;
    // prologue: first sub-tile, then y, then the rest of the ring
    issue(0);
    for (int t = 0; t < nY; ++t) {
        const unsigned q = (unsigned)t * 64 + lane;
        const unsigned o = q / NG, pj = q % NG;
        const unsigned ho = (NG == 16) ? (o & 15u) : ((o & 7u) ^ ((o >> 3) & 1u));
        const unsigned jj = pj ^ (ho & hmask);
        dma1(rY, (o * NG + jj) * 16u, 0u, lbase + K * SLOT + (unsigned)t * 1024u);
    }
    for (int sb = 1; sb < K && sb < S; ++sb) issue(sb);
    // stand-in for the spine fold: a dependent chain
    double z = (double)lane;
    for (int i = 0; i < a.fold; ++i) z = __builtin_fma(z, 1.0000001, 1e-9);

    double acc = z * 1e-30;
    const unsigned hself = (NG == 16) ? (lane & 15u) : fo;
    for (int sb = 0; sb < S; ++sb) {
        // ops younger than DMA(sb): the y pieces (if sb == 0), the later slots' DMAs, the stores of the last K-1 iterations
        int later = S - 1 - sb; if (later > K - 1) later = K - 1;
        int prev = sb; if (prev > K - 1) prev = K - 1;
        const int cnt = nD * later + nS * prev;      // (the y pieces are older than DMA(1): waited for with DMA(0))
        wait_vm(cnt > 63 ? 63 : cnt);
        wave_lds_sync();
        char* sF = base + (sb % K) * SLOT + lane * SEG;
        char* sQ = sF + ARR;
        const char* sY = base + K * SLOT + lane * NG * 16;
        double yv[4];
        {
            const V4 y0 = *reinterpret_cast<const V4*>(sY + (((2 * sb) ^ hself) & hmask) * 16);
            const V4 y1 = *reinterpret_cast<const V4*>(sY + (((2 * sb + 1) ^ hself) & hmask) * 16);
            __builtin_memcpy(&yv[0], &y0, 16);
            __builtin_memcpy(&yv[2], &y1, 16);
        }
#pragma unroll
        for (int i = 0; i < G; ++i) {
            double F[4], Q[4];
            const V4 f0 = *reinterpret_cast<const V4*>(sF + (((2 * i) ^ fo) << 4));
            const V4 f1 = *reinterpret_cast<const V4*>(sF + (((2 * i + 1) ^ fo) << 4));
            const V4 q0 = *reinterpret_cast<const V4*>(sQ + (((2 * i) ^ fo) << 4));
            const V4 q1 = *reinterpret_cast<const V4*>(sQ + (((2 * i + 1) ^ fo) << 4));
            __builtin_memcpy(&F[0], &f0, 16); __builtin_memcpy(&F[2], &f1, 16);
            __builtin_memcpy(&Q[0], &q0, 16); __builtin_memcpy(&Q[2], &q1, 16);
            double w = yv[i];
            for (int j = 0; j < a.work; ++j) w = __builtin_fma(w, 0.999999, F[j & 3] * 1e-12);
            acc += w;
            if (WRITE) {
                double Pn[4] = {F[0] + Q[0], F[1] + Q[1], F[2] + Q[2], F[3] + Q[3]};
                double mn[2] = {yv[i] + F[0], yv[i] - Q[3]};
                V4 p0, p1, m0;
                __builtin_memcpy(&p0, &Pn[0], 16); __builtin_memcpy(&p1, &Pn[2], 16); __builtin_memcpy(&m0, &mn[0], 16);
                *reinterpret_cast<V4*>(sF + (((2 * i) ^ fo) << 4)) = p0;          // F_i is dead: P_i takes its place
                *reinterpret_cast<V4*>(sF + (((2 * i + 1) ^ fo) << 4)) = p1;
                *reinterpret_cast<V4*>(sQ + ((i ^ fo) << 4)) = m0;                // logical piece i of the Q segment (Q_0..Q_i are dead)
            }
        }
        if (WRITE) {
            wave_lds_sync();
            const char* slot = base + (sb % K) * SLOT;
#pragma unroll
            for (int v = 0; v < 8; ++v) {
                const V4 x = *reinterpret_cast<const V4*>(slot + v * 1024 + lane * 16);
                __builtin_amdgcn_raw_buffer_store_b128(x, rP, (v & 1) ? voff1 : voff0, (unsigned)sb * SEG + (unsigned)v * 8u * pitch, 0);
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const unsigned o = 16u * v + (lane >> 2), i = lane & 3;
                const unsigned f = (o & 7) ^ ((o >> 3) & 1);
                const V4 x = *reinterpret_cast<const V4*>(slot + ARR + o * SEG + ((i ^ f) << 4));
                __builtin_amdgcn_raw_buffer_store_b128(x, rM, o * (pitch / 2) + i * 16u, (unsigned)sb * 64u, 0);
            }
        }
        if (sb + K < S) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the slot's LDS reads are done before the DMA overwrites it
            issue(sb + K);
        }
    }
    if (a.sink) a.sink[(long)blockIdx.x * WAVES * 64 + threadIdx.x] = acc;
}

template <int K, int WAVES, bool WRITE, int LCMAX>
static double run(const Args& a, int reps, bool check, const std::vector<double>& hF, const std::vector<double>& hQ,
                  const std::vector<double>& hy) {
    const long lanes = a.N / a.Lc;
    const int nb = (int)(lanes / (WAVES * 64));
    const size_t shmem = (size_t)WAVES * (K * SLOT + LCMAX * 8 * kWave);
    hipFuncSetAttribute((const void*)apply_like<K, WAVES, WRITE, LCMAX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) apply_like<K, WAVES, WRITE, LCMAX><<<nb, WAVES * 64, shmem>>>(a);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) apply_like<K, WAVES, WRITE, LCMAX><<<nb, WAVES * 64, shmem>>>(a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(err)); exit(1); }
    if (check && WRITE) {
        std::vector<double> P(a.N * 4), m(a.N * 2);
        hipMemcpy(P.data(), a.P, P.size() * 8, hipMemcpyDeviceToHost);
        hipMemcpy(m.data(), a.m, m.size() * 8, hipMemcpyDeviceToHost);
        long bad = 0;
        for (long k = 0; k < a.N; ++k) {
            for (int j = 0; j < 4; ++j) if (P[k * 4 + j] != hF[k * 4 + j] + hQ[k * 4 + j]) ++bad;
            if (m[k * 2] != hy[k] + hF[k * 4] || m[k * 2 + 1] != hy[k] - hQ[k * 4 + 3]) ++bad;
        }
        printf("    check K=%d waves=%d Lc=%d: %ld mismatches\n", K, WAVES, a.Lc, bad);
        if (bad) exit(2);
    }
    return ms / reps * 1e3;     // us
}

int main(int argc, char** argv) {
    const long N = 1L << 20;
    std::vector<double> hF(N * 4), hQ(N * 4), hy(N);
    for (long i = 0; i < N * 4; ++i) { hF[i] = (double)(i % 1000003) * 0.5; hQ[i] = (double)(i % 999983) * 0.25 + 1.0; }
    for (long i = 0; i < N; ++i) hy[i] = (double)(i % 7919) - 3000.0;
    Args a{};
    a.N = N;
    hipMalloc((void**)&a.Fs, N * 32); hipMalloc((void**)&a.Qs, N * 32); hipMalloc((void**)&a.ys, N * 8);
    hipMalloc((void**)&a.P, N * 32); hipMalloc((void**)&a.m, N * 16); hipMalloc((void**)&a.sink, (N / 16) * 8);
    hipMemcpy((void*)a.Fs, hF.data(), N * 32, hipMemcpyHostToDevice);
    hipMemcpy((void*)a.Qs, hQ.data(), N * 32, hipMemcpyHostToDevice);
    hipMemcpy((void*)a.ys, hy.data(), N * 8, hipMemcpyHostToDevice);
    const int reps = 50;
    const double mb_r = N * 72.0 / 1e6, mb_rw = N * 120.0 / 1e6;
    printf("N = 2^20 steps, d = 2 fp64: read %.1f MB (F, Q, y), write %.1f MB (P, m) per pass\n", mb_r, mb_rw - mb_r);
    for (int lcsel = 0; lcsel < 2; ++lcsel) {
        a.Lc = lcsel ? 16 : 32;
        printf("== %d steps per lane ==\n", a.Lc);
        for (int work : {0, 100, 225, 350}) {
            for (int fold : {0, 1000}) {
                a.work = work; a.fold = fold;
                double t[8];
                if (a.Lc == 32) {
                    t[0] = run<1, 2, true, 32>(a, reps, work == 0 && fold == 0, hF, hQ, hy);
                    t[1] = run<2, 2, true, 32>(a, reps, work == 0 && fold == 0, hF, hQ, hy);
                    t[2] = run<3, 2, true, 32>(a, reps, work == 0 && fold == 0, hF, hQ, hy);
                    t[3] = run<3, 2, false, 32>(a, reps, false, hF, hQ, hy);
                    printf("  128 lanes x 32: work %3d fold %4d: rw K=1 %6.1f us (%4.2f TB/s)  K=2 %6.1f (%4.2f)  K=3 %6.1f (%4.2f) | read-only K=3 %6.1f (%4.2f)\n",
                           work, fold, t[0], mb_rw / t[0], t[1], mb_rw / t[1], t[2], mb_rw / t[2], t[3], mb_r / t[3]);
                } else {
                    t[0] = run<1, 4, true, 16>(a, reps, work == 0 && fold == 0, hF, hQ, hy);
                    t[1] = run<2, 4, true, 16>(a, reps, work == 0 && fold == 0, hF, hQ, hy);
                    t[2] = run<1, 2, true, 16>(a, reps, work == 0 && fold == 0, hF, hQ, hy);
                    t[3] = run<3, 2, true, 16>(a, reps, work == 0 && fold == 0, hF, hQ, hy);
                    printf("  Lc 16: work %3d fold %4d: 256 lanes K=1 %6.1f us (%4.2f TB/s)  K=2 %6.1f (%4.2f) | 128 lanes (512 wgs) K=1 %6.1f (%4.2f)  K=3 %6.1f (%4.2f)\n",
                           work, fold, t[0], mb_rw / t[0], t[1], mb_rw / t[1], t[2], mb_rw / t[2], t[3], mb_rw / t[3]);
                }
            }
        }
    }
    return 0;
}
