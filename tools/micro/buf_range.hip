// buf_range.hip -- what a 16-byte buffer access returns when only part of it lies inside the descriptor's range
// (raw buffer, stride 0): the row-cooperative kernels' last piece of an odd-d fp64 record reaches 8 bytes beyond the
// record.  Prints the four dwords of a buffer_load_dwordx4 at offsets 0, 8, 16, 24 of a 24-byte buffer, the same through
// LDS-DMA, and what a 16-byte store at offset 16 writes.
//   hipcc -O2 --offload-arch=gfx950 buf_range.hip -o buf_range && ./buf_range
#include <hip/hip_runtime.h>
#include <cstdio>

typedef unsigned int V4 __attribute__((ext_vector_type(4)));

__global__ void probe(const unsigned* src, unsigned* out, unsigned* dst) {
    __shared__ __attribute__((aligned(16))) unsigned lds[64 * 4];
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(src), 0, 24, 0x00020000);
    const int t = threadIdx.x;
    for (int i = t; i < 256; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    const unsigned off = (unsigned)t * 8u;          // lanes 0..3: offsets 0, 8, 16, 24
    const V4 x = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(t < 4 ? off : 0x7ffff000u), 0, 0);
    if (t < 4) { out[4 * t] = x.x; out[4 * t + 1] = x.y; out[4 * t + 2] = x.z; out[4 * t + 3] = x.w; }
    // the same through LDS-DMA
    unsigned keep;
    const unsigned voff = t < 4 ? off : 0x7ffff000u, l = (unsigned)(size_t)lds;
    asm volatile("s_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[lds]\n\ts_nop 4\n\t"
                 "buffer_load_dwordx4 %[v0], %[rs], 0 offen lds\n\ts_mov_b32 m0, %[keep]\n\ts_waitcnt vmcnt(0)"
                 : [keep] "=&s"(keep) : [lds] "s"(l), [v0] "v"(voff), [rs] "s"(rs) : "memory");
    __syncthreads();
    if (t < 4) for (int j = 0; j < 4; ++j) out[16 + 4 * t + j] = lds[4 * t + j];
    // a 16-byte store at offset 16 of a 24-byte destination (dst has 64 bytes allocated, pre-set to 0x11111111)
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(dst, 0, 24, 0x00020000);
    if (t == 0) __builtin_amdgcn_raw_buffer_store_b128(V4{0xa0u, 0xa1u, 0xa2u, 0xa3u}, rd, 16, 0, 0);
}

int main() {
    unsigned h[16], *src, *out, *dst, ho[32], hd[16];
    for (int i = 0; i < 16; ++i) h[i] = 0x100u + i;
    (void)hipMalloc(&src, 64); (void)hipMalloc(&out, 128); (void)hipMalloc(&dst, 64);
    (void)hipMemcpy(src, h, 64, hipMemcpyHostToDevice);
    (void)hipMemset(out, 0, 128); (void)hipMemset(dst, 0x11, 64);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, src, out, dst);
    (void)hipMemcpy(ho, out, 128, hipMemcpyDeviceToHost); (void)hipMemcpy(hd, dst, 64, hipMemcpyDeviceToHost);
    printf("source dwords 0x100 .. 0x10f, descriptor range 24 bytes (6 dwords)\n");
    for (int t = 0; t < 4; ++t) printf("load  offset %2d: %08x %08x %08x %08x   lds-dma: %08x %08x %08x %08x\n", 8 * t, ho[4 * t], ho[4 * t + 1],
                                       ho[4 * t + 2], ho[4 * t + 3], ho[16 + 4 * t], ho[17 + 4 * t], ho[18 + 4 * t], ho[19 + 4 * t]);
    printf("store offset 16 (a0 a1 a2 a3) -> dst dwords 4..7: %08x %08x %08x %08x\n", hd[4], hd[5], hd[6], hd[7]);
    return 0;
}
