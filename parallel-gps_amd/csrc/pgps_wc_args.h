// pgps_wc_args.h -- arguments and record sizes shared by the kernels of the d <= 32 families: the wave-cooperative
// kernels (pgps_wc.hip, all levels) and the two-rows level-1 kernels (pgps_rc2.hip.h, d = 17..32).
#pragma once
#include <hip/hip_runtime.h>

namespace pgps {
namespace wc {

// global (compact) records: filter [A d^2 | C d^2 | J d^2 | b d | eta d], smoother [E d^2 | L d^2 | g d]
__host__ __device__ inline int nfilt(int d) { return 3 * d * d + 2 * d; }
__host__ __device__ inline int nsmth(int d) { return 2 * d * d + d; }

// The two-rows level-1 kernels (pgps_rc2.hip.h) pass operands between lanes in registers (DPP broadcasts,
// v_permlane16_swap): they are used -- and compiled -- only where they need NO scratch memory (tools/scratch_gate.py, DESIGN.md
// section 4k: a register spilled or reloaded under a partial EXEC mask does not carry the inactive lanes' values, and a
// later cross-lane read of such a lane returns garbage; the one wrong result this family ever produced came from a kernel
// with 444 B of scratch).  fp64 spills from d = 24 (140 B per lane in rc2_apply1 at d = 24 .. 1676 B at d = 32), fp32 at
// d = 32 (80 B): those dimensions run on the LDS-tile kernels of pgps_wc.hip, which exchange operands through LDS.
constexpr int kRc2MaxF64 = 23, kRc2MaxF32 = 31;
template <typename T>
__host__ __device__ constexpr bool rc2_covers(int d) { return d >= 17 && d <= (sizeof(T) == 8 ? kRc2MaxF64 : kRc2MaxF32); }

// ---- kernel arguments ------------------------------------------------------------------------------
template <typename T>
struct WcArgs {
    long N;
    int d, Lw;
    long nchunk;            // level-1 chunks
    int ngroup;             // level-2 groups
    int kgroup;             // chunks per group
    const T *P0, *H;
    T R;
    const T *Fs, *Qs, *ys;
    T *fms, *fPs, *sms, *sPs;
    T* Es;                  // (N, d, d) smoother gains, wc_apply1 -> wc_smooth1: sPs itself, or workspace between the
                            // phases of a segment (the smoothed arrays only arrive with the last phase)
    double* ll;
    // workspace (compact records)
    T* agg1;                // (nchunk, nfilt)
    T* lpre1;               // (nchunk, nfilt)  exclusive prefix of agg1 inside its group
    T* agg2;                // (ngroup, nfilt)
    T* carry2;              // (ngroup, d + d^2) filtered (m, P) entering each group
    T* sagg1;               // (nchunk, nsmth)
    T* lsuf1;               // (nchunk, nsmth)  exclusive suffix of sagg1 inside its group
    T* sagg2;               // (ngroup, nsmth)
    T* scarry2;             // (ngroup, d + d^2) smoothed (m, P) of the first step after each group
    double* llpart;         // (nchunk,)
    long ks_n, ks_stride;   // one Kogge-Stone level of the two-rows scan kernels: out[e] = in[e - stride] (x) in[e], e < n
    const T* ks_in;
    T* ks_out;
    T* enter1;              // (nchunk, d + d^2) filtered (m, P) entering each chunk      (two-rows level-1 kernels)
    T* senter1;             // (nchunk, d + d^2) smoothed (m, P) of the first step after each chunk
    T *ksA, *ksB;           // (ngroup, nfilt) each: Kogge-Stone ping-pong over the group totals (both scans)
    // one segment of a series sharded over several GPUs (pgps_seg_*): whole series = first and last, no pointers
    int seg_first, seg_last;
    const T* carry_in;      // (d + d^2) filtered (m, P) entering the segment            (not seg_first)
    const T* carry_back;    // (d + d^2) smoothed (m, P) of the next segment's first step (not seg_last)
    const T *halo_F, *halo_Q;   // (d, d) each: F, Q of the next segment's first step     (not seg_last)
};

}  // namespace wc
}  // namespace pgps
