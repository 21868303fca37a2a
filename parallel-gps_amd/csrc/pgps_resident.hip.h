// pgps_resident.hip.h -- filter + log-likelihood + smoother of a whole series in ONE resident launch (d <= 2).
//
// The three-launch path (pgps_kernels.hip.h) reads Fs, Qs three times -- reduce, Kalman pass, RTS pass -- and writes
// the filtered moments only to read them back: 389 real bytes per step at d = 2 fp64 where the reference's contract
// (pssgp/kalman/parallel.py:121-152, 187-201) touches each array once per function.  At 2^20 steps its streaming loops
// already run at the rate a CU's memory path sustains (10 - 13 B/clk per CU, profiles/r03_stamps_before.txt), so the
// pass gets shorter only by moving fewer bytes.  Here every workgroup keeps its 256 x LC steps ON CHIP between the
// phases of the scan and the only traffic is the contract's minimum: Fs, Qs, ys in once (72 B/step), fms, fPs, sms,
// sPs out once (96 B/step).
//
//   phase 1   stream the lane's LC steps in (coalesced, transposed through the wave's LDS slots): F stays in
//             registers, Q stays in the lane's LDS slot, y in registers; lane-serial filt_extend -> lane aggregate;
//             workgroup scan; the workgroup's total is published (write-through stores)
//   barrier 1 grid-wide (every workgroup resident: grid <= CUs, one workgroup per CU by its LDS footprint)
//   phase 2   fold the totals to the left, Kalman pass over the kept F, Q, y: fms, fPs out (through the slots Q has
//             left, drained coalesced), log-likelihood, and the smoothing element (E, g, L) of every step
//             (parallel.py:159-166) IN PLACE of the step's inputs: E, g in the registers F and y have left, L in
//             the slot; lane aggregate of the elements, workgroup suffix scan, total published
//   barrier 2
//   phase 3   fold the totals to the right, lane-serial smoothing-operator pass (parallel.py:176-184) backwards over
//             the kept elements: sms, sPs out through the slots
//
// On chip per lane at d = 2 fp64, LC = 16: 64 doubles of F / E and 16 of y (later 32 of g) in registers, 512 B of
// Q / filtered P / L / smoothed P in LDS: 528 B slots x 256 lanes = 132 KiB of the CU's 160, one workgroup per CU,
// one wave per SIMD on 512 registers.
//
// Inter-workgroup visibility: the hand-off recipe of k_filter_single (pgps_kernels.hip.h; MI355X_MICROARCH.md, hand-offs
// with sc1 loads in place of the acquire): one lane stores the record with agent-scope atomic (write-through) stores,
// drains them, adds one arrival to the counter shard of its tile; eight lanes poll one shard each with relaxed
// agent-scope loads; the records are read with agent-scope atomic loads only.  Every spin is bounded (status bit 1).
// The two barriers of a launch count on ONE set of eight shards (targets n and 2 n); two sets alternate between
// launches and every launch zeroes the set of the next one, so nothing is memset in front of a launch and a launch that
// gave up leaves no debt.
//
// Ragged series: steps at or beyond N are padded ON LOAD with the scan's identity step (F = I, Q = 0, y = NaN: a pure
// predict that changes nothing), their smoothing elements are forced to the identity, the element of step N - 1 is the
// reference's last element (0, m, P) (parallel.py:155-156), and stores beyond N are predicated off -- one code path.
#pragma once

#include "pgps_fused.hip.h"
#include "pgps_kernels.hip.h"

namespace pgps {


template <typename T, int D, int LC>
struct ResCfg {
    static constexpr int W = (int)sizeof(T), MAT = D * D, G = 4;
    static_assert(LC % G == 0, "whole sub-tiles");
    static constexpr int S = LC / G;
    using GF = StageGeom<MAT * W, G>;               // F, Q, P records of one sub-tile: global side
    using GM = StageGeom<D * W, G>;                 // m records
    static constexpr int SLOT = LC * MAT * W + 16;  // a lane's LDS slot (LC matrix records) + 16 B: conflict-free owner reads
    static constexpr int SLOTS = kWave * SLOT;      // per wave
    static constexpr int MST = GM::BYTES;           // per wave: staging of the means of one sub-tile
    static constexpr int NSCAN = kWaves * Dim<D>::NFILT * W;
    static constexpr int BYTES = kWaves * (SLOTS + MST) + NSCAN + kWaves * 8;
};

#define PGPS_RSTAMP(IDX)                                                                              \
    do {                                                                                              \
        if (ra.stamps && threadIdx.x == 0) ra.stamps[(long)blockIdx.x * 16 + (IDX)] = __builtin_readcyclecounter(); \
    } while (0)

// one 16-byte piece of an identity-step record: kind 0 = F (identity matrix), 1 = Q (zero)
template <typename T, int D>
__device__ __forceinline__ V4 res_pad_piece(int kind, int pos_in_rec) {
    constexpr int PER = 16 / (int)sizeof(T);
    T tmp[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int e = (pos_in_rec * PER + j) % (D * D);
        tmp[j] = (kind == 0 && (e / D) == (e % D)) ? T(1) : T(0);
    }
    V4 v;
    __builtin_memcpy(&v, tmp, 16);
    return v;
}

// global -> registers: the pieces of sub-tile `sb` of one matrix array of this wave (piece q = v * 64 + lane belongs to
// owner q / NV).  `lim` = bytes of the array from the wave's first record on; pieces at or beyond it are padded.
template <typename T, int D, typename GEO>
__device__ __forceinline__ void res_issue(const char* __restrict__ g, int sb, long lane_pitch, long lim, bool full, int kind,
                                          V4* r) {
    const int lane = threadIdx.x & (kWave - 1);
    constexpr int PPR = D * D * (int)sizeof(T) / 16 > 0 ? D * D * (int)sizeof(T) / 16 : 1;     // pieces per record
    if (full) {
#pragma unroll
        for (int v = 0; v < GEO::NV; ++v) {
            const int q = v * kWave + lane;
            r[v] = *reinterpret_cast<const V4*>(g + (long)(q / GEO::NV) * lane_pitch + (long)sb * GEO::SEG + (q % GEO::NV) * 16);
        }
    } else {
#pragma unroll
        for (int v = 0; v < GEO::NV; ++v) {
            const int q = v * kWave + lane;
            const long off = (long)(q / GEO::NV) * lane_pitch + (long)sb * GEO::SEG + (q % GEO::NV) * 16;
            if (off < lim) r[v] = *reinterpret_cast<const V4*>(g + off);
            else r[v] = res_pad_piece<T, D>(kind, (q % GEO::NV) % PPR);
        }
    }
}
// registers -> the owners' slots
template <typename GEO, int SLOT>
__device__ __forceinline__ void res_commit(char* slots, int sb, const V4* r) {
    const int lane = threadIdx.x & (kWave - 1);
#pragma unroll
    for (int v = 0; v < GEO::NV; ++v) {
        const int q = v * kWave + lane;
        *reinterpret_cast<V4*>(slots + (q / GEO::NV) * SLOT + sb * GEO::SEG + (q % GEO::NV) * 16) = r[v];
    }
}
// the owners' slots -> global (coalesced), pieces at or beyond `lim` dropped
template <typename GEO, int SLOT>
__device__ __forceinline__ void res_drain(char* __restrict__ g, int sb, long lane_pitch, long lim, bool full, const char* slots) {
    const int lane = threadIdx.x & (kWave - 1);
#pragma unroll
    for (int v = 0; v < GEO::NV; ++v) {
        const int q = v * kWave + lane;
        const V4 x = *reinterpret_cast<const V4*>(slots + (q / GEO::NV) * SLOT + sb * GEO::SEG + (q % GEO::NV) * 16);
        const long off = (long)(q / GEO::NV) * lane_pitch + (long)sb * GEO::SEG + (q % GEO::NV) * 16;
        if (full || off < lim) *reinterpret_cast<V4*>(g + off) = x;
    }
}
// the means of one sub-tile: staging buffer (one sub-tile deep) -> global
template <typename GEO>
__device__ __forceinline__ void res_drain_m(char* __restrict__ g, int sb, long lane_pitch, long lim, bool full, const char* mst) {
    const int lane = threadIdx.x & (kWave - 1);
#pragma unroll
    for (int v = 0; v < GEO::NV; ++v) {
        const int q = v * kWave + lane;
        const V4 x = *reinterpret_cast<const V4*>(mst + (q / GEO::NV) * GEO::STRIDE + (q % GEO::NV) * 16);
        const long off = (long)(q / GEO::NV) * lane_pitch + (long)sb * GEO::SEG + (q % GEO::NV) * 16;
        if (full || off < lim) *reinterpret_cast<V4*>(g + off) = x;
    }
}

// one arrival of this workgroup at a grid-wide barrier and the wait for `rounds` x (every workgroup's arrival).
// The caller's lane 0 has drained the stores it publishes (s_waitcnt vmcnt(0)) before this is called.
__device__ __forceinline__ void res_grid_barrier(int* bar, int tile, int nblocks, int rounds, int* status) {
    if (threadIdx.x == 0) __hip_atomic_fetch_add(bar + (tile & 7) * 32, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x < 8) {
        const int want = rounds * ((nblocks - (int)threadIdx.x + 7) / 8);       // tiles whose index is threadIdx.x mod 8
        if (want > 0) wait_flag(bar + threadIdx.x * 32, want, status);
    }
    __syncthreads();
}

// smoothing element of step k_next - 1 from the predict of step k_next (parallel.py:159-166), or the series' last
// element (k_next == N, parallel.py:155-156), or the identity (padding beyond the series)
template <typename T, int D>
__device__ __forceinline__ void res_element(long k_next, long N, bool tail, const MeanCov<T, D>& prev, const T* mp, const T* Pp,
                                            const T* FP, SmthElem<T, D>& e) {
    smth_element(prev, mp, Pp, FP, e);
    if (tail) {
        constexpr int MAT = D * D, SYM = Dim<D>::SYM;
        const bool last = (k_next == N), pad = (k_next > N);
        if (last || pad) {
#pragma unroll
            for (int i = 0; i < MAT; ++i) e.E[i] = (pad && (i / D == i % D)) ? T(1) : T(0);
#pragma unroll
            for (int i = 0; i < D; ++i) e.g[i] = last ? prev.m[i] : T(0);
#pragma unroll
            for (int i = 0; i < SYM; ++i) e.L[i] = last ? prev.P[i] : T(0);
        }
    }
}

template <typename T, int D, int LC, bool FUSED>
__global__ __launch_bounds__(kBlock) void k_pkfs_resident(const ResArgs<T> ra) {
    using CFG = ResCfg<T, D, LC>;
    using GF = typename CFG::GF;
    using GM = typename CFG::GM;
    using FE = FiltElem<T, D>;
    using SE = SmthElem<T, D>;
    using MC = MeanCov<T, D>;
    constexpr int MAT = D * D, SYM = Dim<D>::SYM, NF = Dim<D>::NFILT, NS = Dim<D>::NSMTH, G = CFG::G, S = CFG::S;
    constexpr int SLOT = CFG::SLOT;
    static_assert(kBlock == 256, "one workgroup of four waves per CU");
    const ScanArgs<T>& a = ra.s;

    __shared__ __attribute__((aligned(16))) char smem[CFG::BYTES];
    const int tile = blockIdx.x;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    char* slots = smem + wave * CFG::SLOTS;                                 // this wave's 64 slots
    char* mst = smem + kWaves * CFG::SLOTS + wave * CFG::MST;               // this wave's staging of the means
    T* lds = reinterpret_cast<T*>(smem + kWaves * (CFG::SLOTS + CFG::MST));   // the workgroup scans' scratch
    double* lds_ll = reinterpret_cast<double*>(smem + kWaves * (CFG::SLOTS + CFG::MST) + CFG::NSCAN);
    char* myslot = slots + lane * SLOT;

    PGPS_RSTAMP(0);
    if (tile == 0 && threadIdx.x < 8) ra.bar_next[threadIdx.x * 32] = 0;

    const long N = a.N;
    const long gt = (long)tile * kBlock + threadIdx.x;
    const long k0 = gt * LC;
    const long wbase = ((long)tile * kBlock + wave * kWave) * LC;
    const bool full = (wbase + (long)kWave * LC <= N);                      // wave-uniform: no padding, no predicates
    const bool tail = (wbase + (long)kWave * LC + 1 > N);                   // wave-uniform: holds step N - 1 or padding
    const long pitchF = (long)LC * MAT * sizeof(T), pitchM = (long)LC * D * sizeof(T);
    const long limF = (N - wbase) * MAT * (long)sizeof(T), limM = (N - wbase) * D * (long)sizeof(T);

    T h[D], P0[SYM];
    T Rn;
    if constexpr (FUSED) {
        gp_prior<T, D>(ra.m, h, P0);
    } else {
        T P0f[MAT];
#pragma unroll
        for (int i = 0; i < MAT; ++i) P0f[i] = a.P0[i];
        sym_from_full<T, D>(P0f, P0);
#pragma unroll
        for (int i = 0; i < D; ++i) h[i] = a.H[i];
    }
    Rn = a.R;

    // ---------------------------------------------------------------------------------------------
    // phase 1: the chunk comes on chip and is reduced
    // ---------------------------------------------------------------------------------------------
    T Freg[LC][MAT];            // F_k; from phase 2 on: E_k
    T yreg[LC];                 // y_k (FUSED: first the time stamps); from phase 2 on g_k[0]
    T greg[LC];                 // g_k[1..] (D = 2: one value)  -- D - 1 values per step, D <= 2
    static_assert(D <= 2, "resident pass: d <= 2");
    FE agg;
    filt_identity(agg);
    {
        const T nanv = T(__builtin_nan(""));
        if constexpr (FUSED) {
            // time stamps and observations straight from global memory (LC contiguous values per lane)
            T tv[LC];
            const T* tp = ra.m.ts;
            T tprev;
            if (full) {
                load_rec<T, LC>(tp + k0, tv);
                load_rec<T, LC>(a.ys + k0, yreg);
                tprev = (k0 > 0) ? tp[k0 - 1] : ra.m.t_prev;
            } else {
                const T tl = tp[N - 1];
#pragma unroll
                for (int j = 0; j < LC; ++j) {
                    const bool in = k0 + j < N;
                    tv[j] = in ? tp[in ? k0 + j : 0] : tl;
                    yreg[j] = in ? a.ys[in ? k0 + j : 0] : nanv;
                }
                tprev = (k0 > 0) ? (k0 - 1 < N ? tp[k0 - 1 < N ? k0 - 1 : 0] : tl) : ra.m.t_prev;
            }
#pragma unroll
            for (int j = 0; j < LC; ++j) {
                T Qf[MAT];
                lti_step<T, D>(ra.m, tv[j] - tprev, Freg[j], Qf);
                tprev = tv[j];
                store_rec<T, MAT>(reinterpret_cast<T*>(myslot) + j * MAT, Qf);
                if (k0 + j == 0) {
                    filt_first(agg, P0, yreg[j], h, Rn);
                } else {
                    T Q[SYM];
                    sym_from_full<T, D>(Qf, Q);
                    filt_extend(agg, Freg[j], Q, yreg[j], h, Rn);
                }
            }
        } else {
            const char* gF = reinterpret_cast<const char*>(a.Fs + wbase * MAT);
            const char* gQ = reinterpret_cast<const char*>(a.Qs + wbase * MAT);
            V4 rF[GF::NV], rQ[GF::NV];
            res_issue<T, D, GF>(gF, 0, pitchF, limF, full, 0, rF);
            res_issue<T, D, GF>(gQ, 0, pitchF, limF, full, 1, rQ);
            if (full) {
                load_rec<T, LC>(a.ys + k0, yreg);
            } else {
#pragma unroll
                for (int j = 0; j < LC; ++j) yreg[j] = (k0 + j < N) ? a.ys[k0 + j < N ? k0 + j : 0] : nanv;
            }
#pragma unroll
            for (int sb = 0; sb < S; ++sb) {
                // F of the sub-tile through the slots into registers, then Q into the same place, where it stays
                wave_lds_sync();
                res_commit<GF, SLOT>(slots, sb, rF);
                if (sb + 1 < S) res_issue<T, D, GF>(gF, sb + 1, pitchF, limF, full, 0, rF);
                wave_lds_sync();
#pragma unroll
                for (int i = 0; i < G; ++i) load_rec<T, MAT>(reinterpret_cast<const T*>(myslot) + (sb * G + i) * MAT, Freg[sb * G + i]);
                wave_lds_sync();
                res_commit<GF, SLOT>(slots, sb, rQ);
                if (sb + 1 < S) res_issue<T, D, GF>(gQ, sb + 1, pitchF, limF, full, 1, rQ);
                wave_lds_sync();
#pragma unroll
                for (int i = 0; i < G; ++i) {
                    const int j = sb * G + i;
                    T Qf[MAT];
                    load_rec<T, MAT>(reinterpret_cast<const T*>(myslot) + j * MAT, Qf);
                    if (k0 + j == 0) {
                        filt_first(agg, P0, yreg[j], h, Rn);
                    } else {
                        T Q[SYM];
                        sym_from_full<T, D>(Qf, Q);
                        filt_extend(agg, Freg[j], Q, yreg[j], h, Rn);
                    }
                }
            }
        }
    }
    PGPS_RSTAMP(1);
    FE excl;
    {
        FE total;
        block_scan_exclusive<FE, true>(agg, excl, total, lds);
        PGPS_RSTAMP(2);
        if (threadIdx.x == 0) {
            T v[NF];
            pack(total, v);
#pragma unroll
            for (int i = 0; i < NF; ++i) pub_store(a.spine + (long)tile * NF + i, v[i]);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    res_grid_barrier(ra.bar, tile, a.nblocks, 1, a.status);
    PGPS_RSTAMP(3);

    // ---------------------------------------------------------------------------------------------
    // phase 2: carry in, Kalman pass, smoothing elements in place
    // ---------------------------------------------------------------------------------------------
    MC s;
#pragma unroll
    for (int i = 0; i < D; ++i) s.m[i] = T(0);
#pragma unroll
    for (int i = 0; i < SYM; ++i) s.P[i] = P0[i];
    if (tile > 0) {
        FE mine, left;
        filt_identity(mine);
        if ((int)threadIdx.x < tile) {
            T v[NF];
#pragma unroll
            for (int i = 0; i < NF; ++i) v[i] = pub_load(a.spine + (long)threadIdx.x * NF + i);
            unpack(v, mine);
        }
        block_reduce_ordered(mine, left, lds);
        filt_apply(s, left);
    }
    filt_apply(s, excl);
    PGPS_RSTAMP(4);

    LogLik ll;
    SE sagg;
    smth_identity(sagg);
    // F, Q of the step after the chunk: the next lane's first step (its registers / its slot); the wave's last lane
    // reads global memory (the next wave's or workgroup's first step), or pads
    T Fh[MAT], Qh[MAT];
    {
        T Q0f[MAT];
        load_rec<T, MAT>(reinterpret_cast<const T*>(myslot), Q0f);
#pragma unroll
        for (int i = 0; i < MAT; ++i) {
            Fh[i] = wshfl_down(Freg[0][i], 1);
            Qh[i] = wshfl_down(Q0f[i], 1);
        }
        if (lane == kWave - 1) {
            const long k1 = k0 + LC;
            if (k1 < N) {
                if constexpr (FUSED) {
                    lti_step<T, D>(ra.m, ra.m.ts[k1] - ra.m.ts[k1 - 1], Fh, Qh);
                } else {
                    load_rec<T, MAT>(a.Fs + k1 * MAT, Fh);
                    load_rec<T, MAT>(a.Qs + k1 * MAT, Qh);
                }
            } else {
#pragma unroll
                for (int i = 0; i < MAT; ++i) { Fh[i] = (i / D == i % D) ? T(1) : T(0); Qh[i] = T(0); }
            }
        }
    }
    const bool store_f = (a.fms != nullptr);
    char* gP = reinterpret_cast<char*>(a.fPs + wbase * MAT);
    char* gM = reinterpret_cast<char*>(a.fms + wbase * D);
    T Lhold[SYM];
#pragma unroll
    for (int sb = 0; sb < S; ++sb) {
        T Lnew[G][SYM];
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int j = sb * G + i;
            const long k = k0 + j;
            T Qf[MAT], Q[SYM];
            load_rec<T, MAT>(reinterpret_cast<const T*>(myslot) + j * MAT, Qf);
            sym_from_full<T, D>(Qf, Q);
            MC prev = s;
            T mp[D], Pp[SYM], FP[MAT];
            kf_step(s, Freg[j], Q, yreg[j], h, Rn, k == 0, ll, mp, Pp, FP);
            if (j > 0) {
                // element of step j - 1: E, g take the registers F_{j-1}, y_{j-1} have left; L waits for its slot
                SE e, r;
                res_element<T, D>(k, N, tail, prev, mp, Pp, FP, e);
                smth_combine(sagg, e, r);
                sagg = r;
#pragma unroll
                for (int q = 0; q < MAT; ++q) Freg[j - 1][q] = e.E[q];
                yreg[j - 1] = e.g[0];
                if constexpr (D == 2) greg[j - 1] = e.g[1];
#pragma unroll
                for (int q = 0; q < SYM; ++q) {
                    if (i == 0) Lhold[q] = e.L[q]; else Lnew[i - 1][q] = e.L[q];
                }
            }
            // filtered moments of step j: P into the slot Q_j has left, m into the staging buffer
            T Pf[MAT];
            full_from_sym<T, D>(s.P, Pf);
            store_rec<T, MAT>(reinterpret_cast<T*>(myslot) + j * MAT, Pf);
            store_rec<T, D>(reinterpret_cast<T*>(mst + lane * GM::STRIDE) + i * D, s.m);
        }
        wave_lds_sync();
        if (store_f) {
            res_drain<GF, SLOT>(gP, sb, pitchF, limF, full, slots);
            res_drain_m<GM>(gM, sb, pitchM, limM, full, mst);
        }
        wave_lds_sync();
        // L of steps 4 sb - 1 .. 4 sb + 2 into their slots (drained above; LDS keeps a wave's accesses in order)
        if (sb > 0) store_rec<T, SYM>(reinterpret_cast<T*>(myslot) + (sb * G - 1) * MAT, Lhold);
#pragma unroll
        for (int i = 0; i + 1 < G; ++i) store_rec<T, SYM>(reinterpret_cast<T*>(myslot) + (sb * G + i) * MAT, Lnew[i]);
    }
    {
        // element of the chunk's last step from the step after the chunk
        constexpr int j = LC - 1;
        T Q[SYM], mp[D], Pp[SYM], FP[MAT];
        sym_from_full<T, D>(Qh, Q);
        mat_vec<T, D>(Fh, s.m, mp);
        predict_cov<T, D>(Fh, s.P, Q, FP, Pp);
        SE e, r;
        res_element<T, D>(k0 + LC, N, tail, s, mp, Pp, FP, e);
        smth_combine(sagg, e, r);
        sagg = r;
#pragma unroll
        for (int q = 0; q < MAT; ++q) Freg[j][q] = e.E[q];
        yreg[j] = e.g[0];
        if constexpr (D == 2) greg[j] = e.g[1];
        store_rec<T, SYM>(reinterpret_cast<T*>(myslot) + j * MAT, e.L);
    }
    PGPS_RSTAMP(5);
    SE sexcl;
    {
        const double v = ll.value();
        const double t = block_sum_double(v, lds_ll);
        SE stotal;
        block_scan_exclusive<SE, false>(sagg, sexcl, stotal, lds);
        PGPS_RSTAMP(6);
        if (threadIdx.x == 0) {
            T vv[NS];
            pack(stotal, vv);
#pragma unroll
            for (int i = 0; i < NS; ++i) pub_store(a.sspine + (long)tile * NS + i, vv[i]);
            pub_store(a.llpart + tile, t);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    res_grid_barrier(ra.bar, tile, a.nblocks, 2, a.status);
    PGPS_RSTAMP(7);

    // ---------------------------------------------------------------------------------------------
    // phase 3: carry back, smoothing pass over the kept elements
    // ---------------------------------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < D; ++i) s.m[i] = T(0);
#pragma unroll
    for (int i = 0; i < SYM; ++i) s.P[i] = T(0);
    if (tile + 1 < a.nblocks) {
        SE mine, right;
        smth_identity(mine);
        const int b = tile + 1 + (int)threadIdx.x;
        if (b < a.nblocks) {
            T v[NS];
#pragma unroll
            for (int i = 0; i < NS; ++i) v[i] = pub_load(a.sspine + (long)b * NS + i);
            unpack(v, mine);
        }
        block_reduce_ordered(mine, right, lds);
        smth_apply(right, s);
    }
    smth_apply(sexcl, s);
    PGPS_RSTAMP(8);
    if (tile == 0 && a.ll != nullptr) {
        double v = 0.0;
        for (int b = threadIdx.x; b < a.nblocks; b += kBlock) v += pub_load(a.llpart + b);
        const double t = block_sum_double(v, lds_ll);
        if (threadIdx.x == 0) *a.ll = t;
    }
    char* oP = reinterpret_cast<char*>(a.sPs + wbase * MAT);
    char* oM = reinterpret_cast<char*>(a.sms + wbase * D);
#pragma unroll
    for (int sb = S - 1; sb >= 0; --sb) {
#pragma unroll
        for (int i = G - 1; i >= 0; --i) {
            const int j = sb * G + i;
            SE e;
#pragma unroll
            for (int q = 0; q < MAT; ++q) e.E[q] = Freg[j][q];
            e.g[0] = yreg[j];
            if constexpr (D == 2) e.g[1] = greg[j];
            load_rec<T, SYM>(reinterpret_cast<const T*>(myslot) + j * MAT, e.L);
            smth_apply(e, s);
            T Pf[MAT];
            full_from_sym<T, D>(s.P, Pf);
            store_rec<T, MAT>(reinterpret_cast<T*>(myslot) + j * MAT, Pf);
            store_rec<T, D>(reinterpret_cast<T*>(mst + lane * GM::STRIDE) + i * D, s.m);
        }
        wave_lds_sync();
        res_drain<GF, SLOT>(oP, sb, pitchF, limF, full, slots);
        res_drain_m<GM>(oM, sb, pitchM, limM, full, mst);
        wave_lds_sync();
    }
    PGPS_RSTAMP(9);
}

}  // namespace pgps
