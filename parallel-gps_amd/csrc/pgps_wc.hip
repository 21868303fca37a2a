// pgps_wc.hip -- the "wave-cooperative" scan family: state dimensions 7 <= d <= 32 (runtime d).
//
// For d x d operands that no longer fit one lane's registers (RBF order 15, Periodic order 10, the
// d = 11 quasi-periodic config) the parallelism is turned around: ONE WAVEFRONT owns a chunk of Lw
// consecutive time steps and its 64 lanes share every matrix operation, with all operands resident
// in LDS (a pool of d x d slots per wave).  The scan has three levels:
//
//   level 1  one wave per chunk of Lw steps          (N / Lw waves)     wc_reduce1 / wc_apply1 / wc_smooth1
//   level 2  one wave per group of 64 chunk totals   (N / 64 Lw waves)  wc_reduce2 / wc_sreduce2
//   level 3  one wave folds the group totals                            wc_carry3 / wc_scarry3
//
// and runs as  wc_reduce1 -> wc_reduce2 -> wc_carry3 -> wc_apply1 (writes fms, fPs, ll partials and
// the smoothing aggregates) -> wc_sreduce2 -> wc_scarry3 -> wc_smooth1 (writes sms, sPs).
// Same algebra as pgps_math.h (filt_extend / filt_combine / filt_apply / kf_step / smth_*), written
// against LDS matrices; symmetric quantities are kept as full d x d arrays and re-symmetrised.
// MFMA is not used: the contractions are d <= 32 wide and interleaved with solves and rank-one
// updates; the kernels are bound by LDS traffic and fp64 VALU.
//
// Reference: pssgp/kalman/parallel.py (elements 13-72, operator 100-118, log-lik 135-151, smoothing
// elements 155-173, smoothing operator 176-184).
#include <hip/hip_runtime.h>

#include <cmath>

#include "pgps_internal.h"
#include "pgps_math.h"

namespace pgps {
namespace wc {

constexpr int kGroup = 64;      // level-1 totals per level-2 wave

__device__ __forceinline__ void sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

template <typename T>
__device__ __forceinline__ T wave_sum(T x) {
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) x += __shfl_xor(x, s, 64);
    return x;
}

// ---- entry-parallel helpers (all 64 lanes; operands in LDS unless noted) -----------------------
// (i, j) walks this lane's entries e = lane, lane + 64, ... of a d x d matrix without dividing per entry
__device__ __forceinline__ void wc_advance(int d, int& i, int& j) {
    j += 64;
    while (j >= d) { j -= d; ++i; }
}
#define WC_FOR_ENTRIES(d)                                                     \
    for (int e_ = lane_id(), i = e_ / (d), j = e_ - i * (d); e_ < (d) * (d); \
         e_ += 64, wc_advance((d), i, j))

template <typename T>
__device__ __forceinline__ void mat_copy(int d, const T* A, T* C) {
    for (int e = lane_id(); e < d * d; e += 64) C[e] = A[e];
}
template <typename T>
__device__ __forceinline__ void mat_zero(int d, T* C) {
    for (int e = lane_id(); e < d * d; e += 64) C[e] = T(0);
}
template <typename T>
__device__ __forceinline__ void mat_eye(int d, T* C) {
    WC_FOR_ENTRIES(d) C[i * d + j] = (i == j) ? T(1) : T(0);
}
template <typename T>
__device__ __forceinline__ void vec_copy(int d, const T* a, T* c) {
    if (lane_id() < d) c[lane_id()] = a[lane_id()];
}
template <typename T>
__device__ __forceinline__ void vec_zero(int d, T* c) {
    if (lane_id() < d) c[lane_id()] = T(0);
}

// C = op(A) op(B) (+ Add).  MODE 0: A B, 1: A B^T, 2: A^T B.  C must not alias A or B.
template <typename T, int MODE>
__device__ __forceinline__ void mm(int d, const T* __restrict__ A, const T* __restrict__ B, T* __restrict__ C,
                                   const T* __restrict__ Add = nullptr) {
    WC_FOR_ENTRIES(d) {
        T acc0 = Add ? Add[i * d + j] : T(0), acc1 = T(0);
        const T* pa = (MODE == 2) ? A + i : A + i * d;
        const T* pb = (MODE == 1) ? B + j * d : B + j;
        const int sa = (MODE == 2) ? d : 1, sb = (MODE == 1) ? 1 : d;
        int k = 0;
#pragma unroll 2
        for (; k + 4 <= d; k += 4) {        // four independent LDS reads per operand in flight
            const T a0 = pa[(k + 0) * sa], a1 = pa[(k + 1) * sa], a2 = pa[(k + 2) * sa], a3 = pa[(k + 3) * sa];
            const T b0 = pb[(k + 0) * sb], b1 = pb[(k + 1) * sb], b2 = pb[(k + 2) * sb], b3 = pb[(k + 3) * sb];
            acc0 += a0 * b0;
            acc1 += a1 * b1;
            acc0 += a2 * b2;
            acc1 += a3 * b3;
        }
        for (; k < d; ++k) acc0 += pa[k * sa] * pb[k * sb];
        C[i * d + j] = acc0 + acc1;
    }
}

// C = sym_part(C) in place
template <typename T>
__device__ __forceinline__ void symmetrise(int d, T* C) {
    WC_FOR_ENTRIES(d) {
        if (i < j) {
            const T v = T(0.5) * (C[i * d + j] + C[j * d + i]);
            C[i * d + j] = v;
            C[j * d + i] = v;
        }
    }
}

// y = A x (TRANS = false) or A^T x; optional add; y must not alias x
template <typename T, bool TRANS>
__device__ __forceinline__ void mv(int d, const T* __restrict__ A, const T* __restrict__ x, T* __restrict__ y,
                                   const T* __restrict__ add = nullptr) {
    const int i = lane_id();
    if (i < d) {
        T acc0 = add ? add[i] : T(0), acc1 = T(0);
        const T* pa = TRANS ? A + i : A + i * d;
        const int sa = TRANS ? d : 1;
        int k = 0;
#pragma unroll 2
        for (; k + 4 <= d; k += 4) {
            const T a0 = pa[(k + 0) * sa], a1 = pa[(k + 1) * sa], a2 = pa[(k + 2) * sa], a3 = pa[(k + 3) * sa];
            const T x0 = x[k], x1 = x[k + 1], x2 = x[k + 2], x3 = x[k + 3];
            acc0 += a0 * x0;
            acc1 += a1 * x1;
            acc0 += a2 * x2;
            acc1 += a3 * x3;
        }
        for (; k < d; ++k) acc0 += pa[k * sa] * x[k];
        y[i] = acc0 + acc1;
    }
}

template <typename T>
__device__ __forceinline__ T dot(int d, const T* a, const T* b) {
    const int i = lane_id();
    return wave_sum(i < d ? a[i] * b[i] : T(0));
}

// Gauss-Jordan: M X = B in place (M d x d, B d x nr, row-major with leading dimension nr).
// Partial pivoting when PIVOT.  Ends synchronised.
template <typename T, bool PIVOT>
__device__ __forceinline__ void solve(int d, T* M, T* B, int nr) {
    const int lane = lane_id();
    for (int c = 0; c < d; ++c) {
        if (PIVOT) {
            T best = T(-1);
            int row = c;
            if (lane >= c && lane < d) { best = fabs(M[lane * d + c]); row = lane; }
#pragma unroll
            for (int s = 32; s > 0; s >>= 1) {
                const T ob = __shfl_xor(best, s, 64);
                const int orow = __shfl_xor(row, s, 64);
                if (ob > best || (ob == best && orow < row)) { best = ob; row = orow; }
            }
            if (row != c) {
                for (int j = lane; j < d; j += 64) { const T t = M[c * d + j]; M[c * d + j] = M[row * d + j]; M[row * d + j] = t; }
                for (int j = lane; j < nr; j += 64) { const T t = B[c * nr + j]; B[c * nr + j] = B[row * nr + j]; B[row * nr + j] = t; }
            }
            sync();
        }
        const T inv = T(1) / M[c * d + c];
        sync();                                     // everyone has read the pivot
        for (int j = lane; j < d; j += 64) if (j != c) M[c * d + j] *= inv;
        for (int j = lane; j < nr; j += 64) B[c * nr + j] *= inv;
        sync();
        // eliminate column c from every other row (column c itself is left as it is: never read again)
        const int wcols = d - c - 1 + nr;           // columns c+1..d-1 of M, then all of B
        for (int e = lane; e < d * wcols; e += 64) {
            const int r = e / wcols, q = e - r * wcols;
            if (r == c) continue;
            const T f = M[r * d + c];
            if (q < d - c - 1) M[r * d + c + 1 + q] -= f * M[c * d + c + 1 + q];
            else B[r * nr + (q - (d - c - 1))] -= f * B[c * nr + (q - (d - c - 1))];
        }
        sync();
    }
}

// ---- element records in global memory ------------------------------------------------------------
// filter 5-tuple: [A d^2 | C d^2 | J d^2 | b d | eta d];  smoother 3-tuple: [E d^2 | L d^2 | g d]
__host__ __device__ inline int nfilt(int d) { return 3 * d * d + 2 * d; }
__host__ __device__ inline int nsmth(int d) { return 2 * d * d + d; }

template <typename T>
__device__ __forceinline__ void rec_copy(int n, const T* src, T* dst) {
    for (int e = lane_id(); e < n; e += 64) dst[e] = src[e];
}

template <typename T>
struct Filt {           // views into one nfilt(d) record (LDS or global)
    T *A, *C, *J, *b, *eta;
    __device__ Filt(T* p, int d) : A(p), C(p + d * d), J(p + 2 * d * d), b(p + 3 * d * d), eta(p + 3 * d * d + d) {}
};
template <typename T>
struct Smth {
    T *E, *L, *g;
    __device__ Smth(T* p, int d) : E(p), L(p + d * d), g(p + 2 * d * d) {}
};

template <typename T>
__device__ __forceinline__ void filt_set_identity(int d, T* rec) {
    Filt<T> f(rec, d);
    mat_eye(d, f.A);
    mat_zero(d, f.C);
    mat_zero(d, f.J);
    vec_zero(d, f.b);
    vec_zero(d, f.eta);
}
template <typename T>
__device__ __forceinline__ void smth_set_identity(int d, T* rec) {
    Smth<T> s(rec, d);
    mat_eye(d, s.E);
    mat_zero(d, s.L);
    vec_zero(d, s.g);
}

// ---- kernel arguments ------------------------------------------------------------------------------
template <typename T>
struct WcArgs {
    long N;
    int d, Lw;
    long nchunk;            // level-1 chunks
    int ngroup;             // level-2 groups
    const T *P0, *H;
    T R;
    const T *Fs, *Qs, *ys;
    T *fms, *fPs, *sms, *sPs;
    double* ll;
    // workspace
    T* agg1;                // (nchunk, nfilt)
    T* lpre1;               // (nchunk, nfilt)  exclusive prefix of agg1 inside its group
    T* agg2;                // (ngroup, nfilt)
    T* carry2;              // (ngroup, d + d^2) filtered (m, P) entering each group
    T* sagg1;               // (nchunk, nsmth)
    T* lsuf1;               // (nchunk, nsmth)  exclusive suffix of sagg1 inside its group
    T* sagg2;               // (ngroup, nsmth)
    T* scarry2;             // (ngroup, d + d^2) smoothed (m, P) of the first step after each group
    double* llpart;         // (nchunk,)
};

// load F_k, symmetrised Q_k into LDS
template <typename T>
__device__ __forceinline__ void load_step(int d, const T* Fg, const T* Qg, T* F, T* Q) {
    for (int e = lane_id(); e < d * d; e += 64) { F[e] = Fg[e]; Q[e] = Qg[e]; }
    sync();
    symmetrise(d, Q);
    sync();
}

// acc <- acc (x) raw step (F, Q, y): predict the conditional, scalar-innovation update (filt_extend)
template <typename T>
__device__ __forceinline__ void extend(int d, T* acc, const T* F, const T* Q, T y, const T* h, T R, T* t1, T* t2,
                                       T* v1, T* v2, T* v3) {
    Filt<T> a(acc, d);
    mm<T, 0>(d, F, a.A, t1);                // A' = F A
    mv<T, false>(d, F, a.b, v1);            // b' = F b
    mm<T, 0>(d, F, a.C, t2);                // F C
    sync();
    mat_copy(d, t1, a.A);
    mm<T, 1>(d, t2, F, a.C, Q);             // C' = F C F^T + Q   (a.C is no longer an input)
    vec_copy(d, v1, a.b);
    sync();
    symmetrise(d, a.C);
    sync();
    if (y != y) return;
    mv<T, false>(d, a.C, h, v2);            // u = C' h
    mv<T, true>(d, a.A, h, v3);             // v = (h A')^T
    sync();
    const T S = dot(d, h, v2) + R;
    const T hb = dot(d, h, a.b);
    const T inv = T(1) / S;
    const T res = y - hb;
    WC_FOR_ENTRIES(d) {
        const T ui = v2[i], uj = v2[j], vi = v3[i], vj = v3[j];
        a.A[i * d + j] -= ui * inv * vj;
        a.C[i * d + j] -= ui * uj * inv;
        a.J[i * d + j] += vi * vj * inv;
    }
    if (lane_id() < d) {
        a.b[lane_id()] += v2[lane_id()] * inv * res;
        a.eta[lane_id()] += v3[lane_id()] * res * inv;
    }
    sync();
}

// first element of the series: update of the prior without a predict (filt_first)
template <typename T>
__device__ __forceinline__ void first_element(int d, T* acc, const T* P0, T y, const T* h, T R, T* v2) {
    Filt<T> a(acc, d);
    mat_zero(d, a.A);
    mat_zero(d, a.J);
    vec_zero(d, a.b);
    vec_zero(d, a.eta);
    mat_copy(d, P0, a.C);
    sync();
    symmetrise(d, a.C);
    sync();
    if (y != y) return;
    mv<T, false>(d, a.C, h, v2);
    sync();
    const T S = dot(d, h, v2) + R;
    const T inv = T(1) / S;
    WC_FOR_ENTRIES(d) a.C[i * d + j] -= v2[i] * v2[j] * inv;
    if (lane_id() < d) a.b[lane_id()] = v2[lane_id()] * y * inv;
    sync();
}

// out = e1 (x) e2 (filt_combine).  scratch: M (d^2), rhs (d x (2d+1)), X (d^2), vtmp (2d)
template <typename T>
__device__ __forceinline__ void combine(int d, const T* r1, const T* r2, T* rout, T* M, T* rhs, T* X, T* vt) {
    Filt<T> e1(const_cast<T*>(r1), d), e2(const_cast<T*>(r2), d), o(rout, d);
    const int nr = 2 * d + 1;
    // M = I + C1 J2 ; rhs = [A1 | C1 | b1 + C1 eta2]
    WC_FOR_ENTRIES(d) {
        T acc = (i == j) ? T(1) : T(0);
        for (int k = 0; k < d; ++k) acc += e1.C[i * d + k] * e2.J[k * d + j];
        M[i * d + j] = acc;
        rhs[i * nr + j] = e1.A[i * d + j];
        rhs[i * nr + d + j] = e1.C[i * d + j];
    }
    if (lane_id() < d) {
        const int i = lane_id();
        T acc = e1.b[i];
        for (int k = 0; k < d; ++k) acc += e1.C[i * d + k] * e2.eta[k];
        rhs[i * nr + 2 * d] = acc;
    }
    sync();
    solve<T, true>(d, M, rhs, nr);          // rhs = [G | Nm | w]
    // A = A2 G ; X = A2 Nm ; b = A2 w + b2
    WC_FOR_ENTRIES(d) {
        T acc = T(0), acx = T(0);
        for (int k = 0; k < d; ++k) {
            const T a2 = e2.A[i * d + k];
            acc += a2 * rhs[k * nr + j];
            acx += a2 * rhs[k * nr + d + j];
        }
        o.A[i * d + j] = acc;
        X[i * d + j] = acx;
    }
    if (lane_id() < d) {
        const int i = lane_id();
        T acc = e2.b[i], z = e2.eta[i];
        for (int k = 0; k < d; ++k) { acc += e2.A[i * d + k] * rhs[k * nr + 2 * d]; z -= e2.J[i * d + k] * e1.b[k]; }
        o.b[i] = acc;
        vt[i] = z;                          // z = eta2 - J2 b1
    }
    sync();
    // C = X A2^T + C2 ; M <- J2 A1
    WC_FOR_ENTRIES(d) {
        T acc = e2.C[i * d + j], acy = T(0);
        for (int k = 0; k < d; ++k) { acc += X[i * d + k] * e2.A[j * d + k]; acy += e2.J[i * d + k] * e1.A[k * d + j]; }
        o.C[i * d + j] = acc;
        M[i * d + j] = acy;
    }
    sync();
    // J = G^T (J2 A1) + J1 ; eta = G^T z + eta1
    WC_FOR_ENTRIES(d) {
        T acc = e1.J[i * d + j];
        for (int k = 0; k < d; ++k) acc += rhs[k * nr + i] * M[k * d + j];
        o.J[i * d + j] = acc;
    }
    if (lane_id() < d) {
        const int i = lane_id();
        T acc = e1.eta[i];
        for (int k = 0; k < d; ++k) acc += rhs[k * nr + i] * vt[k];
        o.eta[i] = acc;
    }
    sync();
    symmetrise(d, o.C);
    symmetrise(d, o.J);
    sync();
}

// (m, P) <- (m, P) pushed through aggregate r2 (filt_apply).  scratch: M, rhs (d x (d+1)), X
template <typename T>
__device__ __forceinline__ void apply(int d, T* m, T* P, const T* r2, T* M, T* rhs, T* X) {
    Filt<T> e2(const_cast<T*>(r2), d);
    const int nr = d + 1;
    WC_FOR_ENTRIES(d) {
        T acc = (i == j) ? T(1) : T(0);
        for (int k = 0; k < d; ++k) acc += P[i * d + k] * e2.J[k * d + j];
        M[i * d + j] = acc;
        rhs[i * nr + j] = P[i * d + j];
    }
    if (lane_id() < d) {
        const int i = lane_id();
        T acc = m[i];
        for (int k = 0; k < d; ++k) acc += P[i * d + k] * e2.eta[k];
        rhs[i * nr + d] = acc;
    }
    sync();
    solve<T, true>(d, M, rhs, nr);
    WC_FOR_ENTRIES(d) {
        T acx = T(0);
        for (int k = 0; k < d; ++k) acx += e2.A[i * d + k] * rhs[k * nr + j];
        X[i * d + j] = acx;
    }
    if (lane_id() < d) {
        const int i = lane_id();
        T acc = e2.b[i];
        for (int k = 0; k < d; ++k) acc += e2.A[i * d + k] * rhs[k * nr + d];
        m[i] = acc;
    }
    sync();
    mm<T, 1>(d, X, e2.A, P, e2.C);
    sync();
    symmetrise(d, P);
    sync();
}

// out = a (x) b in time order (a earlier): E = Ea Eb, g = Ea gb + ga, L = Ea Lb Ea^T + La.  scratch X
template <typename T>
__device__ __forceinline__ void scombine(int d, const T* ra, const T* rb, T* rout, T* X) {
    Smth<T> a(const_cast<T*>(ra), d), b(const_cast<T*>(rb), d), o(rout, d);
    mm<T, 0>(d, a.E, b.E, o.E);
    mm<T, 0>(d, a.E, b.L, X);
    mv<T, false>(d, a.E, b.g, o.g, a.g);
    sync();
    mm<T, 1>(d, X, a.E, o.L, a.L);
    sync();
    symmetrise(d, o.L);
    sync();
}

// (sm, sP) after an aggregate -> at its first step: sm = E sm + g, sP = E sP E^T + L.  scratch X, v
template <typename T>
__device__ __forceinline__ void sapply(int d, const T* ra, T* sm, T* sP, T* X, T* v, T* Y) {
    Smth<T> a(const_cast<T*>(ra), d);
    mv<T, false>(d, a.E, sm, v, a.g);
    mm<T, 0>(d, a.E, sP, X);
    sync();
    vec_copy(d, v, sm);
    mm<T, 1>(d, X, a.E, Y, a.L);
    sync();
    mat_copy(d, Y, sP);
    sync();
    symmetrise(d, sP);
    sync();
}

// ---- LDS pool ---------------------------------------------------------------------------------------
template <typename T>
struct Pool {
    T* base;
    int off = 0;
    __device__ explicit Pool(T* b) : base(b) {}
    __device__ T* take(int n) { T* p = base + off; off += (n + 1) & ~1; return p; }
};

// ====================================================================================================
// level 1: reduce
// ====================================================================================================
template <typename T>
__global__ __launch_bounds__(64) void wc_reduce1(const WcArgs<T> a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int d = a.d, dd = d * d;
    Pool<T> pool(reinterpret_cast<T*>(smem));
    T* acc = pool.take(nfilt(d));
    T* F = pool.take(dd); T* Q = pool.take(dd); T* t1 = pool.take(dd); T* t2 = pool.take(dd);
    T* h = pool.take(d); T* v1 = pool.take(d); T* v2 = pool.take(d); T* v3 = pool.take(d);
    T* P0 = pool.take(dd);
    const long c = blockIdx.x;
    const long k0 = c * a.Lw, k1 = min(a.N, k0 + a.Lw);
    vec_copy(d, a.H, h);
    filt_set_identity(d, acc);
    sync();
    for (long k = k0; k < k1; ++k) {
        load_step(d, a.Fs + k * dd, a.Qs + k * dd, F, Q);
        const T y = a.ys[k];
        if (k == 0) {
            mat_copy(d, a.P0, P0);
            sync();
            first_element(d, acc, P0, y, h, a.R, v2);
        } else {
            extend(d, acc, F, Q, y, h, a.R, t1, t2, v1, v2, v3);
        }
    }
    rec_copy(nfilt(d), acc, a.agg1 + c * nfilt(d));
}

// level 2: serial combine of a group's chunk totals; stores every chunk's exclusive in-group prefix
template <typename T>
__global__ __launch_bounds__(64) void wc_reduce2(const WcArgs<T> a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int d = a.d, dd = d * d, nf = nfilt(d);
    Pool<T> pool(reinterpret_cast<T*>(smem));
    T* acc = pool.take(nf); T* cur = pool.take(nf); T* out = pool.take(nf);
    T* M = pool.take(dd); T* rhs = pool.take(d * (2 * d + 1)); T* X = pool.take(dd); T* vt = pool.take(2 * d);
    const int g = blockIdx.x;
    const long c0 = (long)g * kGroup, c1 = min(a.nchunk, c0 + kGroup);
    filt_set_identity(d, acc);
    sync();
    for (long c = c0; c < c1; ++c) {
        rec_copy(nf, acc, a.lpre1 + c * nf);
        rec_copy(nf, a.agg1 + c * nf, cur);
        sync();
        if (c == c0) {
            rec_copy(nf, cur, acc);
        } else {
            combine(d, acc, cur, out, M, rhs, X, vt);
            rec_copy(nf, out, acc);
        }
        sync();
    }
    rec_copy(nf, acc, a.agg2 + (long)g * nf);
}

// level 3: one wave carries (m, P) across the groups
template <typename T>
__global__ __launch_bounds__(64) void wc_carry3(const WcArgs<T> a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int d = a.d, dd = d * d, nf = nfilt(d);
    Pool<T> pool(reinterpret_cast<T*>(smem));
    T* m = pool.take(d); T* P = pool.take(dd); T* cur = pool.take(nf);
    T* M = pool.take(dd); T* rhs = pool.take(d * (d + 1)); T* X = pool.take(dd);
    vec_zero(d, m);
    mat_copy(d, a.P0, P);
    sync();
    symmetrise(d, P);
    sync();
    for (int g = 0; g < a.ngroup; ++g) {
        T* out = a.carry2 + (long)g * (d + dd);
        vec_copy(d, m, out);
        mat_copy(d, P, out + d);
        rec_copy(nf, a.agg2 + (long)g * nf, cur);
        sync();
        apply(d, m, P, cur, M, rhs, X);
    }
}

// ====================================================================================================
// level 1: apply -- Kalman pass over the chunk, log-likelihood, smoothing aggregate
// ====================================================================================================
template <typename T, bool SMOOTH>
__global__ __launch_bounds__(64) void wc_apply1(const WcArgs<T> a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int d = a.d, dd = d * d, nf = nfilt(d), ns = nsmth(d);
    Pool<T> pool(reinterpret_cast<T*>(smem));
    T* m = pool.take(d); T* P = pool.take(dd);
    T* F = pool.take(dd); T* Q = pool.take(dd);
    T* FP = pool.take(dd); T* Pp = pool.take(dd); T* Ee = pool.take(ns); T* X = pool.take(dd);
    T* sacc = pool.take(ns); T* sout = pool.take(ns);
    T* h = pool.take(d); T* mp = pool.take(d); T* u = pool.take(d); T* mprev = pool.take(d);
    T* Pprev = pool.take(dd);
    T* rec = pool.take(nf);                             // this chunk's in-group prefix
    T* rhsA = pool.take(d * (d + 1));
    const long c = blockIdx.x;
    const int g = (int)(c / kGroup);
    const long k0 = c * a.Lw, k1 = min(a.N, k0 + a.Lw);
    vec_copy(d, a.H, h);
    {   // state entering the chunk: group carry pushed through the in-group prefix
        const T* cg = a.carry2 + (long)g * (d + dd);
        vec_copy(d, cg, m);
        mat_copy(d, cg + d, P);
        rec_copy(nf, a.lpre1 + c * nf, rec);
        sync();
        apply(d, m, P, rec, Pp, rhsA, X);               // Pp, X are free until the loop starts
    }
    if (SMOOTH) smth_set_identity(d, sacc);
    sync();
    double quad = 0.0, mant = 1.0;
    long long expo = 0, count = 0;
    for (long k = k0; k <= k1; ++k) {
        const bool halo = (k == k1);
        if (halo && (!SMOOTH || k == a.N)) break;
        load_step(d, a.Fs + k * dd, a.Qs + k * dd, F, Q);
        if (SMOOTH) { vec_copy(d, m, mprev); mat_copy(d, P, Pprev); }
        // predict
        mv<T, false>(d, F, m, mp);
        mm<T, 0>(d, F, P, FP);
        sync();
        mm<T, 1>(d, FP, F, Pp, Q);
        sync();
        symmetrise(d, Pp);
        sync();
        if (SMOOTH && k > k0) {
            // element of step k-1: E = (Pp^-1 F P)^T, g = m - E mp, L = P - sym(E F P)
            Smth<T> e(Ee, d);
            mat_copy(d, Pp, X);
            mat_copy(d, FP, e.L);                       // rhs (d x d) solved in place
            sync();
            solve<T, false>(d, X, e.L, d);              // e.L = Pp^-1 F P = E^T
            WC_FOR_ENTRIES(d) e.E[i * d + j] = e.L[j * d + i];
            sync();
            mv<T, false>(d, e.E, mp, u);
            mm<T, 0>(d, e.E, FP, X);
            sync();
            if (lane_id() < d) e.g[lane_id()] = mprev[lane_id()] - u[lane_id()];
            WC_FOR_ENTRIES(d) e.L[i * d + j] = Pprev[i * d + j] - T(0.5) * (X[i * d + j] + X[j * d + i]);
            sync();
            scombine(d, sacc, Ee, sout, X);
            rec_copy(ns, sout, sacc);
            sync();
        }
        if (halo) break;
        const T y = a.ys[k];
        const bool obs = !(y != y);
        const bool first = (k == 0);
        // log-likelihood term from the predicted moments (also for the first step)
        mv<T, false>(d, Pp, h, u);
        sync();
        const T S = dot(d, h, u) + a.R;
        const T mu = dot(d, h, mp);
        if (obs) {
            const double r = double(y) - double(mu);
            quad += r * r / double(S);
            int ex;
            mant = frexp(mant * double(S), &ex);
            expo += ex;
            count += 1;
        }
        if (first) {
            // update straight from the prior (m, P still hold m0 = 0, P0)
            mv<T, false>(d, P, h, u);
            sync();
            const T S0 = dot(d, h, u) + a.R;
            const T mu0 = dot(d, h, m);
            if (obs) {
                const T inv = T(1) / S0;
                WC_FOR_ENTRIES(d) P[i * d + j] -= u[i] * u[j] * inv;
                if (lane_id() < d) m[lane_id()] += u[lane_id()] * (y - mu0) * inv;
            }
        } else if (obs) {
            const T inv = T(1) / S;
            WC_FOR_ENTRIES(d) P[i * d + j] = Pp[i * d + j] - u[i] * u[j] * inv;
            if (lane_id() < d) m[lane_id()] = mp[lane_id()] + u[lane_id()] * (y - mu) * inv;
        } else {
            mat_copy(d, Pp, P);
            vec_copy(d, mp, m);
        }
        sync();
        if (lane_id() < d) a.fms[k * d + lane_id()] = m[lane_id()];
        for (int e = lane_id(); e < dd; e += 64) a.fPs[k * dd + e] = P[e];
    }
    if (SMOOTH && k1 == a.N) {
        // last element of the series: (0, m_N, P_N)
        Smth<T> e(Ee, d);
        mat_zero(d, e.E);
        mat_copy(d, P, e.L);
        vec_copy(d, m, e.g);
        sync();
        scombine(d, sacc, Ee, sout, X);
        rec_copy(ns, sout, sacc);
        sync();
    }
    if (SMOOTH) rec_copy(ns, sacc, a.sagg1 + c * ns);
    if (lane_id() == 0) {
        const double logdet = log(mant) + double(expo) * 0.6931471805599453;
        a.llpart[c] = -0.5 * (double(count) * 1.8378770664093453 + logdet + quad);
    }
}

// level 2 (smoother): serial suffix combine of a group's chunk aggregates
template <typename T>
__global__ __launch_bounds__(64) void wc_sreduce2(const WcArgs<T> a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int d = a.d, dd = d * d, ns = nsmth(d);
    Pool<T> pool(reinterpret_cast<T*>(smem));
    T* acc = pool.take(ns); T* cur = pool.take(ns); T* out = pool.take(ns); T* X = pool.take(dd);
    const int g = blockIdx.x;
    const long c0 = (long)g * kGroup, c1 = min(a.nchunk, c0 + kGroup);
    smth_set_identity(d, acc);
    sync();
    for (long c = c1 - 1; c >= c0; --c) {
        rec_copy(ns, acc, a.lsuf1 + c * ns);
        rec_copy(ns, a.sagg1 + c * ns, cur);
        sync();
        if (c == c1 - 1) {
            rec_copy(ns, cur, acc);
        } else {
            scombine(d, cur, acc, out, X);
            rec_copy(ns, out, acc);
        }
        sync();
    }
    rec_copy(ns, acc, a.sagg2 + (long)g * ns);
}

// level 3 (smoother): one wave carries (sm, sP) from the right; also sums the log-likelihood partials
template <typename T>
__global__ __launch_bounds__(64) void wc_scarry3(const WcArgs<T> a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int d = a.d, dd = d * d, ns = nsmth(d);
    Pool<T> pool(reinterpret_cast<T*>(smem));
    T* sm = pool.take(d); T* sP = pool.take(dd); T* cur = pool.take(ns);
    T* X = pool.take(dd); T* Y = pool.take(dd); T* v = pool.take(d);
    vec_zero(d, sm);
    mat_zero(d, sP);
    sync();
    for (int g = a.ngroup - 1; g >= 0; --g) {
        T* out = a.scarry2 + (long)g * (d + dd);
        vec_copy(d, sm, out);
        mat_copy(d, sP, out + d);
        rec_copy(ns, a.sagg2 + (long)g * ns, cur);
        sync();
        sapply(d, cur, sm, sP, X, v, Y);
    }
    if (a.ll) {
        double t = 0.0;
        for (long c = lane_id(); c < a.nchunk; c += 64) t += a.llpart[c];
        t = wave_sum(t);
        if (lane_id() == 0) *a.ll = t;
    }
}

// level 1 (smoother): RTS pass backwards over the chunk
template <typename T>
__global__ __launch_bounds__(64) void wc_smooth1(const WcArgs<T> a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int d = a.d, dd = d * d, ns = nsmth(d);
    Pool<T> pool(reinterpret_cast<T*>(smem));
    T* sm = pool.take(d); T* sP = pool.take(dd);
    T* F = pool.take(dd); T* Q = pool.take(dd); T* P = pool.take(dd); T* m = pool.take(d);
    T* FP = pool.take(dd); T* Pp = pool.take(dd); T* E = pool.take(dd); T* X = pool.take(dd); T* Y = pool.take(dd);
    T* mp = pool.take(d); T* v = pool.take(d); T* rec = pool.take(ns);
    const long c = blockIdx.x;
    const int g = (int)(c / kGroup);
    const long k0 = c * a.Lw, k1 = min(a.N, k0 + a.Lw);
    {
        const T* cg = a.scarry2 + (long)g * (d + dd);
        vec_copy(d, cg, sm);
        mat_copy(d, cg + d, sP);
        rec_copy(ns, a.lsuf1 + c * ns, rec);
        sync();
        sapply(d, rec, sm, sP, X, v, Y);
    }
    for (long k = k1 - 1; k >= k0; --k) {
        for (int e = lane_id(); e < dd; e += 64) P[e] = a.fPs[k * dd + e];
        if (lane_id() < d) m[lane_id()] = a.fms[k * d + lane_id()];
        sync();
        if (k == a.N - 1) {
            vec_copy(d, m, sm);
            mat_copy(d, P, sP);
            sync();
        } else {
            load_step(d, a.Fs + (k + 1) * dd, a.Qs + (k + 1) * dd, F, Q);
            mv<T, false>(d, F, m, mp);
            mm<T, 0>(d, F, P, FP);
            sync();
            mm<T, 1>(d, FP, F, Pp, Q);
            sync();
            symmetrise(d, Pp);
            sync();
            // E^T = Pp^-1 F P
            mat_copy(d, Pp, X);
            mat_copy(d, FP, Y);
            sync();
            solve<T, false>(d, X, Y, d);
            WC_FOR_ENTRIES(d) E[i * d + j] = Y[j * d + i];
            if (lane_id() < d) v[lane_id()] = sm[lane_id()] - mp[lane_id()];
            WC_FOR_ENTRIES(d) X[i * d + j] = sP[i * d + j] - Pp[i * d + j];
            sync();
            mv<T, false>(d, E, v, sm, m);               // sm = m + E (sm' - mp)
            mm<T, 0>(d, E, X, Y);
            sync();
            mm<T, 1>(d, Y, E, sP, P);                   // sP = P + E (sP' - Pp) E^T
            sync();
            symmetrise(d, sP);
            sync();
        }
        if (lane_id() < d) a.sms[k * d + lane_id()] = sm[lane_id()];
        for (int e = lane_id(); e < dd; e += 64) a.sPs[k * dd + e] = sP[e];
    }
}

// ====================================================================================================
// discretisation for d > 6: one wave per time step, Pade-13 scaling and squaring in LDS (fp64)
// ====================================================================================================
template <typename T>
__global__ __launch_bounds__(64) void wc_discretise(long N, int d, int steps_per_wave, const T* Fg, const T* Pg,
                                                    const T* ts, T t_prev, T* Fs, T* Qs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int dd = d * d;
    Pool<double> pool(reinterpret_cast<double*>(smem));
    double* A = pool.take(dd); double* A2 = pool.take(dd); double* A4 = pool.take(dd); double* A6 = pool.take(dd);
    double* W = pool.take(dd); double* U = pool.take(dd); double* V = pool.take(dd); double* Pm = pool.take(dd);
    const double b[14] = {64764752532480000., 32382376266240000., 7771770303897600., 1187353796428800.,
                          129060195264000., 10559470521600., 670442572800., 33522128640.,
                          1323241920., 40840800., 960960., 16380., 182., 1.};
    for (int e = lane_id(); e < dd; e += 64) Pm[e] = double(Pg[e]);
    sync();
    for (int q = 0; q < steps_per_wave; ++q) {
        const long k = (long)blockIdx.x * steps_per_wave + q;
        if (k >= N) break;
        const double dt = double(ts[k] - (k > 0 ? ts[k - 1] : t_prev));
        for (int e = lane_id(); e < dd; e += 64) A[e] = dt * double(Fg[e]);
        sync();
        // 1-norm -> scaling
        double col = 0.0;
        if (lane_id() < d) for (int i = 0; i < d; ++i) col += fabs(A[i * d + lane_id()]);
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) col = fmax(col, __shfl_xor(col, s, 64));
        int sq = 0;
        if (col > 5.371920351148152) {
            sq = (int)ceil(log2(col / 5.371920351148152));
            sq = sq < 0 ? 0 : (sq > 60 ? 60 : sq);
        }
        const double sc = ldexp(1.0, -sq);
        for (int e = lane_id(); e < dd; e += 64) A[e] *= sc;
        sync();
        mm<double, 0>(d, A, A, A2);
        sync();
        mm<double, 0>(d, A2, A2, A4);
        sync();
        mm<double, 0>(d, A4, A2, A6);
        sync();
        for (int e = lane_id(); e < dd; e += 64) W[e] = b[13] * A6[e] + b[11] * A4[e] + b[9] * A2[e];
        sync();
        mm<double, 0>(d, A6, W, V);
        sync();
        WC_FOR_ENTRIES(d) {
            const int e = i * d + j;
            W[e] = V[e] + b[7] * A6[e] + b[5] * A4[e] + b[3] * A2[e] + (i == j ? b[1] : 0.0);
        }
        sync();
        mm<double, 0>(d, A, W, U);
        for (int e = lane_id(); e < dd; e += 64) W[e] = b[12] * A6[e] + b[10] * A4[e] + b[8] * A2[e];
        sync();
        mm<double, 0>(d, A6, W, V);
        sync();
        WC_FOR_ENTRIES(d) {
            const int e = i * d + j;
            const double v = V[e] + b[6] * A6[e] + b[4] * A4[e] + b[2] * A2[e] + (i == j ? b[0] : 0.0);
            W[e] = v - U[e];            // M = V - U
            A2[e] = v + U[e];           // rhs = V + U
        }
        sync();
        solve<double, true>(d, W, A2, d);       // A2 = expm(A / 2^sq)
        double* R = A2;
        double* R2 = A4;
        for (int t = 0; t < sq; ++t) {
            mm<double, 0>(d, R, R, R2);
            sync();
            double* tmp = R; R = R2; R2 = tmp;
        }
        // Q = Pinf - sym(R Pinf R^T)
        mm<double, 0>(d, R, Pm, U);
        sync();
        mm<double, 1>(d, U, R, V);
        sync();
        WC_FOR_ENTRIES(d) {
            const double qv = 0.5 * (Pm[i * d + j] + Pm[j * d + i]) - 0.5 * (V[i * d + j] + V[j * d + i]);
            Qs[k * dd + i * d + j] = T(qv);
            Fs[k * dd + i * d + j] = T(R[i * d + j]);
        }
        sync();
    }
}

static __global__ __launch_bounds__(64) void wc_ll_finalize(const double* llpart, long n, double* ll) {
    double t = 0.0;
    for (long c = threadIdx.x; c < n; c += 64) t += llpart[c];
    t = wave_sum(t);
    if (threadIdx.x == 0) *ll = t;
}

}  // namespace wc

// ---- host side ----------------------------------------------------------------------------------------
static inline size_t wc_align(size_t x) { return (x + 255) / 256 * 256; }

template <typename T>
int launch_scan_wc(pgps_ctx* ctx, ScanArgs<T> sa, int d, Mode mode) {
    using namespace wc;
    if (mode == MODE_PKS || mode == MODE_SEG_REDUCE || mode == MODE_SEG_FILTER || mode == MODE_SEG_SMOOTHER)
        return PGPS_E_UNSUPPORTED_DIM;      // stand-alone pks / segments: lane-chunk family (d <= 6) only
    HIPCHK(ctx, hipSetDevice(ctx->device));
    WcArgs<T> a{};
    a.N = sa.N; a.d = d;
    a.Lw = ctx->chunk > 0 ? ctx->chunk : 32;
    a.nchunk = (sa.N + a.Lw - 1) / a.Lw;
    a.ngroup = (int)((a.nchunk + kGroup - 1) / kGroup);
    a.P0 = sa.P0; a.H = sa.H; a.R = sa.R; a.Fs = sa.Fs; a.Qs = sa.Qs; a.ys = sa.ys;
    a.fms = sa.fms; a.fPs = sa.fPs; a.sms = sa.sms; a.sPs = sa.sPs; a.ll = sa.ll;
    const size_t dd = (size_t)d * d, nf = nfilt(d), ns = nsmth(d), nc = (size_t)a.nchunk, ng = (size_t)a.ngroup;
    size_t off = 0;
    const size_t o_agg1 = off;   off = wc_align(off + nc * nf * sizeof(T));
    const size_t o_lpre1 = off;  off = wc_align(off + nc * nf * sizeof(T));
    const size_t o_agg2 = off;   off = wc_align(off + ng * nf * sizeof(T));
    const size_t o_carry2 = off; off = wc_align(off + ng * (d + dd) * sizeof(T));
    const size_t o_sagg1 = off;  off = wc_align(off + nc * ns * sizeof(T));
    const size_t o_lsuf1 = off;  off = wc_align(off + nc * ns * sizeof(T));
    const size_t o_sagg2 = off;  off = wc_align(off + ng * ns * sizeof(T));
    const size_t o_sc2 = off;    off = wc_align(off + ng * (d + dd) * sizeof(T));
    const size_t o_ll = off;     off = wc_align(off + nc * sizeof(double));
    int rc = ensure(ctx, ctx->ws, off);
    if (rc) return rc;
    char* base = (char*)ctx->ws.p;
    a.agg1 = (T*)(base + o_agg1); a.lpre1 = (T*)(base + o_lpre1); a.agg2 = (T*)(base + o_agg2);
    a.carry2 = (T*)(base + o_carry2); a.sagg1 = (T*)(base + o_sagg1); a.lsuf1 = (T*)(base + o_lsuf1);
    a.sagg2 = (T*)(base + o_sagg2); a.scarry2 = (T*)(base + o_sc2); a.llpart = (double*)(base + o_ll);

    // LDS pool sizes in scalars (upper bounds of the take() sequences above, +1 per take for alignment)
    const size_t pad = 32;
    const size_t l_reduce1 = nf + 5 * dd + 4 * d + pad;
    const size_t l_reduce2 = 3 * nf + 2 * dd + (size_t)d * (2 * d + 1) + 2 * d + pad;
    const size_t l_carry3 = d + 3 * dd + nf + (size_t)d * (d + 1) + pad;
    const size_t l_apply1 = 5 * d + 7 * dd + 3 * ns + nf + (size_t)d * (d + 1) + pad;
    const size_t l_sred2 = 3 * ns + dd + pad;
    const size_t l_scarry3 = 2 * d + 3 * dd + ns + pad;
    const size_t l_smooth1 = 4 * d + 9 * dd + ns + pad;
    auto bytes = [](size_t n) { return n * sizeof(T); };
    size_t need = 0;
    for (size_t v : {l_reduce1, l_reduce2, l_carry3, l_apply1, l_sred2, l_scarry3, l_smooth1}) need = need > v ? need : v;
    if (bytes(need) > 160 * 1024) return PGPS_E_UNSUPPORTED_DIM;
#define WC_ATTR(K, L)                                                                                              \
    HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(K), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                    (int)bytes(L)))
    WC_ATTR((wc_reduce1<T>), l_reduce1);
    WC_ATTR((wc_reduce2<T>), l_reduce2);
    WC_ATTR((wc_carry3<T>), l_carry3);
    WC_ATTR((wc_apply1<T, true>), l_apply1);
    WC_ATTR((wc_apply1<T, false>), l_apply1);
    WC_ATTR((wc_sreduce2<T>), l_sred2);
    WC_ATTR((wc_scarry3<T>), l_scarry3);
    WC_ATTR((wc_smooth1<T>), l_smooth1);
#undef WC_ATTR
    const dim3 blk(64), g1((unsigned)a.nchunk), g2((unsigned)a.ngroup);
    timed_launch(ctx, PGPS_K_FILTER_REDUCE, wc_reduce1<T>, g1, blk, (unsigned)bytes(l_reduce1), a);
    timed_launch(ctx, PGPS_K_FILTER_REDUCE, wc_reduce2<T>, g2, blk, (unsigned)bytes(l_reduce2), a);
    timed_launch(ctx, PGPS_K_FILTER_REDUCE, wc_carry3<T>, dim3(1), blk, (unsigned)bytes(l_carry3), a);
    if (mode == MODE_PKFS) {
        timed_launch(ctx, PGPS_K_FILTER_APPLY, wc_apply1<T, true>, g1, blk, (unsigned)bytes(l_apply1), a);
        timed_launch(ctx, PGPS_K_SMOOTHER_REDUCE, wc_sreduce2<T>, g2, blk, (unsigned)bytes(l_sred2), a);
        timed_launch(ctx, PGPS_K_SMOOTHER_REDUCE, wc_scarry3<T>, dim3(1), blk, (unsigned)bytes(l_scarry3), a);
        timed_launch(ctx, PGPS_K_SMOOTHER_APPLY, wc_smooth1<T>, g1, blk, (unsigned)bytes(l_smooth1), a);
    } else {
        timed_launch(ctx, PGPS_K_FILTER_APPLY, wc_apply1<T, false>, g1, blk, (unsigned)bytes(l_apply1), a);
        if (a.ll)
            timed_launch(ctx, PGPS_K_LL_FINALIZE, wc::wc_ll_finalize, dim3(1), blk, 0u, (const double*)a.llpart,
                         (long)a.nchunk, a.ll);
    }
    HIPCHK(ctx, hipGetLastError());
    return PGPS_OK;
}

template <typename T>
int launch_disc_wc(pgps_ctx* ctx, long N, int d, const T* F, const T* Pinf, const T* ts, T t0, T* Fs, T* Qs) {
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int spw = 8;
    const size_t lds = ((size_t)8 * d * d + 64) * sizeof(double);
    if (lds > 160 * 1024) return PGPS_E_UNSUPPORTED_DIM;
    HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(wc::wc_discretise<T>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const long grid = (N + spw - 1) / spw;
    timed_launch(ctx, PGPS_K_DISCRETISE, wc::wc_discretise<T>, dim3((unsigned)grid), dim3(64), (unsigned)lds, N, d, spw,
                 F, Pinf, ts, t0, Fs, Qs);
    HIPCHK(ctx, hipGetLastError());
    return PGPS_OK;
}

template int launch_disc_wc<double>(pgps_ctx*, long, int, const double*, const double*, const double*, double, double*,
                                    double*);
template int launch_disc_wc<float>(pgps_ctx*, long, int, const float*, const float*, const float*, float, float*, float*);
template int launch_scan_wc<double>(pgps_ctx*, ScanArgs<double>, int, Mode);
template int launch_scan_wc<float>(pgps_ctx*, ScanArgs<float>, int, Mode);

}  // namespace pgps
