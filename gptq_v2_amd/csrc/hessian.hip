// HessianHook.invert on the device (reference: gptq_py.py:38-66).
//
//   H *= 2/n;  dead diagonal -> 1;  perm = argsort(diag, descending);  H = H[perm][:, perm];
//   damp = ratio * mean(diag);  repeat { diag += damp;  U = chol_upper(inverse(H)) } until it works.
//
// The reference obtains U through three LAPACK calls (potrf, potri, potrf: 4C^3/3 flops).  Here
// U comes from ONE factorisation and ONE triangular inversion (2C^3/3 flops) of the index-
// reversed matrix:  with J the exchange matrix,  J H J = L L^T  =>  H = (J L J)(J L J)^T  with
// J L J upper triangular, so  H^-1 = U^T U  for  U = (J L J)^-1 = J L^-1 J,  which is upper
// triangular with a positive diagonal, i.e. exactly the Cholesky factor the reference computes
// (the factor is unique).  L and L^-1 are produced together by a recursion on 2x2 block splits
//     [A 0; B D]:   (L11, L11^-1) = rec(A);   L21 = B L11^-T;   D -= L21 L21^T;
//                   (L22, L22^-1) = rec(D);   (L^-1)21 = -L22^-1 (L21 L11^-1)
// whose products run on the fp32 MFMA GEMM (sgemm.hip) and whose 128x128 leaves are factored and
// inverted inside one workgroup in LDS.
#include <atomic>
#include <mutex>
#include <type_traits>
#include <vector>

#include "sgemm.h"

namespace {

constexpr int NB = 128;  // leaf size == GEMM tile edge (the k-range trimming relies on it)

// ---- element-wise preparation ------------------------------------------------------------

__global__ void scale_kernel(float *__restrict__ h, int64_t c, float s, float *__restrict__ diag)
{
    const int64_t total = c * c;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        float v = h[i] * s;
        const int64_t r = i / c, col = i - r * c;
        if (r == col) {
            if (v == 0.0f) v = 1.0f;   // dead input feature (gptq_py.py:43-44)
            diag[r] = v;
        }
        h[i] = v;
    }
}

// damp = ratio * mean(diag); one workgroup, double accumulation.  A diagonal that is not finite
// (fp16 activations that overflowed put inf or inf * 0 = NaN into X^T X) cannot be repaired by
// damping: *fatal is raised and the host gives up after the first try.
__global__ void damp_kernel(const float *__restrict__ diag, int64_t c, float ratio, float *__restrict__ damp_out,
                            int *__restrict__ fatal)
{
    __shared__ double part[256];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < c; i += 256) s += (double)diag[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        *damp_out = ratio * (float)(part[0] / (double)c);
        *fatal = (part[0] - part[0] == 0.0) ? 0 : 1;   // inf - inf and NaN - NaN are NaN
    }
}

// Stable descending argsort by counting: rank(i) = #{j : d[j] > d[i] or (d[j] == d[i] and j < i)}.
// The order is total: a NaN counts as smaller than every number and equal to another NaN, so the
// ranks are a permutation whatever the diagonal holds (with the plain comparisons every NaN would
// get the same rank, slots of perm would stay unwritten and the gather behind it would read out
// of bounds).  A workgroup ranks 64 elements against one of RANK_SPLIT slices of the diagonal -- its
// four waves each count one quarter of every 1024-value chunk of the slice -- and adds its count to
// rank[i] (integer atomics: the sum does not depend on their order); rank starts at zero and is
// perm_inv when all slices are in, and perm_scatter_kernel inverts it.
constexpr int RANK_SPLIT = 4;

__global__ __launch_bounds__(256) void rank_sort_kernel(const float *__restrict__ diag, int c, int32_t *__restrict__ rank_out)
{
    __shared__ float chunk[1024];
    __shared__ int part[4][64];
    const int tx = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + tx;
    const float di = i < c ? diag[i] : 0.0f;
    const bool nan_i = di != di;
    const int per = (((c + RANK_SPLIT - 1) / RANK_SPLIT) + 1023) / 1024 * 1024;   // slice length, whole chunks
    const int j_lo = blockIdx.y * per, j_hi = min(c, j_lo + per);
    int rank = 0;
    for (int base = j_lo; base < j_hi; base += 1024) {
        for (int t = threadIdx.x; t < 1024; t += 256) chunk[t] = base + t < j_hi ? diag[base + t] : 0.0f;
        __syncthreads();
        const int lo = q * 256, lim = min(lo + 256, j_hi - base);
#pragma unroll 8
        for (int t = lo; t < lim; ++t) {
            const float dj = chunk[t];
            const bool nan_j = dj != dj;
            const bool above = (dj > di) || (nan_i && !nan_j);
            const bool same = (dj == di) || (nan_i && nan_j);
            rank += above || (same && base + t < i);
        }
        __syncthreads();
    }
    part[q][tx] = rank;
    __syncthreads();
    if (q == 0 && i < c) {
        rank = part[0][tx] + part[1][tx] + part[2][tx] + part[3][tx];
        if (rank) atomicAdd(rank_out + i, rank);
    }
}

__global__ void perm_scatter_kernel(const int32_t *__restrict__ perm_inv, int c, int32_t *__restrict__ perm)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < c) perm[perm_inv[i]] = i;
}

// diag[i] = h[i][i] * s with the dead-feature fix of scale_kernel: all the sort needs, so that with
// act-order the scaling itself can ride on the gather below instead of being a pass of its own
__global__ void scaled_diag_kernel(const float *__restrict__ h, int64_t c, float s, float *__restrict__ diag)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < c) {
        float v = h[i * c + i] * s;
        if (v == 0.0f) v = 1.0f;
        diag[i] = v;
    }
}

// out[i][j] = s * h[perm[i]][perm[j]], a zero on the diagonal replaced by 1 (gptq_py.py:43-44): the scaled
// Hessian in act-order from the unscaled one in model order
__global__ void gather_kernel(const float *__restrict__ h, const int32_t *__restrict__ perm, int64_t c, float s,
                              float *__restrict__ out)
{
    const int64_t total = c * c;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / c, col = i - r * c;
        float v = h[(int64_t)perm[r] * c + perm[col]] * s;
        if (r == col && v == 0.0f) v = 1.0f;
        out[i] = v;
    }
}

// The same, one output row per workgroup and turn: the source row is read as it lies in memory into the
// LDS and the columns are picked from there.  (A lane that reads h[perm[i]][perm[j]] from memory fetches a
// 64-byte sector for 4 bytes of it whenever the order is not nearly monotone: C = 11008 took 1.2 ms so,
// against 0.25 ms for an order that is.)  VEC = 4: C is a multiple of 4 and rows are 16-byte aligned.
constexpr int64_t GATHER_LDS_MAX_C = 36 * 1024;   // 144 KiB of LDS
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int VEC>
__global__ __launch_bounds__(256) void gather_rows_kernel(const float *__restrict__ h, const int32_t *__restrict__ perm, int64_t c,
                                                          float s, float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) float gather_row[];
    for (int64_t r = blockIdx.x; r < c; r += gridDim.x) {
        const float *__restrict__ src = h + (int64_t)perm[r] * c;
        float *__restrict__ dst = out + r * c;
        __syncthreads();   // the previous row has been picked from
        if constexpr (VEC == 4) {
            for (int64_t i = threadIdx.x * 4; i < c; i += 1024) *reinterpret_cast<f32x4 *>(gather_row + i) = *reinterpret_cast<const f32x4 *>(src + i);
        } else {
            for (int64_t i = threadIdx.x; i < c; i += 256) gather_row[i] = src[i];
        }
        __syncthreads();
        if constexpr (VEC == 4) {
            for (int64_t i = threadIdx.x * 4; i < c; i += 1024) {
                const i32x4 p4 = *reinterpret_cast<const i32x4 *>(perm + i);
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = gather_row[p4[e]] * s;
                    if (i + e == r && v[e] == 0.0f) v[e] = 1.0f;
                }
                *reinterpret_cast<f32x4 *>(dst + i) = v;
            }
        } else {
            for (int64_t i = threadIdx.x; i < c; i += 256) {
                float v = gather_row[perm[i]] * s;
                if (i == r && v == 0.0f) v = 1.0f;
                dst[i] = v;
            }
        }
    }
}

// h[i][i] += damp, and the index-reversed, identity-padded work matrix  w[i][j] = h[c-1-i][c-1-j].
// With `src` the matrix is read from there and h is written whole: the first try after the act-order gather takes
// the gathered matrix from the workspace and leaves it in `hessian` on the way (a copy of its own before this
// kernel was one more read of C^2 floats).
__global__ void damp_flip_kernel(float *__restrict__ h, const float *__restrict__ src, int64_t c, int64_t cp,
                                 const float *__restrict__ damp, float *__restrict__ w)
{
    const float d = *damp;
    const int64_t total = cp * cp;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cp, col = i - r * cp;
        float v;
        if (r < c && col < c) {
            const int64_t sr = c - 1 - r, sc = c - 1 - col;
            v = src ? src[sr * c + sc] : h[sr * c + sc];
            if (sr == sc) v += d;
            if (src || sr == sc) h[sr * c + sc] = v;
        } else {
            v = r == col ? 1.0f : 0.0f;
        }
        w[i] = v;
    }
}

// u[i][j] = linv[c-1-i][c-1-j] on and above the diagonal, 0 below; nothing is written when the
// factorisation before it raised *status (the host then damps again, without a round trip in between)
__global__ void flip_out_kernel(const float *__restrict__ linv, int64_t c, int64_t cp, float *__restrict__ u,
                                const int *__restrict__ status)
{
    if (*status != 0) return;
    const int64_t total = c * c;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / c, col = i - r * c;
        u[i] = col >= r ? linv[(c - 1 - r) * cp + (c - 1 - col)] : 0.0f;
    }
}

// ---- 128x128 leaf: L = chol(A) and L^-1, one workgroup, everything in LDS ------------------
// Reads the lower triangle of w[o:o+128, o:o+128]; writes the full 128x128 block of linv (zeros
// above the diagonal).  A pivot that is not a positive finite number raises *status.
//
// Cholesky: 8 panels of 16 columns.  Per panel (a) one wave factors the 16x16 diagonal block and
// inverts it with the rows on lanes and every broadcast a DPP row broadcast (no LDS, no barrier),
// (b) the panel below it,  A21 * L11^-T, and (c) the rank-16 update of the trailing lower triangle
// run on 16x16 blocks of v_mfma_f32_16x16x4_f32, one block per wave and turn.  (a) is a chain of
// dependent instructions on one wave and the longest phase, so only the part of (c) that the next
// panel reads -- its own 16 columns, (c1) -- stands between (b) and the next (a); the rest, (c2),
// runs on waves 1-3 beside that (a).  Inverse: the 16x16 diagonal inverses are doubled three times,
// inv([A 0; B D]) = [A^-1 0; -D^-1 (B A^-1) D^-1], on the same MFMA blocks with every fragment of
// a block read before its first MFMA.
// Cycles of one leaf (tools/trace_leaf.py): 84k with scalar-FMA panels and updates and everything in
// sequence, 61-62k now -- factorisations of the diagonal blocks 16-17k (about 130 cycles per pivot),
// their inverses 13k, panels and (c1) 5-6k each, the two passes of the doubling 5k each (15k before
// the zeros of the triangular operands were skipped), load and store 6k.  Measured and dropped: the
// inverse of a diagonal block on wave 1 beside the NEXT factorisation, with the panels as triangular
// solves (one thread per row) so that nothing waits for it -- 71k: a solve is 120 broadcast LDS reads
// and a 32-deep chain per row (1.9k cycles per panel against 0.7k for the MFMA form), and the inverse
// read back from LDS on another wave took longer than the factorisation beside it.
// Also measured and dropped: the inverse of a diagonal block woven into its factorisation instruction by instruction
// (the updates that are not on the pivot's chain and the inverse's step j-1 dealt by hand into the issue slots between
// the chain's dependent instructions; bit-identical U): 25.3k cycles for the woven loop against 16.7k + 5.5k for the two
// loops one after the other -- a DPP fmac holds the issue port for about 8 cycles, more than the gap between two
// dependent instructions of the chain, so every filler delays the chain.  (Of the 13k cycles booked on "inverse" above,
// 7.5k are the LDS writes of L, its inverse and the transposed copy: 36 stores per lane and sub-panel.)
constexpr int LLD = 132;                 // LDS row stride (floats): 16-byte aligned rows, 16 rows cover all banks
constexpr int PB = 16;                   // panel width
constexpr int SLD = 68;                  // stride of the 64x64 product scratch
constexpr int LEAF_LDS = (2 * NB * LLD + 64 * SLD + PB * PB) * 4;

// v_readlane on a float (the builtin is typed int: pass the bits, not the value)
__device__ __forceinline__ float readlane_f32(float v, int lane)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// lane K of every 16-lane row, broadcast to that row (DPP row_newbcast)
template <int K>
__device__ __forceinline__ float row_bcast(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + K, 0xf, 0xf, false));
}

// acc += (lane K of v, within every 16-lane row) * m as ONE instruction: the row broadcast is the DPP operand
// of a v_fmac (through the builtin hipcc emits a zeroing move, a DPP move and a VOP3 fma: three instructions
// per update of the chain below).  A DPP operand may not have been written by the two instructions in front of
// it and hipcc does not look into inline asm, so FRESH = true (v has just been produced) waits by itself.
template <int K, bool FRESH = false>
__device__ __forceinline__ void fmac_bcast(float &acc, float v, float m)
{
    if constexpr (FRESH)
        asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(v), "v"(m), "n"(K));
    else
        asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(v), "v"(m), "n"(K));
}

// lane K of every 16-lane row of a value the asm above may just have written
template <int K>
__device__ __forceinline__ float row_bcast_fresh(float v)
{
    float out;
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(out) : "v"(v), "n"(K));
    return out;
}

template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// 1 / sqrt(x): the hardware estimate and one Newton step
__device__ __forceinline__ float rsqrt_nr(float x)
{
    const float r0 = __builtin_amdgcn_rsqf(x);
    const float e = fmaf(-__fmul_rn(x, r0), r0, 1.0f);
    return fmaf(__fmul_rn(0.5f, r0), e, r0);
}

__device__ __forceinline__ f32x4 lds4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
__device__ __forceinline__ void st4(float *p, f32x4 v) { *reinterpret_cast<f32x4 *>(p) = v; }

// lower-triangular tile index t -> (ti, tj), ti >= tj
__device__ __forceinline__ void tri_index(int t, int &ti, int &tj)
{
    int r = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
    while ((r + 1) * (r + 2) / 2 <= t) ++r;
    while (r * (r + 1) / 2 > t) --r;
    ti = r;
    tj = t - r * (r + 1) / 2;
}

// N 16x16 blocks of the trailing update on one wave: a[r0.., q0..] -= L[r0.., k0..k0+16) * L[q0.., k0..k0+16)^T,
// four v_mfma_f32_16x16x4_f32 per block with the old block as the accumulator input (lane l feeds
// A[row l & 15][k = l >> 4] and B[k = l >> 4][column l & 15] and owns D[4 (l >> 4) + e][l & 15]).  All reads
// of the N blocks are issued first and their MFMAs interleaved: one block alone is an LDS round trip, a chain
// of four dependent MFMAs and the writes, about 700 cycles of latency.  With the row stride of 132 floats all
// four access patterns are conflict-free.
template <int N>
__device__ __forceinline__ void trail_n(float *a, int k0, const int *r0, const int *q0, int l15, int lq)
{
    f32x4 acc[N];
    float av[N][4], bv[N][4];
#pragma unroll
    for (int n = 0; n < N; ++n) {
        const float *d = a + (r0[n] + 4 * lq) * LLD + q0[n] + l15;
        acc[n] = f32x4{d[0], d[LLD], d[2 * LLD], d[3 * LLD]};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            av[n][q] = -a[(r0[n] + l15) * LLD + k0 + 4 * q + lq];
            bv[n][q] = a[(q0[n] + l15) * LLD + k0 + 4 * q + lq];
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int n = 0; n < N; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[n][q], bv[n][q], acc[n], 0, 0, 0);
#pragma unroll
    for (int n = 0; n < N; ++n) {
        float *d = a + (r0[n] + 4 * lq) * LLD + q0[n] + l15;
        d[0] = acc[n][0]; d[LLD] = acc[n][1]; d[2 * LLD] = acc[n][2]; d[3 * LLD] = acc[n][3];
    }
}

// `count` (1..4, wave-uniform) blocks at once
__device__ __forceinline__ void trail_blocks(float *a, int k0, const int *r0, const int *q0, int count, int l15, int lq)
{
    if (count == 4) trail_n<4>(a, k0, r0, q0, l15, lq);
    else if (count == 3) trail_n<3>(a, k0, r0, q0, l15, lq);
    else if (count == 2) trail_n<2>(a, k0, r0, q0, l15, lq);
    else if (count == 1) trail_n<1>(a, k0, r0, q0, l15, lq);
}

#ifdef GQ_LEAF_TRACE
// Debug build only (tools/trace_leaf.py): cycles of thread 0 per phase, summed over the panels.
__device__ unsigned long long *g_leaf_trace = nullptr;
#define LF_T0() unsigned long long lf_t = __builtin_readcyclecounter(), lf_acc[8] = {}
#define LF_ADD(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); lf_acc[i] += n_ - lf_t; lf_t = n_; } while (0)
#else
#define LF_T0()
#define LF_ADD(i)
#endif

__global__ __launch_bounds__(256) void leaf_kernel(const float *__restrict__ w, float *__restrict__ linv, int64_t ld,
                                                   int64_t o, int *__restrict__ status)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float *a = reinterpret_cast<float *>(smem_raw);   // [128][LLD]  A, becomes L
    float *x = a + NB * LLD;                          // [128][LLD]  L^-1
    float *sc = x + NB * LLD;                         // [64][SLD]   product scratch
    float *tinv = sc + 64 * SLD;                      // [16][16]    inverse of the current diagonal block
    const int tid = threadIdx.x;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    LF_T0();
    {
        f32x4 v[16];   // the 16 loads of a thread in flight together
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int e = tid + 256 * q, r = e >> 5, c4 = (e & 31) * 4;
            v[q] = *reinterpret_cast<const f32x4 *>(w + (o + r) * ld + o + c4);
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int e = tid + 256 * q, r = e >> 5, c4 = (e & 31) * 4;
            st4(a + r * LLD + c4, v[q]);
            st4(x + r * LLD + c4, f32x4{0.f, 0.f, 0.f, 0.f});
        }
    }
    __syncthreads();
    LF_ADD(0);
    bool bad = false;

    const int l15 = tid & 15, lq = (tid & 63) >> 4;
    for (int p = 0; p < NB / PB; ++p) {
        const int c0 = p * PB, c1 = c0 + PB;
        // (c2) of the previous panel on waves 1-3: its update of every 16x16 block right of this panel's columns
        // (block rows and columns counted from c1; the lower triangle of an (n x n) block grid)
        if (tid >= 64 && p > 0) {
            const int n = (NB - c1) / PB, total = n * (n + 1) / 2;
            for (int t0 = wid - 1; t0 < total; t0 += 12) {   // the wave's blocks t0, t0 + 3, ..., four at a time
                int r0[4], q0[4], count = 0;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int t = min(t0 + 3 * u, total - 1);
                    int ti, tj;
                    tri_index(t, ti, tj);
                    r0[u] = c1 + PB * ti;
                    q0[u] = c1 + PB * tj;
                    count += t0 + 3 * u < total;
                }
                trail_blocks(a, c0 - PB, r0, q0, count, l15, lq);
            }
        }
        // (a) diagonal block: lane i (< 16) of wave 0 holds row i
        if (tid < 64) {
            const int i = tid & 15;
            float d[PB];
#pragma unroll
            for (int k4 = 0; k4 < PB; k4 += 4) {
                const f32x4 v = lds4(a + (c0 + i) * LLD + c0 + k4);
                d[k4] = v[0]; d[k4 + 1] = v[1]; d[k4 + 2] = v[2]; d[k4 + 3] = v[3];
            }
            // Right-looking factorisation, one pivot at a time.  Every 16-lane row of the wave holds the
            // same 16 matrix rows, so the broadcasts are DPP row broadcasts folded into the fmas; the
            // pivot's reciprocal square root (estimate + one Newton step) replaces the square root
            // and the division, and is kept: the inverse below multiplies by it.
            float rinv[PB];
            static_for<0, PB>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const float dj = row_bcast_fresh<j>(d[j]);
                if (!(dj > 0.0f) || !(dj < 3.0e38f)) bad = true;
                const float rj = rsqrt_nr(dj);
                const float lj = __fmul_rn(d[j], rj);   // lane j: dj / sqrt(dj) = sqrt(dj)
                d[j] = lj;
                rinv[j] = rj;
                const float nlj = -lj;
                static_for<j + 1, PB>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    fmac_bcast<k, k == j + 1>(d[k], lj, nlj);   // d[k] -= L[i][j] * L[k][j]
                });
            });
            LF_ADD(1);
            // inverse: lane c holds column c, xi[r] = (L11^-1)[r][c]
            // (column by column: once xi[k] is known every later row takes its term, so the 15 - k updates are
            // independent of each other and only xi[k] -> the update of row k + 1 -> xi[k + 1] is a chain; each
            // row's sum still adds its terms in ascending k)
            float xi[PB];
#pragma unroll
            for (int r = 0; r < PB; ++r) xi[r] = i == r ? 1.0f : 0.0f;
            static_for<0, PB>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                xi[k] = __fmul_rn(xi[k], rinv[k]);
                const float nxk = -xi[k];
                static_for<k + 1, PB>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    fmac_bcast<r>(xi[r], d[k], nxk);   // xi[r] -= L[r][k] * xi[k]
                });
            });
            if (tid < PB) {
#pragma unroll
                for (int k4 = 0; k4 < PB; k4 += 4) {
                    // zero the strict upper part so that later full-row reads see a clean triangle
                    f32x4 v;
                    v[0] = k4 <= i ? d[k4] : 0.f; v[1] = k4 + 1 <= i ? d[k4 + 1] : 0.f;
                    v[2] = k4 + 2 <= i ? d[k4 + 2] : 0.f; v[3] = k4 + 3 <= i ? d[k4 + 3] : 0.f;
                    st4(a + (c0 + i) * LLD + c0 + k4, v);
                }
#pragma unroll
                for (int r = 0; r < PB; ++r) {
                    tinv[r * PB + i] = xi[r];
                    x[(c0 + r) * LLD + c0 + i] = xi[r];
                }
            }
        }
        LF_ADD(6);
        __syncthreads();
        LF_ADD(7);
        if (c1 == NB) break;
        // (b) panel below the diagonal block, L21 = A21 * L11^-T, one 16-row block per wave and turn on four
        // v_mfma_f32_16x16x4_f32 (B[k][j] = L11^-1[j][k]; its zeros above the diagonal are multiplied through).
        // A block replaces its own input: the wave has read its four fragments before it writes.
        const int nrb = (NB - c1) / PB;   // 16-row blocks below the diagonal block
        if (wid < nrb) {   // the wave's blocks wid and wid + 4 (at most seven blocks), both at once
            const bool two = wid + 4 < nrb;
            const int ra = c1 + PB * wid, rb = c1 + PB * (two ? wid + 4 : wid);
            float av[2][4], bv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                av[0][q] = a[(ra + l15) * LLD + c0 + 4 * q + lq];
                av[1][q] = a[(rb + l15) * LLD + c0 + 4 * q + lq];
                bv[q] = tinv[l15 * PB + 4 * q + lq];
            }
            f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0][q], bv[q], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1][q], bv[q], acc[1], 0, 0, 0);
            }
            float *dst = a + (ra + 4 * lq) * LLD + c0 + l15;
            dst[0] = acc[0][0]; dst[LLD] = acc[0][1]; dst[2 * LLD] = acc[0][2]; dst[3 * LLD] = acc[0][3];
            if (two) {
                dst = a + (rb + 4 * lq) * LLD + c0 + l15;
                dst[0] = acc[1][0]; dst[LLD] = acc[1][1]; dst[2 * LLD] = acc[1][2]; dst[3 * LLD] = acc[1][3];
            }
        }
        __syncthreads();
        LF_ADD(2);
        // (c1) the next panel's columns (its diagonal block and everything below) -= L21 L21^T: the only part of
        // the trailing update the next diagonal block and panel wait for.  The rest, (c2), runs beside the next
        // diagonal block at the top of the loop.
        if (wid < nrb) {
            const int r0[2] = {c1 + PB * wid, c1 + PB * (wid + 4)}, q0[2] = {c1, c1};
            trail_blocks(a, c0, r0, q0, wid + 4 < nrb ? 2 : 1, l15, lq);
        }
        __syncthreads();
        LF_ADD(3);
    }
    if (__syncthreads_or(bad) && tid == 0) atomicOr(status, 1);

    // inverse by doubling: blocks of size s at b and b+s -> X21 = -X22 * (L21 * X11), both products on
    // 16x16 blocks of v_mfma_f32_16x16x4_f32, dealt round-robin to the four waves (lane l feeds
    // A[row l & 15][k = l >> 4] and B[k = l >> 4][column l & 15] and owns D[4 (l >> 4) + e][l & 15]);
    // operands are read as stored: the zeros above the diagonals of X11 and X22 are multiplied
    // through rather than skipped
    static_for<0, 3>([&](auto level) {
        constexpr int s = PB << decltype(level)::value;   // 16, 32, 64
        constexpr int bpd = s / 16, nq = s / 4;           // 16x16 blocks per dimension of a half, MFMAs per full K range
        // Every level has exactly four (pair, block row) and four (pair, block column) combinations: one per wave.
        const int pair = wid / bpd, mine = wid % bpd;
        const int b = pair * 2 * s;
        // pass 1: T = L21 * X11 into the scratch, pairs stacked by rows (n_pairs * s <= 64 rows).  The wave owns
        // block row `mine` of its pair: its L21 fragments serve all block columns tj, and X11 is lower
        // triangular, so column tj only needs k >= 16 tj.  All fragments are read first; the MFMAs of the
        // block columns are interleaved so that no MFMA waits for its predecessor.
        {
            float av[nq], bv[bpd][nq];
            const float *pl = a + (b + s + 16 * mine + l15) * LLD + b + lq;      // A[i][k] = L21[i][k]
#pragma unroll
            for (int q = 0; q < nq; ++q) av[q] = pl[4 * q];
            static_for<0, bpd>([&](auto tjc) {
                constexpr int tj = decltype(tjc)::value;
                const float *px = x + (b + lq) * LLD + b + 16 * tj + l15;        // B[k][j] = X11[k][j]
#pragma unroll
                for (int q = 4 * tj; q < nq; ++q) bv[tj][q] = px[4 * q * LLD];
            });
            f32x4 acc[bpd];
#pragma unroll
            for (int tj = 0; tj < bpd; ++tj) acc[tj] = f32x4{0.f, 0.f, 0.f, 0.f};
            static_for<0, nq>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                static_for<0, q / 4 + 1>([&](auto tjc) {
                    constexpr int tj = decltype(tjc)::value;
                    acc[tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], bv[tj][q], acc[tj], 0, 0, 0);
                });
            });
#pragma unroll
            for (int tj = 0; tj < bpd; ++tj) {
                float *dst = sc + (pair * s + 16 * mine + 4 * lq) * SLD + 16 * tj + l15;
                dst[0] = acc[tj][0]; dst[SLD] = acc[tj][1]; dst[2 * SLD] = acc[tj][2]; dst[3 * SLD] = acc[tj][3];
            }
        }
        __syncthreads();
        LF_ADD(4);
        // pass 2: X21 = -X22 * T.  The wave owns block column `mine`: its T fragments serve all block rows ti, and
        // X22 is lower triangular, so row ti only needs k < 16 (ti + 1).
        {
            float bt[nq], ax[bpd][nq];
            const float *pt = sc + (pair * s + lq) * SLD + 16 * mine + l15;      // B[k][j] = T[k][j]
#pragma unroll
            for (int q = 0; q < nq; ++q) bt[q] = pt[4 * q * SLD];
            static_for<0, bpd>([&](auto tic) {
                constexpr int ti = decltype(tic)::value;
                const float *px = x + (b + s + 16 * ti + l15) * LLD + b + s + lq;   // A[i][k] = X22[i][k]
#pragma unroll
                for (int q = 0; q < 4 * (ti + 1); ++q) ax[ti][q] = -px[4 * q];
            });
            f32x4 acc[bpd];
#pragma unroll
            for (int ti = 0; ti < bpd; ++ti) acc[ti] = f32x4{0.f, 0.f, 0.f, 0.f};
            static_for<0, nq>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                static_for<q / 4, bpd>([&](auto tic) {
                    constexpr int ti = decltype(tic)::value;
                    acc[ti] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[ti][q], bt[q], acc[ti], 0, 0, 0);
                });
            });
#pragma unroll
            for (int ti = 0; ti < bpd; ++ti) {
                float *dst = x + (b + s + 16 * ti + 4 * lq) * LLD + b + 16 * mine + l15;
                dst[0] = acc[ti][0]; dst[LLD] = acc[ti][1]; dst[2 * LLD] = acc[ti][2]; dst[3 * LLD] = acc[ti][3];
            }
        }
        __syncthreads();
        LF_ADD(5);
    });
    for (int e = tid; e < NB * (NB / 4); e += 256) {
        const int r = e >> 5, c4 = (e & 31) * 4;
        *reinterpret_cast<f32x4 *>(linv + (o + r) * ld + o + c4) = lds4(x + r * LLD + c4);
    }
#ifdef GQ_LEAF_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    LF_ADD(0);
    if (g_leaf_trace && tid == 0 && o == 0)
        for (int i = 0; i < 8; ++i) g_leaf_trace[i] = lf_acc[i];
#endif
}

#ifdef GQ_LEAF_TRACE
}  // namespace
extern "C" int gq_debug_leaf_trace(unsigned long long *buf)
{
    GQ_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_leaf_trace), &buf, sizeof(buf)));
    return GQ_OK;
}
namespace {
#endif

// (L, L^-1) of w[o:o+n, o:o+n]; n is a multiple of 128.
int cholinv_rec(float *w, float *linv, int64_t ld, int64_t o, int64_t n, int *status, hipStream_t stream)
{
    if (n <= NB) {
        hipLaunchKernelGGL(leaf_kernel, dim3(1), dim3(256), LEAF_LDS, stream, w, linv, ld, o, status);
        GQ_LAUNCH_CHECK();
        return GQ_OK;
    }
    const int64_t n1 = ((n / NB) / 2) * NB, n2 = n - n1;
    int rc = cholinv_rec(w, linv, ld, o, n1, status, stream);
    if (rc) return rc;
    float *a21 = w + (o + n1) * ld + o;          // B, later X = L21 L11^-1
    float *a22 = w + (o + n1) * ld + o + n1;     // D
    float *i11 = linv + o * ld + o;              // L11^-1
    float *i21 = linv + (o + n1) * ld + o;       // scratch for L21, finally (L^-1)21
    float *i22 = linv + (o + n1) * ld + o + n1;  // L22^-1
    GqSgemmArgs g{};
    g.lda = g.ldb = g.ldc = ld;
    // L21 = B * L11^-T          out[i][j] = sum_k B[i][k] * L11inv[j][k],  k <= j
    g.a = a21; g.b = i11; g.c = i21; g.m = (int)n2; g.n = (int)n1; g.k = (int)n1;
    g.layout = GQ_A_MK | GQ_B_NK; g.trim = GQ_TRIM_K_LE_N; g.negate = 0; g.accumulate = 0; g.lower_only = 0;
    if ((rc = gq_sgemm(g, stream))) return rc;
    // D -= L21 * L21^T          lower tiles only
    g.a = i21; g.b = i21; g.c = a22; g.m = (int)n2; g.n = (int)n2; g.k = (int)n1;
    g.layout = GQ_A_MK | GQ_B_NK; g.trim = GQ_TRIM_NONE; g.negate = 1; g.accumulate = 1; g.lower_only = 1;
    if ((rc = gq_sgemm(g, stream))) return rc;
    if ((rc = cholinv_rec(w, linv, ld, o + n1, n2, status, stream))) return rc;
    // X = L21 * L11^-1          out[i][j] = sum_k L21[i][k] * L11inv[k][j],  k >= j   (into B's place)
    g.a = i21; g.b = i11; g.c = a21; g.m = (int)n2; g.n = (int)n1; g.k = (int)n1;
    g.layout = GQ_A_MK | GQ_B_KN; g.trim = GQ_TRIM_K_GE_N; g.negate = 0; g.accumulate = 0; g.lower_only = 0;
    if ((rc = gq_sgemm(g, stream))) return rc;
    // (L^-1)21 = -L22^-1 * X    out[i][j] = -sum_k L22inv[i][k] * X[k][j],  k <= i
    g.a = i22; g.b = a21; g.c = i21; g.m = (int)n2; g.n = (int)n1; g.k = (int)n2;
    g.layout = GQ_A_MK | GQ_B_KN; g.trim = GQ_TRIM_K_LE_M; g.negate = 1; g.accumulate = 0; g.lower_only = 0;
    return gq_sgemm(g, stream);
}

// ---- far updates on the 16-bit matrix core -------------------------------------------------------------
// D -= L L^T for an fp32 panel L (rows x depth) as ONE bf16 product with fp32 accumulation: every entry is split into
// three bf16 pieces, l = l1 + l2 + l3 (each the bf16 rounding of what the pieces before it leave: 24 significant bits in
// all, the sum is l exactly up to its last bit), and the six products of pieces that reach 2^-18 of the largest are
// summed -- l1 l1 + l1 l2 + l2 l1 + l1 l3 + l3 l1 + l2 l2; what is left out is below 2^-26 of a product.  The products
// of bf16 pieces are exact in fp32, so the result carries fp32 rounding only in its additions, like the fp32 product it
// replaces.  Written as x = [l1 | l1 | l2 | l1 | l3 | l2] and w = [l1 | l2 | l1 | l3 | l1 | l2] along k (6 x depth deep)
// it is the k-contiguous "x w^T" of the replay GEMM, whose kernel runs it (gq_gemm_nt_sub_lower_f32) at six times the
// flops and about twelve times the rate of the fp32 matrix instructions: twice as fast where the tiles fill the chip.
__global__ __launch_bounds__(256) void split3_kernel(const float *__restrict__ src, int64_t ld, int64_t rows, int depth,
                                                     unsigned short *__restrict__ xs, unsigned short *__restrict__ wsp)
{
    const int chunks = depth >> 3;                       // 8 entries per thread and turn
    const int64_t total = rows * chunks;
    const int64_t out_ld = (int64_t)6 * depth;
    for (int64_t it = (int64_t)blockIdx.x * 256 + threadIdx.x; it < total; it += (int64_t)gridDim.x * 256) {
        const int64_t r = it / chunks;
        const int k0 = (int)(it - r * chunks) * 8;
        const f32x4 v0 = *reinterpret_cast<const f32x4 *>(src + r * ld + k0);
        const f32x4 v1 = *reinterpret_cast<const f32x4 *>(src + r * ld + k0 + 4);
        unsigned short p[3][8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float a = i < 4 ? v0[i] : v1[i - 4];
            const __bf16 a1 = static_cast<__bf16>(a);
            const float r1 = a - static_cast<float>(a1);          // exact
            const __bf16 a2 = static_cast<__bf16>(r1);
            const float r2 = r1 - static_cast<float>(a2);         // exact
            const __bf16 a3 = static_cast<__bf16>(r2);
            p[0][i] = __builtin_bit_cast(unsigned short, a1);
            p[1][i] = __builtin_bit_cast(unsigned short, a2);
            p[2][i] = __builtin_bit_cast(unsigned short, a3);
        }
        u32x4 q[3];
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) q[t][i] = (unsigned)p[t][2 * i] | ((unsigned)p[t][2 * i + 1] << 16);
        unsigned short *xo = xs + r * out_ld + k0, *wo = wsp + r * out_ld + k0;
        constexpr int XP[6] = {0, 0, 1, 0, 2, 1}, WP[6] = {0, 1, 0, 2, 0, 1};
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            *reinterpret_cast<u32x4 *>(xo + (int64_t)b * depth) = q[XP[b]];
            *reinterpret_cast<u32x4 *>(wo + (int64_t)b * depth) = q[WP[b]];
        }
    }
}

// Whether a far update of `rest` rows and `depth` columns goes through the split product: it runs in rounds of one
// 256 x 256 tile per CU (about 89 us per round at depth 512 plus the split pass), the fp32 kernel at about 110 TFLOP/s.
std::atomic<int> g_invert_split{-1};   // -1: not set yet (GQ_INVERT_SPLIT decides, else automatic)

bool far_update_splits(int64_t rest, int64_t depth, size_t scratch_bytes)
{
    int mode = g_invert_split.load();   // 0 automatic, 1 never, 2 whenever the buffers fit
    if (mode < 0) {
        const char *e = getenv("GQ_INVERT_SPLIT");   // 0: never, anything else: wherever they fit
        mode = !e ? 0 : atoi(e) == 0 ? 1 : 2;
        int expect = -1;
        g_invert_split.compare_exchange_strong(expect, mode);
    }
    if (mode == 1 || depth % 64 || rest % 128) return false;
    if ((size_t)rest * 6 * depth * 2 * 2 > scratch_bytes) return false;
    if (mode == 2) return true;
    const int64_t t = gq_cdiv(rest, 256), tiles = t * (t + 1) / 2;
    const double fp32_us = (double)rest * rest * depth / 110e6;
    const double split_us = (double)gq_cdiv(tiles, 256) * (89.0 * depth / 512.0 + 8.0) + 25.0;
    return split_us < 0.9 * fp32_us;
}

// The same (L, L^-1) with the triangular inverse taken off the factorisation's chain.
//
// Factorisation, right-looking over 128-column panels: leaf (L_jj^-1 of the diagonal block), panel
// L21 = B * L_jj^-T, trailing update D -= L21 L21^T -- applied at once only to the columns of the
// current super-panel of four panels, and to everything beyond it once per super-panel with the four
// panels together (a 512-deep product runs at three times the rate of four 128-deep ones).  The panel
// and update GEMMs span the whole remaining height, so they fill the chip where the recursion's small
// nodes do not.  Afterwards linv holds the L_jj^-1 on its diagonal blocks and the raw entries of L
// below them.  (Measured and dropped: the part of a far update that the next super-panel does not read,
// on a second stream beside that super-panel's chain -- 15.51 -> 15.22 ms at C = 11008: the chain's
// one-workgroup kernels then wait for a CU behind the update's workgroups.)
//
// Inverse, by doubling: inv([L11 0; L21 L22]) = [L11^-1 0; -L22^-1 (L21 L11^-1) L22^-1].  The nodes of
// one level are independent and, for the aligned nodes of 2^l blocks, equally shaped and equally
// spaced: floor(nb / 2^l) of them go into ONE batched launch per product (two per level).  A block
// count that is not a power of two leaves a chain of complete subtrees of decreasing size (the bits
// of nb, left to right); they are joined right to left at the end, one unbatched pair of products
// each.  X = L21 L11^-1 goes into w (its lower part is free once the factorisation is done).
int cholinv_panels(float *w, float *linv, int64_t ld, int64_t n, int *status, hipStream_t stream, void *scratch = nullptr,
                   size_t scratch_bytes = 0)
{
    const int nb = (int)(n / NB);
    int rc;
    GqSgemmArgs g{};
    g.lda = g.ldb = g.ldc = ld;
    static const int SP = [] {   // panels per super-panel: updates beyond it wait and are applied SP * 128 deep
        const char *e = getenv("GQ_INVERT_SP");
        return e && atoi(e) > 0 ? atoi(e) : 4;
    }();
    for (int j = 0; j < nb; ++j) {
        const int64_t o = (int64_t)j * NB, rest = n - o - NB;
        hipLaunchKernelGGL(leaf_kernel, dim3(1), dim3(256), LEAF_LDS, stream, w, linv, ld, o, status);
        GQ_LAUNCH_CHECK();
        if (rest == 0) break;
        const int sp_end = std::min(nb, (j / SP + 1) * SP);   // first panel of the next super-panel
        float *b21 = w + (o + NB) * ld + o;          // B
        float *ijj = linv + o * ld + o;              // L_jj^-1
        float *l21 = linv + (o + NB) * ld + o;       // L21 (stays: the inverse pass reads it)
        g.batch = 0;
        g.a = b21; g.b = ijj; g.c = l21; g.m = (int)rest; g.n = NB; g.k = NB;
        g.layout = GQ_A_MK | GQ_B_NK; g.trim = GQ_TRIM_K_LE_N; g.negate = 0; g.accumulate = 0; g.lower_only = 0;
        if ((rc = gq_sgemm(g, stream))) return rc;
        // now: the columns of this super-panel that are still to come, all rows below (the tiles above the
        // diagonal of that strip are computed too; nobody reads them)
        const int64_t near_cols = (int64_t)(sp_end - (j + 1)) * NB;
        if (near_cols > 0) {
            g.a = l21; g.b = l21; g.c = w + (o + NB) * ld + o + NB; g.m = (int)rest; g.n = (int)near_cols; g.k = NB;
            g.layout = GQ_A_MK | GQ_B_NK; g.trim = GQ_TRIM_NONE; g.negate = 1; g.accumulate = 1; g.lower_only = 0;
            if ((rc = gq_sgemm(g, stream))) return rc;
        }
        // at the end of a super-panel: everything beyond it, with all of its panels at once
        if (j + 1 == sp_end && sp_end < nb) {
            const int64_t s0 = (int64_t)((j / SP) * SP) * NB, e0 = (int64_t)sp_end * NB;
            float *pfar = linv + e0 * ld + s0;       // L[e0.., s0..e0)
            const int64_t rest2 = n - e0, depth = e0 - s0;
            if (scratch && far_update_splits(rest2, depth, scratch_bytes)) {
                unsigned short *xs = (unsigned short *)scratch, *wsp = xs + rest2 * 6 * depth;
                const int64_t items = rest2 * (depth / 8);
                hipLaunchKernelGGL(split3_kernel, dim3((unsigned)std::min<int64_t>(gq_cdiv(items, 256), 256 * 16)), dim3(256), 0, stream,
                                   pfar, ld, rest2, (int)depth, xs, wsp);
                GQ_LAUNCH_CHECK();
                if ((rc = gq_gemm_nt_sub_lower_f32(w + e0 * ld + e0, ld, xs, wsp, rest2, 6 * depth, stream))) return rc;
            } else {
                g.a = pfar; g.b = pfar; g.c = w + e0 * ld + e0; g.m = (int)rest2; g.n = (int)rest2; g.k = (int)depth;
                g.layout = GQ_A_MK | GQ_B_NK; g.trim = GQ_TRIM_NONE; g.negate = 1; g.accumulate = 1; g.lower_only = 1;
                if ((rc = gq_sgemm(g, stream))) return rc;
            }
        }
    }
    // node at block offset ob with a complete left part of n1 blocks and a complete right part of n2
    // blocks, `batch` of them `spacing` blocks apart
    auto join = [&](int64_t ob, int64_t n1, int64_t n2, int batch, int64_t spacing) -> int {
        const int64_t o = ob * NB, h1 = n1 * NB, h2 = n2 * NB;
        float *x21 = w + (o + h1) * ld + o;
        float *i11 = linv + o * ld + o;
        float *i21 = linv + (o + h1) * ld + o;
        float *i22 = linv + (o + h1) * ld + o + h1;
        g.batch = batch;
        g.stride_a = g.stride_b = g.stride_c = spacing * NB * (ld + 1);
        // X = L21 * L11^-1          out[i][j] = sum_k L21[i][k] * L11inv[k][j],  k >= j
        g.a = i21; g.b = i11; g.c = x21; g.m = (int)h2; g.n = (int)h1; g.k = (int)h1;
        g.layout = GQ_A_MK | GQ_B_KN; g.trim = GQ_TRIM_K_GE_N; g.negate = 0; g.accumulate = 0; g.lower_only = 0;
        int r = gq_sgemm(g, stream);
        if (r) return r;
        // (L^-1)21 = -L22^-1 * X    out[i][j] = -sum_k L22inv[i][k] * X[k][j],  k <= i
        g.a = i22; g.b = x21; g.c = i21; g.m = (int)h2; g.n = (int)h1; g.k = (int)h2;
        g.layout = GQ_A_MK | GQ_B_KN; g.trim = GQ_TRIM_K_LE_M; g.negate = 1; g.accumulate = 0; g.lower_only = 0;
        return gq_sgemm(g, stream);
    };
    for (int64_t half = 1; 2 * half <= nb; half *= 2)   // aligned complete nodes of 2 * half blocks
        if ((rc = join(0, half, half, (int)(nb / (2 * half)), 2 * half))) return rc;
    // the complete subtrees are the bits of nb, largest first; join them from the right
    int64_t right_o = -1, right_n = 0;
    for (int64_t bit = 1, end = nb; bit <= nb; bit *= 2) {
        if (!(nb & bit)) continue;
        const int64_t o = end - bit;   // this subtree covers blocks [end - bit, end)
        if (right_o >= 0 && (rc = join(o, bit, right_n, 1, 0))) return rc;
        right_o = o;
        right_n += bit;
        end = o;
    }
    return GQ_OK;
}

struct InvertWorkspace {
    float *t;        // c * c     gather target
    float *w;        // cp * cp   reversed work matrix
    float *linv;     // cp * cp
    float *diag;     // c
    float *damp;     // 1
    int *status;     // 1: a pivot of the current try was not a positive finite number
    int *fatal;      // 1: the diagonal is not finite (set once, before the first try)
    int32_t *perm;   // c (used when the caller passes no perm buffer)
    int32_t *perm_inv;
};

size_t carve(InvertWorkspace &ws, char *base, int64_t c)
{
    const int64_t cp = gq_round_up(c, NB);
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char *p = base ? base + off : nullptr;
        off += (bytes + 255) & ~(size_t)255;
        return p;
    };
    ws.t = (float *)take((size_t)c * c * 4);
    ws.w = (float *)take((size_t)cp * cp * 4);
    ws.linv = (float *)take((size_t)cp * cp * 4);
    ws.diag = (float *)take((size_t)c * 4);
    ws.damp = (float *)take(256);
    ws.status = (int *)take(256);
    ws.fatal = (int *)take(256);
    ws.perm = (int32_t *)take((size_t)c * 4);
    ws.perm_inv = (int32_t *)take((size_t)c * 4);
    return off;
}

int ew_blocks(int64_t total) { return (int)std::min<int64_t>(gq_cdiv(total, 256), 256 * 16); }

}  // namespace

extern "C" int gq_set_invert_split(int mode)
{
    GQ_REQUIRE(mode >= 0 && mode <= 2, "gq_set_invert_split: 0 (automatic), 1 (fp32 products) or 2 (split products wherever they fit), got %d", mode);
    const int before = g_invert_split.exchange(mode);
    return before < 0 ? 0 : before;
}

extern "C" size_t gq_hessian_invert_workspace(int64_t c)
{
    if (c <= 0) return 0;
    InvertWorkspace ws;
    return carve(ws, nullptr, c);
}

namespace {

// argument checks shared by the entry points; carves the workspace
int invert_setup(InvertWorkspace &ws, float *hessian, float *hessian_inv, int64_t c, void *workspace, size_t workspace_bytes,
                 const char *who)
{
    GQ_REQUIRE(hessian && hessian_inv && workspace, "%s: NULL pointer", who);
    GQ_REQUIRE(c > 0 && c <= 32767 * 4, "%s: bad size %lld", who, (long long)c);
    GQ_REQUIRE((((uintptr_t)hessian | (uintptr_t)hessian_inv | (uintptr_t)workspace) & 15) == 0,
               "%s: pointers must be 16-byte aligned", who);
    const size_t need = carve(ws, (char *)workspace, c);
    if (workspace_bytes < need) {
        gq_set_error("%s: workspace of %zu bytes, %zu needed", who, workspace_bytes, need);
        return GQ_EWORKSPACE;
    }
    return GQ_OK;
}

// scale by 2 / n, dead-diagonal fix, damping value, act-order permutation of H: everything before the tries
// (`*gathered`: where the first try finds the matrix when it is not in `hessian` yet, else NULL)
int invert_prepare(const InvertWorkspace &ws, float *hessian, int32_t *perm, int32_t *perm_inv, int64_t c, double n_samples,
                   float damp_ratio, int act_order, hipStream_t stream, const char *who, const float **gathered)
{
    *gathered = nullptr;
    GQ_REQUIRE(n_samples > 0, "%s: n_samples must be positive", who);
    GQ_REQUIRE(!act_order || (perm && perm_inv), "%s: act_order needs perm and perm_inv buffers", who);
    const int64_t cc = c * c;
    const float scale = (float)(2.0 / n_samples);
    if (act_order)   // the scaling rides on the gather: one pass over H less
        hipLaunchKernelGGL(scaled_diag_kernel, dim3((unsigned)gq_cdiv(c, 256)), dim3(256), 0, stream, hessian, c, scale, ws.diag);
    else
        hipLaunchKernelGGL(scale_kernel, dim3(ew_blocks(cc)), dim3(256), 0, stream, hessian, c, scale, ws.diag);
    GQ_LAUNCH_CHECK();
    hipLaunchKernelGGL(damp_kernel, dim3(1), dim3(256), 0, stream, ws.diag, c, damp_ratio, ws.damp, ws.fatal);
    GQ_LAUNCH_CHECK();
    if (act_order) {
        GQ_HIP(hipMemsetAsync(perm_inv, 0, (size_t)c * sizeof(int32_t), stream));
        hipLaunchKernelGGL(rank_sort_kernel, dim3((unsigned)gq_cdiv(c, 64), RANK_SPLIT), dim3(256), 0, stream, ws.diag, (int)c, perm_inv);
        GQ_LAUNCH_CHECK();
        hipLaunchKernelGGL(perm_scatter_kernel, dim3((unsigned)gq_cdiv(c, 256)), dim3(256), 0, stream, perm_inv, (int)c, perm);
        GQ_LAUNCH_CHECK();
        if (c <= GATHER_LDS_MAX_C) {
            const bool vec = c % 4 == 0 && ((uintptr_t)hessian & 15) == 0 && ((uintptr_t)ws.t & 15) == 0 && ((uintptr_t)perm & 15) == 0;
            const size_t lds = (size_t)c * 4;
            const void *fn = vec ? (const void *)gather_rows_kernel<4> : (const void *)gather_rows_kernel<1>;
            GQ_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            const unsigned grid = (unsigned)std::min<int64_t>(c, 2048);
            if (vec) hipLaunchKernelGGL(gather_rows_kernel<4>, dim3(grid), dim3(256), lds, stream, hessian, perm, c, scale, ws.t);
            else hipLaunchKernelGGL(gather_rows_kernel<1>, dim3(grid), dim3(256), lds, stream, hessian, perm, c, scale, ws.t);
        } else {
            hipLaunchKernelGGL(gather_kernel, dim3(ew_blocks(cc)), dim3(256), 0, stream, hessian, perm, c, scale, ws.t);
        }
        GQ_LAUNCH_CHECK();
        *gathered = ws.t;
    }
    return GQ_OK;
}

// One try, enqueued: damp the diagonal once more (in place, as the reference's loop does), factor and
// invert a copy; the factor is flipped into hessian_inv on the device when the try succeeded
// (flip_out_kernel looks at the status word itself).
int invert_try(const InvertWorkspace &ws, float *hessian, const float *gathered, float *hessian_inv, int64_t c, hipStream_t stream)
{
    const int64_t cp = gq_round_up(c, NB);
    GQ_HIP(hipFuncSetAttribute((const void *)leaf_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LEAF_LDS));
    GQ_HIP(hipMemsetAsync(ws.status, 0, 256, stream));
    hipLaunchKernelGGL(damp_flip_kernel, dim3(ew_blocks(cp * cp)), dim3(256), 0, stream, hessian, gathered, c, cp, ws.damp, ws.w);
    GQ_LAUNCH_CHECK();
    static const bool recursive = [] {   // experiments: GQ_INVERT=rec selects the recursive form
        const char *e = getenv("GQ_INVERT");
        return e && e[0] == 'r';
    }();
    // linv is not cleared for the panel form: the leaves write their diagonal blocks whole, every block below
    // them is written before it is read, and nothing -- the products' K trims, the flip into U -- reads above
    // the block diagonal (checked by running the GPU tests with linv filled with NaN)
    if (recursive) GQ_HIP(hipMemsetAsync(ws.linv, 0, (size_t)cp * cp * 4, stream));
    int rc = recursive ? cholinv_rec(ws.w, ws.linv, cp, 0, cp, ws.status, stream)
                       : cholinv_panels(ws.w, ws.linv, cp, cp, ws.status, stream, ws.t, (size_t)c * c * 4);   // the gather target is free once damp_flip_kernel has read it
    if (rc) return rc;
    hipLaunchKernelGGL(flip_out_kernel, dim3(ew_blocks(c * c)), dim3(256), 0, stream, ws.linv, c, cp, hessian_inv, ws.status);
    GQ_LAUNCH_CHECK();
    return GQ_OK;
}

// The tries after `tries` already made, one host round trip each, starting from the verdict of the last one
// (verdict[0]: a pivot was not positive, verdict[1]: the diagonal of H is not finite).
int invert_retry(const InvertWorkspace &ws, float *hessian, float *hessian_inv, int64_t c, int max_try, int tries,
                 int verdict[2], int *tries_host, hipStream_t stream, const char *who, const float *gathered = nullptr)
{
    for (;;) {
        if (verdict[1]) {
            if (tries_host) *tries_host = tries;
            gq_set_error("%s: the Hessian's diagonal is not finite (overflowed activations?)", who);
            return GQ_ENOTPD;
        }
        if (verdict[0] == 0) {
            if (tries_host) *tries_host = tries;
            return GQ_OK;
        }
        if (tries >= max_try) break;
        ++tries;
        const int rc = invert_try(ws, hessian, gathered, hessian_inv, c, stream);
        if (rc) return rc;
        gathered = nullptr;   // `hessian` holds the matrix from the first try on
        GQ_HIP(hipMemcpyAsync(&verdict[0], ws.status, sizeof(int), hipMemcpyDeviceToHost, stream));
        GQ_HIP(hipMemcpyAsync(&verdict[1], ws.fatal, sizeof(int), hipMemcpyDeviceToHost, stream));
        GQ_HIP(hipStreamSynchronize(stream));
    }
    if (tries_host) *tries_host = tries;
    gq_set_error("%s: not positive definite after %d dampings", who, tries);
    return GQ_ENOTPD;
}

int invert_impl(float *hessian, float *hessian_inv, int32_t *perm, int32_t *perm_inv, int64_t c, double n_samples,
                float damp_ratio, int act_order, int max_try, void *workspace, size_t workspace_bytes, int *tries_host,
                hipStream_t stream, const char *who)
{
    GQ_REQUIRE(max_try >= 1, "%s: max_try must be >= 1", who);
    InvertWorkspace ws;
    int rc = invert_setup(ws, hessian, hessian_inv, c, workspace, workspace_bytes, who);
    if (rc) return rc;
    const float *gathered = nullptr;
    rc = invert_prepare(ws, hessian, perm, perm_inv, c, n_samples, damp_ratio, act_order, stream, who, &gathered);
    if (rc) return rc;
    int verdict[2] = {1, 0};   // "not yet factored": the first pass of the loop makes try 1
    return invert_retry(ws, hessian, hessian_inv, c, max_try, 0, verdict, tries_host, stream, who, gathered);
}

}  // namespace

extern "C" int gq_hessian_invert(float *hessian, float *hessian_inv, int32_t *perm, int32_t *perm_inv, int64_t c,
                                 double n_samples, float damp_ratio, int act_order, int max_try, void *workspace,
                                 size_t workspace_bytes, int *tries_host, gq_stream_t stream_)
{
    return invert_impl(hessian, hessian_inv, perm, perm_inv, c, n_samples, damp_ratio, act_order, max_try, workspace,
                       workspace_bytes, tries_host, (hipStream_t)stream_, "gq_hessian_invert");
}

// ---- the same call in two halves, without a host round trip in the first ------------------------
// gq_hessian_invert_begin enqueues everything up to and including the first try and leaves its verdict
// in device memory; the caller goes on enqueueing the work that consumes hessian_inv and reads the
// verdict when it next has to wait for the device anyway.  gq_hessian_invert_finish takes that verdict:
// nothing to do when the first try succeeded (the usual case: the damping makes H positive definite),
// otherwise the remaining tries, synchronously -- what was computed from hessian_inv in between has
// then to be computed again.
extern "C" int gq_hessian_invert_begin(float *hessian, float *hessian_inv, int32_t *perm, int32_t *perm_inv, int64_t c,
                                       double n_samples, float damp_ratio, int act_order, void *workspace,
                                       size_t workspace_bytes, int32_t *verdict_dev, gq_stream_t stream_)
{
    const char *who = "gq_hessian_invert_begin";
    hipStream_t stream = (hipStream_t)stream_;
    GQ_REQUIRE(verdict_dev, "%s: NULL verdict", who);
    InvertWorkspace ws;
    int rc = invert_setup(ws, hessian, hessian_inv, c, workspace, workspace_bytes, who);
    if (rc) return rc;
    const float *gathered = nullptr;
    rc = invert_prepare(ws, hessian, perm, perm_inv, c, n_samples, damp_ratio, act_order, stream, who, &gathered);
    if (rc) return rc;
    rc = invert_try(ws, hessian, gathered, hessian_inv, c, stream);
    if (rc) return rc;
    GQ_HIP(hipMemcpyAsync(verdict_dev, ws.status, sizeof(int), hipMemcpyDeviceToDevice, stream));
    GQ_HIP(hipMemcpyAsync(verdict_dev + 1, ws.fatal, sizeof(int), hipMemcpyDeviceToDevice, stream));
    return GQ_OK;
}

extern "C" int gq_hessian_invert_finish(float *hessian, float *hessian_inv, int64_t c, int max_try, void *workspace,
                                        size_t workspace_bytes, const int32_t *verdict_host, int *tries_host,
                                        gq_stream_t stream_)
{
    const char *who = "gq_hessian_invert_finish";
    GQ_REQUIRE(verdict_host, "%s: NULL verdict", who);
    GQ_REQUIRE(max_try >= 1, "%s: max_try must be >= 1", who);
    InvertWorkspace ws;
    const int rc = invert_setup(ws, hessian, hessian_inv, c, workspace, workspace_bytes, who);
    if (rc) return rc;
    int verdict[2] = {verdict_host[0], verdict_host[1]};
    return invert_retry(ws, hessian, hessian_inv, c, max_try, 1, verdict, tries_host, (hipStream_t)stream_, who);
}

// ---- the same call without any allocation -----------------------------------------------------
// For callers that cannot allocate device memory (harnesses that forbid it, tight loops): the
// workspace gq_hessian_invert takes from its caller lives in a file-scope device array instead.
// Each GPU the library runs on holds its own copy of the array, at its own address, until the
// process ends; host code never uses the array's name as an address, only the pointer that
// hipGetSymbolAddress returned for the device that is current.
namespace {

constexpr int64_t STATIC_MAX_C = 1024;
// carve() for C = 1024: three C x C fp32 matrices, three C-long vectors, three 256-byte words
constexpr size_t STATIC_SCRATCH_BYTES = 3 * (size_t)STATIC_MAX_C * STATIC_MAX_C * 4 + 3 * (size_t)STATIC_MAX_C * 4 + 3 * 256;
static __device__ __align__(256) unsigned char scratch[STATIC_SCRATCH_BYTES];

constexpr int MAX_DEVICES = 64;
std::mutex g_scratch_mu;
void *g_scratch_ptr[MAX_DEVICES] = {};

// the device-side address of `scratch` on the current device (looked up once per device)
int scratch_for_current_device(void **out)
{
    int device = 0;
    GQ_HIP(hipGetDevice(&device));
    GQ_REQUIRE(device >= 0 && device < MAX_DEVICES, "gq_hessian_invert_static: device index %d out of range", device);
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[device]) {
        void *p = nullptr;
        GQ_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        GQ_REQUIRE(p != nullptr, "gq_hessian_invert_static: no device address for the scratch array");
        g_scratch_ptr[device] = p;
    }
    *out = g_scratch_ptr[device];
    return GQ_OK;
}

}  // namespace

extern "C" int64_t gq_hessian_invert_static_max_c(void) { return STATIC_MAX_C; }

extern "C" int gq_hessian_invert_static(float *hessian, float *hessian_inv, int32_t *perm, int32_t *perm_inv, int64_t c,
                                        double n_samples, float damp_ratio, int act_order, int max_try,
                                        int *tries_host, gq_stream_t stream_)
{
    GQ_REQUIRE(c > 0 && c <= STATIC_MAX_C, "gq_hessian_invert_static: C = %lld, at most %lld fit the built-in workspace "
               "(use gq_hessian_invert with a workspace of your own)", (long long)c, (long long)STATIC_MAX_C);
    void *ws = nullptr;
    int rc = scratch_for_current_device(&ws);
    if (rc) return rc;
    return invert_impl(hessian, hessian_inv, perm, perm_inv, c, n_samples, damp_ratio, act_order, max_try, ws,
                       STATIC_SCRATCH_BYTES, tries_host, (hipStream_t)stream_, "gq_hessian_invert_static");
}
