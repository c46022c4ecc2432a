// fp32 GEMM for gfx950 on v_mfma_f32_32x32x2_f32 (exact fp32: every output element is one
// k-ascending fma chain).  128x128 (or 64x64) tile per 256-thread workgroup, 2x2 waves,
// 32-deep k-steps staged global -> registers -> LDS (two LDS stages, one barrier per step; the
// global loads of step s + 2 are issued before the MFMAs of step s, and the registers of step s + 1
// are written to LDS during them: a load has a step and a half to arrive, which matters when a
// small product leaves one workgroup alone on its CU).
//
// LDS images, chosen by how the operand lies in memory so that global loads are 16-byte and
// coalesced and fragment reads (one ds_read_b32 per lane per MFMA operand) are conflict-free:
//   k-contiguous source (A_MK, B_NK): [128 rows][33 floats]   read at row*33 + k
//   row-contiguous source (A_KM, B_KN): [32 k][128 floats]    read at k*128 + row
#include <type_traits>

#include <algorithm>
#include <cstdlib>

#include "sgemm.h"

namespace {

constexpr int BKS = 32;   // k-step

struct Prefetch {
    f32x4 v[4];
};

// BT x BT output tile per 256-thread workgroup (BT = 128: 2x2 waves of 64x64; BT = 64: 2x2 waves of
// 32x32, used when the 128-tile grid would leave most CUs idle).  The result does not depend on BT:
// every element is the same k-ascending chain.
#ifdef GQ_SG_TRACE
__device__ unsigned long long *g_sg_trace = nullptr;
#endif

template <int LAYOUT, int BT>
__global__ __launch_bounds__(256) void sgemm_kernel(GqSgemmArgs g, int tiles_m, int tiles_n)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
#ifdef GQ_SG_TRACE
    const unsigned long long t_begin = __builtin_readcyclecounter();
    const unsigned long long r_begin = __builtin_amdgcn_s_memrealtime();
#endif
    float *smem = reinterpret_cast<float *>(smem_raw);
    constexpr bool A_KM = (LAYOUT & GQ_A_KM) != 0;
    constexpr bool B_KN = (LAYOUT & GQ_B_KN) != 0;
    constexpr int WB = BT / 64;                     // MFMA blocks per wave and dimension
    constexpr int IMG_FLOATS = BT * 33;             // the larger of the two images
    constexpr int STAGE_FLOATS = 2 * IMG_FLOATS;    // A + B

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    g.a += (int64_t)blockIdx.y * g.stride_a;
    g.b += (int64_t)blockIdx.y * g.stride_b;
    g.c += (int64_t)blockIdx.y * g.stride_c;
    int tm, tn;
    if (g.lower_only) {
        const int bid = blockIdx.x;
        int r = (int)((sqrtf(8.0f * (float)bid + 1.0f) - 1.0f) * 0.5f);
        while ((r + 1) * (r + 2) / 2 <= bid) ++r;
        while (r * (r + 1) / 2 > bid) --r;
        tm = r;
        tn = bid - r * (r + 1) / 2;
    } else if (g.trim == GQ_TRIM_K_LE_N) {
        // tiles differ in the length of their K range: walk them longest first, so that the last
        // round of workgroups is made of the short ones
        tn = tiles_n - 1 - (int)(blockIdx.x / tiles_m);
        tm = blockIdx.x % tiles_m;
    } else if (g.trim == GQ_TRIM_K_GE_N) {
        tn = blockIdx.x / tiles_m;
        tm = blockIdx.x % tiles_m;
    } else if (g.trim == GQ_TRIM_K_LE_M) {
        tm = tiles_m - 1 - (int)(blockIdx.x / tiles_n);
        tn = blockIdx.x % tiles_n;
    } else {
        tm = blockIdx.x / tiles_n;
        tn = blockIdx.x % tiles_n;
    }
    const int m0 = tm * BT, n0 = tn * BT;

    // the part of K this tile needs (one operand is triangular); bounds rounded to the tile edge
    int k_lo = 0, k_hi = g.k;
    if (g.trim == GQ_TRIM_K_LE_N) k_hi = min(g.k, (tn + 1) * BT);
    else if (g.trim == GQ_TRIM_K_GE_N) k_lo = min(g.k, tn * BT);
    else if (g.trim == GQ_TRIM_K_LE_M) k_hi = min(g.k, (tm + 1) * BT);

    const int wr = wave >> 1, wc = wave & 1;
    const int li = lane & 31, lh = lane >> 5;

    f32x16 acc[WB][WB];
#pragma unroll
    for (int a = 0; a < WB; ++a)
#pragma unroll
        for (int b = 0; b < WB; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = 0.0f;
                if (g.accumulate) {
                    const int row = m0 + (wr * WB + a) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    const int col = n0 + (wc * WB + b) * 32 + li;
                    if (row < g.m && col < g.n) v = g.c[(int64_t)row * g.ldc + col];
                }
                acc[a][b][e] = v;
            }

    const float sgn = g.negate ? -1.0f : 1.0f;
    const int n_steps = (k_hi - k_lo + BKS - 1) / BKS;
    constexpr int NP = BT / 32;   // 16-byte pieces per thread and operand
    // Per-thread source pointers of the pieces, set up once and advanced by a constant per step
    // (the per-step address arithmetic otherwise delays the first MFMAs of every step); a piece
    // whose tile row lies outside the matrix is never loaded.
    const float *pa[NP], *pb[NP];
    bool oka[NP], okb[NP];
    int ka[NP], kb[NP];   // first k of the piece at step 0 (for the ragged last step)
#pragma unroll
    for (int pass = 0; pass < NP; ++pass) {
        if constexpr (A_KM) {
            constexpr int TPR = BT / 4, KPP = 256 / TPR;
            const int row = m0 + (tid % TPR) * 4, k = k_lo + tid / TPR + KPP * pass;
            oka[pass] = row < g.m; ka[pass] = k;
            pa[pass] = g.a + (int64_t)k * g.lda + row;
        } else {
            const int row = m0 + (tid >> 3) + 32 * pass, k = k_lo + (tid & 7) * 4;
            oka[pass] = row < g.m; ka[pass] = k;
            pa[pass] = g.a + (int64_t)row * g.lda + k;
        }
        if constexpr (B_KN) {
            constexpr int TPR = BT / 4, KPP = 256 / TPR;
            const int row = n0 + (tid % TPR) * 4, k = k_lo + tid / TPR + KPP * pass;
            okb[pass] = row < g.n; kb[pass] = k;
            pb[pass] = g.b + (int64_t)k * g.ldb + row;
        } else {
            const int row = n0 + (tid >> 3) + 32 * pass, k = k_lo + (tid & 7) * 4;
            okb[pass] = row < g.n; kb[pass] = k;
            pb[pass] = g.b + (int64_t)row * g.ldb + k;
        }
    }
    const int64_t inc_a = A_KM ? (int64_t)BKS * g.lda : BKS, inc_b = B_KN ? (int64_t)BKS * g.ldb : BKS;
    // A 64-tile that lies inside both matrices and whose K range is a whole number of steps (every
    // small product of the Hessian inversion, nearly every one of the column loop) takes a copy of
    // the pipeline whose steady state has no branch at all: behind a branch -- per load, or around a
    // step's loads -- hipcc no longer knows which loads are in flight and waits for all of them, and
    // the two-step prefetch collapses into a memory round trip per step.
    const bool full_tile = BT == 64 && m0 + BT <= g.m && n0 + BT <= g.n && (k_hi - k_lo) % BKS == 0;
#ifdef GQ_SG_TRACE
    unsigned long long t_loop = 0;
#endif
    auto pipeline = [&](auto full_c) {
        constexpr bool FULL = decltype(full_c)::value;
        // 64-tiles keep the registers of two steps in flight (indexed by step parity); 128-tiles, which
        // run two workgroups per CU and have no registers to spare, one
        constexpr bool DEEP = BT == 64;
        Prefetch qa[2], qb[2];
        auto gload = [&](int step, Prefetch &da, Prefetch &db) {
            const int koff = step * BKS;
    #pragma unroll
            for (int pass = 0; pass < NP; ++pass) {
                if constexpr (FULL) {
                    da.v[pass] = *reinterpret_cast<const f32x4 *>(pa[pass]);
                    db.v[pass] = *reinterpret_cast<const f32x4 *>(pb[pass]);
                } else {
                    f32x4 va = {0.f, 0.f, 0.f, 0.f}, vb = {0.f, 0.f, 0.f, 0.f};
                    if (oka[pass] && ka[pass] + koff < k_hi) va = *reinterpret_cast<const f32x4 *>(pa[pass]);
                    if (okb[pass] && kb[pass] + koff < k_hi) vb = *reinterpret_cast<const f32x4 *>(pb[pass]);
                    da.v[pass] = va;
                    db.v[pass] = vb;
                }
                pa[pass] += inc_a;
                pb[pass] += inc_b;
            }
        };
        // piece `pass` of operand A or B into the LDS image of `stage`
        auto lstore_a = [&](int stage, int pass, const Prefetch &q) {
            float *ia = smem + stage * STAGE_FLOATS;
            if constexpr (A_KM) {
                constexpr int TPR = BT / 4, KPP = 256 / TPR;
                *reinterpret_cast<f32x4 *>(ia + (tid / TPR + KPP * pass) * BT + (tid % TPR) * 4) = q.v[pass];
            } else {
                float *d = ia + ((tid >> 3) + 32 * pass) * 33 + (tid & 7) * 4;
                d[0] = q.v[pass][0]; d[1] = q.v[pass][1]; d[2] = q.v[pass][2]; d[3] = q.v[pass][3];
            }
        };
        auto lstore_b = [&](int stage, int pass, const Prefetch &q) {
            float *ib = smem + stage * STAGE_FLOATS + IMG_FLOATS;
            if constexpr (B_KN) {
                constexpr int TPR = BT / 4, KPP = 256 / TPR;
                *reinterpret_cast<f32x4 *>(ib + (tid / TPR + KPP * pass) * BT + (tid % TPR) * 4) = q.v[pass];
            } else {
                float *d = ib + ((tid >> 3) + 32 * pass) * 33 + (tid & 7) * 4;
                d[0] = q.v[pass][0]; d[1] = q.v[pass][1]; d[2] = q.v[pass][2]; d[3] = q.v[pass][3];
            }
        };

        if (n_steps > 0) {
            gload(0, qa[0], qb[0]);
            if (DEEP && n_steps > 1) gload(1, qa[1], qb[1]);
    #pragma unroll
            for (int pass = 0; pass < NP; ++pass) {
                lstore_a(0, pass, qa[0]);
                lstore_b(0, pass, qb[0]);
            }
        }
        __syncthreads();
    #ifdef GQ_SG_TRACE
        t_loop = __builtin_readcyclecounter();
    #endif
        // one k-step; P = parity of `step`, the register set that held this step's data and is free again
        // STEADY: two more steps are known to follow, nothing in the step is conditional
        auto k_step = [&](int step, auto par, auto steady_c) {
            constexpr int P = decltype(par)::value;
            constexpr bool STEADY = decltype(steady_c)::value;
            const int cur = P;
            const bool more = STEADY || step + 1 < n_steps;
            if constexpr (DEEP) {
                if (STEADY || step + 2 < n_steps) gload(step + 2, qa[P], qb[P]);
            } else {
                if (more) gload(step + 1, qa[P ^ 1], qb[P ^ 1]);
            }
            const float *ia = smem + cur * STAGE_FLOATS, *ib = ia + IMG_FLOATS;
            if constexpr (DEEP) {
                // one wave per SIMD: nobody hides the latency of the fragment reads, so all 32 of a step
                // are issued before its first MFMA
                float fa[BKS / 2], fb[BKS / 2];
                const int ra = wr * 32 + li, rb = wc * 32 + li;
    #pragma unroll
                for (int s = 0; s < BKS / 2; ++s) {
                    const int k = 2 * s + lh;
                    fa[s] = A_KM ? ia[k * BT + ra] : ia[ra * 33 + k];
                    fb[s] = B_KN ? ib[k * BT + rb] : ib[rb * 33 + k];
                }
    #pragma unroll
                for (int s = 0; s < BKS / 2; ++s) {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(sgn * fa[s], fb[s], acc[0][0], 0, 0, 0);
                    if (more && s >= BKS / 2 - 2 * NP) {
                        const int q = s - (BKS / 2 - 2 * NP);
                        if (q & 1) lstore_b(cur ^ 1, q >> 1, qb[P ^ 1]);
                        else lstore_a(cur ^ 1, q >> 1, qa[P ^ 1]);
                    }
                }
            } else {
    #pragma unroll
                for (int s = 0; s < BKS / 2; ++s) {
                    const int k = 2 * s + lh;
                    float fa[WB], fb[WB];
    #pragma unroll
                    for (int u = 0; u < WB; ++u) {
                        const int ra = (wr * WB + u) * 32 + li, rb = (wc * WB + u) * 32 + li;
                        fa[u] = sgn * (A_KM ? ia[k * BT + ra] : ia[ra * 33 + k]);
                        fb[u] = B_KN ? ib[k * BT + rb] : ib[rb * 33 + k];
                    }
    #pragma unroll
                    for (int a = 0; a < WB; ++a)
    #pragma unroll
                        for (int b = 0; b < WB; ++b)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a], fb[b], acc[a][b], 0, 0, 0);
                    // the next step's pieces go to the other LDS stage one at a time in the second half of
                    // the step, not in a burst before the barrier
                    if (more && s >= BKS / 2 - 2 * NP) {
                        const int q = s - (BKS / 2 - 2 * NP);
                        if (q & 1) lstore_b(cur ^ 1, q >> 1, qb[P ^ 1]);
                        else lstore_a(cur ^ 1, q >> 1, qa[P ^ 1]);
                    }
                }
            }
            __syncthreads();
        };
        using P0 = std::integral_constant<int, 0>;
        using P1 = std::integral_constant<int, 1>;
        int step = 0;
        if constexpr (FULL)   // the branch-free steady state: hipcc counts the waits of its loads exactly
            for (; step + 3 < n_steps; step += 2) {
                k_step(step, P0{}, std::true_type{});
                k_step(step + 1, P1{}, std::true_type{});
            }
        for (; step < n_steps; step += 2) {
            k_step(step, P0{}, std::false_type{});
            if (step + 1 < n_steps) k_step(step + 1, P1{}, std::false_type{});
        }
    };
    if constexpr (BT == 64) {
        if (full_tile) pipeline(std::true_type{});
        else pipeline(std::false_type{});
    } else {
        pipeline(std::false_type{});
    }
#ifdef GQ_SG_TRACE
    const unsigned long long t_end = __builtin_readcyclecounter();
#endif

#pragma unroll
    for (int a = 0; a < WB; ++a)
#pragma unroll
        for (int b = 0; b < WB; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + (wr * WB + a) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                const int col = n0 + (wc * WB + b) * 32 + li;
                if (row < g.m && col < g.n) g.c[(int64_t)row * g.ldc + col] = acc[a][b][e];
            }
#ifdef GQ_SG_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (g_sg_trace && blockIdx.x == 300 && lane == 0) {
        unsigned long long *o = g_sg_trace + wave * 8;
        o[0] = t_begin; o[1] = t_loop; o[2] = t_end; o[3] = __builtin_readcyclecounter(); o[4] = r_begin;
        o[5] = __builtin_amdgcn_s_memrealtime(); o[6] = n_steps;
    }
#endif
}


// ---- 128x128 tiles that lie whole inside both operands, K range a whole number of 32-deep steps ----
// The same product (every element one k-ascending fma chain on v_mfma_f32_32x32x2_f32, lanes 0-31
// carrying k = 2s and lanes 32-63 k = 2s + 1 of sub-step s), with the schedule written by hand:
//   * both operands go global -> LDS by LDS-DMA (16 bytes per lane, 1 KiB per wave instruction, no
//     register staging): the eight pieces a wave owns of step t+1 are issued inside the first quarter
//     of step t and waited for (vmcnt(0)) in front of the step's one barrier;
//   * fragment reads are inline-asm ds_read_b32 with immediate offsets, those of the next pair of
//     sub-steps issued between the MFMAs of the current pair, so no MFMA ever waits for the LDS;
//   * the barrier of a step stands in front of its last eight MFMAs, and the first fragments of the
//     next step are read behind it, under those MFMAs.
// LDS images (16 KiB per operand and stage, two stages):
//   k-contiguous source  : [128 rows][8 chunks of 16 B]; row r keeps source chunk c at slot c ^ ((r >> 1) & 7)
//                          (the DMA writes LDS linearly, so the swizzle is applied to the source address)
//   row-contiguous source: [32 k][32 chunks of 16 B];     odd k keep row chunk q at slot q ^ 8
// so the 64 lanes of a fragment read (32 rows x 2 values of k) fall on at least 32 different banks.
// Measured (4096^3 and the 4096 x 10496 x 512 accumulating update of the column loop, TFLOP/s; the fp32 MFMA
// ceiling of the chip is 154.5, tools/src/mfma_rate_f32.hip): 104-124 -> 138-140 and 96-104 -> 119-122 against
// the compiler-scheduled kernel above (which read every fragment right in front of its MFMA and waited for
// all global loads at the top of every step); a k-contiguous operand whose leading dimension is a power of
// two stays at 127 (its 128 rows map to few memory channels).  A k-step takes 9260 cycles for the two
// workgroups of a CU against 8192 of pure MFMA issue.  A variant with 16-deep steps in three stages (48 KiB,
// three workgroups per CU, pieces fetched two steps ahead) reached 6190 cycles per three workgroups and
// step (99 % of the matrix pipe) and changed neither figure nor the bench: what is left is per launch, not
// per step -- all first workgroups load their 64 KiB of C at once (a 9 us prologue), the last round is
// partly empty, and a short-K update is 16 steps per tile.
template <int B, int E, typename F>
__device__ __forceinline__ void sg_for(F &&f)
{
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        sg_for<B + 1, E>(f);
    }
}

template <int OFF>
__device__ __forceinline__ void sg_read_b32(float &dst, unsigned addr)
{
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}

constexpr int SG_OP_BYTES = 128 * 32 * 4;          // one operand image
constexpr int SG_STAGE_BYTES = 2 * SG_OP_BYTES;    // A + B
constexpr int SG_LDS_BYTES = 2 * SG_STAGE_BYTES;   // two stages: 64 KiB, two workgroups per CU

template <int LAYOUT>
__global__ __launch_bounds__(256, 2) void sgemm_dma_kernel(GqSgemmArgs g, int tiles_m, int tiles_n)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr bool A_KM = (LAYOUT & GQ_A_KM) != 0;
    constexpr bool B_KN = (LAYOUT & GQ_B_KN) != 0;
    constexpr int BT = 128;
#ifdef GQ_SG_TRACE
    const unsigned long long t_begin = __builtin_readcyclecounter();
    const unsigned long long r_begin = __builtin_amdgcn_s_memrealtime();
    unsigned long long t_loop = 0;
#endif
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    g.a += (int64_t)blockIdx.y * g.stride_a;
    g.b += (int64_t)blockIdx.y * g.stride_b;
    g.c += (int64_t)blockIdx.y * g.stride_c;
    int tm, tn;
    if (g.lower_only) {
        const int bid = blockIdx.x;
        int r = (int)((sqrtf(8.0f * (float)bid + 1.0f) - 1.0f) * 0.5f);
        while ((r + 1) * (r + 2) / 2 <= bid) ++r;
        while (r * (r + 1) / 2 > bid) --r;
        tm = r;
        tn = bid - r * (r + 1) / 2;
    } else if (g.trim == GQ_TRIM_K_LE_N) {
        tn = tiles_n - 1 - (int)(blockIdx.x / tiles_m);
        tm = blockIdx.x % tiles_m;
    } else if (g.trim == GQ_TRIM_K_GE_N) {
        tn = blockIdx.x / tiles_m;
        tm = blockIdx.x % tiles_m;
    } else if (g.trim == GQ_TRIM_K_LE_M) {
        tm = tiles_m - 1 - (int)(blockIdx.x / tiles_n);
        tn = blockIdx.x % tiles_n;
    } else {
        tm = blockIdx.x / tiles_n;
        tn = blockIdx.x % tiles_n;
    }
    const int m0 = tm * BT, n0 = tn * BT;
    int k_lo = 0, k_hi = g.k;
    if (g.trim == GQ_TRIM_K_LE_N) k_hi = min(g.k, (tn + 1) * BT);
    else if (g.trim == GQ_TRIM_K_GE_N) k_lo = min(g.k, tn * BT);
    else if (g.trim == GQ_TRIM_K_LE_M) k_hi = min(g.k, (tm + 1) * BT);
    const int n_steps = (k_hi - k_lo) / BKS;

    const int wr = wave >> 1, wc = wave & 1;
    const int li = lane & 31, lh = lane >> 5;

    f32x16 acc[2][2];
    if (g.accumulate) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = m0 + (wr * 2 + a) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    const int col = n0 + (wc * 2 + b) * 32 + li;
                    acc[a][b][e] = g.c[(int64_t)row * g.ldc + col];
                }
    } else {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.0f;
    }
    const float sgn = g.negate ? -1.0f : 1.0f;

    // ---- DMA sources: a uniform base that advances by one k-step, and a 32-bit lane offset per piece
    unsigned voff_a[4], voff_b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int p = wave * 4 + j;
        if constexpr (A_KM) {
            const int k = 2 * p + (lane >> 5), q = (lane & 31) ^ (8 * (lane >> 5));
            voff_a[j] = (unsigned)(((int64_t)k * g.lda + 4 * q) * 4);
        } else {
            const int row = 8 * p + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
            voff_a[j] = (unsigned)(((int64_t)row * g.lda + 4 * c) * 4);
        }
        if constexpr (B_KN) {
            const int k = 2 * p + (lane >> 5), q = (lane & 31) ^ (8 * (lane >> 5));
            voff_b[j] = (unsigned)(((int64_t)k * g.ldb + 4 * q) * 4);
        } else {
            const int row = 8 * p + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
            voff_b[j] = (unsigned)(((int64_t)row * g.ldb + 4 * c) * 4);
        }
    }
    const char *ua = reinterpret_cast<const char *>(A_KM ? g.a + (int64_t)k_lo * g.lda + m0 : g.a + (int64_t)m0 * g.lda + k_lo);
    const char *ub = reinterpret_cast<const char *>(B_KN ? g.b + (int64_t)k_lo * g.ldb + n0 : g.b + (int64_t)n0 * g.ldb + k_lo);
    const int64_t inc_a = (A_KM ? (int64_t)BKS * g.lda : (int64_t)BKS) * 4, inc_b = (B_KN ? (int64_t)BKS * g.ldb : (int64_t)BKS) * 4;
    auto dma_a = [&](int stage, int j) __attribute__((always_inline)) {
        __builtin_amdgcn_global_load_lds(GQ_GLOBAL_PTR(ua + voff_a[j]),
                                         GQ_LDS_PTR(void, smem_raw + stage * SG_STAGE_BYTES + (wave * 4 + j) * 1024), 16, 0, 0);
    };
    auto dma_b = [&](int stage, int j) __attribute__((always_inline)) {
        __builtin_amdgcn_global_load_lds(GQ_GLOBAL_PTR(ub + voff_b[j]),
                                         GQ_LDS_PTR(void, smem_raw + stage * SG_STAGE_BYTES + SG_OP_BYTES + (wave * 4 + j) * 1024), 16, 0, 0);
    };

    // ---- fragment read addresses
    const unsigned lds_base = (unsigned)(uintptr_t)GQ_LDS_PTR(char, smem_raw);
    // k-contiguous image: address of chunk c of the lane's row (block u adds 4096, the second sub-step 8)
    // row-contiguous image: address of the lane's row at k = lh for block u (sub-step s adds 1024)
    unsigned addr_a[A_KM ? 2 : 8], addr_b[B_KN ? 2 : 8];
    if constexpr (A_KM) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int r = wr * 64 + u * 32 + li;
            addr_a[u] = lds_base + lh * 512 + ((((unsigned)r >> 2) ^ (8u * lh)) << 4) + ((r & 3) << 2);
        }
    } else {
        const int r = wr * 64 + li;
#pragma unroll
        for (int c = 0; c < 8; ++c) addr_a[c] = lds_base + r * 128 + lh * 4 + ((c ^ ((r >> 1) & 7)) << 4);
    }
    if constexpr (B_KN) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int r = wc * 64 + u * 32 + li;
            addr_b[u] = lds_base + SG_OP_BYTES + lh * 512 + ((((unsigned)r >> 2) ^ (8u * lh)) << 4) + ((r & 3) << 2);
        }
    } else {
        const int r = wc * 64 + li;
#pragma unroll
        for (int c = 0; c < 8; ++c) addr_b[c] = lds_base + SG_OP_BYTES + r * 128 + lh * 4 + ((c ^ ((r >> 1) & 7)) << 4);
    }

    float fa[2][2][2], fb[2][2][2];   // [set][sub-step of the pair][block]
    // the eight fragment reads of sub-step pair C of stage ST into register set SET, read number I (0..7)
    auto frag_read = [&](auto st_c, auto c_c, auto set_c, auto i_c) __attribute__((always_inline)) {
        constexpr int ST = decltype(st_c)::value, C = decltype(c_c)::value, SET = decltype(set_c)::value, I = decltype(i_c)::value;
        constexpr int OP = I >> 2, SUB = (I >> 1) & 1, U = I & 1;
        if constexpr (OP == 0) {
            if constexpr (A_KM) sg_read_b32<ST * SG_STAGE_BYTES + (2 * C + SUB) * 1024>(fa[SET][SUB][U], addr_a[U]);
            else sg_read_b32<ST * SG_STAGE_BYTES + U * 4096 + SUB * 8>(fa[SET][SUB][U], addr_a[C]);
        } else {
            if constexpr (B_KN) sg_read_b32<ST * SG_STAGE_BYTES + (2 * C + SUB) * 1024>(fb[SET][SUB][U], addr_b[U]);
            else sg_read_b32<ST * SG_STAGE_BYTES + U * 4096 + SUB * 8>(fb[SET][SUB][U], addr_b[C]);
        }
    };
    auto reads_done = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    if (n_steps > 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            dma_a(0, j);
            dma_b(0, j);
        }
        ua += inc_a;
        ub += inc_b;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        sg_for<0, 8>([&](auto i) { frag_read(I0{}, I0{}, I0{}, i); });
        reads_done();
    }
#ifdef GQ_SG_TRACE
    t_loop = __builtin_readcyclecounter();
#endif

    // one k-step on stage ST; MORE: another step follows (its pieces are fetched here, and its first
    // fragments read behind the barrier)
    auto k_step = [&](auto st_c, auto more_c) __attribute__((always_inline)) {
        constexpr int ST = decltype(st_c)::value;
        constexpr bool MORE = decltype(more_c)::value;
        sg_for<0, 8>([&](auto c_c) {
            constexpr int C = decltype(c_c)::value;
            constexpr int SET = C & 1;
            using NSET = std::integral_constant<int, SET ^ 1>;
            // the sign goes on the A fragments (exact); those of the second sub-step behind the first MFMA
#pragma unroll
            for (int u = 0; u < 2; ++u) fa[SET][0][u] *= sgn;
            if constexpr (C == 7 && MORE) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the next step
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
            sg_for<0, 8>([&](auto i_c) {
                constexpr int I = decltype(i_c)::value;
                constexpr int SUB = I >> 2, A = (I >> 1) & 1, B = I & 1;
                acc[A][B] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[SET][SUB][A], fb[SET][SUB][B], acc[A][B], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (I == 0) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) fa[SET][1][u] *= sgn;
                }
                // the eight reads of the next pair behind the first four MFMAs: they have the other four to arrive
                if constexpr (I < 4) {
                    using R0 = std::integral_constant<int, 2 * I>;
                    using R1 = std::integral_constant<int, 2 * I + 1>;
                    if constexpr (C < 7) {
                        frag_read(st_c, std::integral_constant<int, (C + 1) & 7>{}, NSET{}, R0{});
                        frag_read(st_c, std::integral_constant<int, (C + 1) & 7>{}, NSET{}, R1{});
                    } else if constexpr (MORE) {
                        frag_read(std::integral_constant<int, ST ^ 1>{}, I0{}, NSET{}, R0{});
                        frag_read(std::integral_constant<int, ST ^ 1>{}, I0{}, NSET{}, R1{});
                    }
                }
                if constexpr (MORE && C < 2) {
                    // pieces of the next step: A 0..3 in the first pair, B 0..3 in the second, one per two MFMAs
                    if constexpr ((I & 1) == 1) {
                        if constexpr (C == 0) dma_a(ST ^ 1, I >> 1);
                        else dma_b(ST ^ 1, I >> 1);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            if constexpr (C < 7 || MORE) reads_done();
        });
        if constexpr (MORE) {
            ua += inc_a;
            ub += inc_b;
        }
    };
    {
        using T = std::true_type;
        using F = std::false_type;
        int step = 0;
        for (; step + 2 < n_steps; step += 2) {
            k_step(I0{}, T{});
            k_step(I1{}, T{});
        }
        if (n_steps - step == 2) {
            k_step(I0{}, T{});
            k_step(I1{}, F{});
        } else if (n_steps - step == 1) {
            k_step(I0{}, F{});
        }
    }
#ifdef GQ_SG_TRACE
    const unsigned long long t_end = __builtin_readcyclecounter();
#endif

#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + (wr * 2 + a) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                const int col = n0 + (wc * 2 + b) * 32 + li;
                g.c[(int64_t)row * g.ldc + col] = acc[a][b][e];
            }
#ifdef GQ_SG_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (g_sg_trace && blockIdx.x == 300 && lane == 0) {
        unsigned long long *o = g_sg_trace + wave * 8;
        o[0] = t_begin; o[1] = t_loop; o[2] = t_end; o[3] = __builtin_readcyclecounter(); o[4] = r_begin;
        o[5] = __builtin_amdgcn_s_memrealtime(); o[6] = n_steps;
    }
#endif
}

// ---- 64x64 tiles with a K range of at most 128: the short products on the dependency chains ----
// The panel and near-update products of the Cholesky factorisation and the near updates of the column loop
// are 128 deep, a few hundred 64-tiles at most, and hundreds of them follow each other per decoder block:
// what they cost is latency, not arithmetic (64 MFMAs per wave).  Here the whole K range of a tile -- up to
// four 32-deep stages of both operands, 64 KiB -- is fetched by LDS-DMA at once, together with the tile of
// C, so a tile is ONE memory round trip, then its MFMA chain with the fragment reads of the next pair of
// sub-steps issued under the current pair, then the stores; no barrier inside.  Same images and swizzles as
// the 128-tile kernel above (64 rows: a k-contiguous image is [64 rows][8 chunks], a row-contiguous one
// [32 k][16 chunks] with the halves swapped on odd k), same k-ascending chain per element.
constexpr int SS_OP_BYTES = 64 * 32 * 4;           // one operand image of one stage: 8 KiB
constexpr int SS_STAGE_BYTES = 2 * SS_OP_BYTES;
constexpr int SS_LDS_BYTES = 4 * SS_STAGE_BYTES;   // 64 KiB

template <int LAYOUT>
__global__ __launch_bounds__(256, 2) void sgemm_small_kernel(GqSgemmArgs g, int tiles_m, int tiles_n)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr bool A_KM = (LAYOUT & GQ_A_KM) != 0;
    constexpr bool B_KN = (LAYOUT & GQ_B_KN) != 0;
    constexpr int BT = 64;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    g.a += (int64_t)blockIdx.y * g.stride_a;
    g.b += (int64_t)blockIdx.y * g.stride_b;
    g.c += (int64_t)blockIdx.y * g.stride_c;
    int tm, tn;
    if (g.lower_only) {
        const int bid = blockIdx.x;
        int r = (int)((sqrtf(8.0f * (float)bid + 1.0f) - 1.0f) * 0.5f);
        while ((r + 1) * (r + 2) / 2 <= bid) ++r;
        while (r * (r + 1) / 2 > bid) --r;
        tm = r;
        tn = bid - r * (r + 1) / 2;
    } else {
        tm = blockIdx.x / tiles_n;
        tn = blockIdx.x % tiles_n;
    }
    const int m0 = tm * BT, n0 = tn * BT;
    // the K range of the tile (one operand triangular), cut at the tile's own edge
    int k_lo = 0, k_hi = g.k;
    if (g.trim == GQ_TRIM_K_LE_N) k_hi = min(g.k, (tn + 1) * BT);
    else if (g.trim == GQ_TRIM_K_GE_N) k_lo = min(g.k, tn * BT);
    else if (g.trim == GQ_TRIM_K_LE_M) k_hi = min(g.k, (tm + 1) * BT);
    const int n_steps = (k_hi - k_lo) / BKS;   // 0..4 (the host checks k <= 128)

    const int wr = wave >> 1, wc = wave & 1;
    const int li = lane & 31, lh = lane >> 5;

    // everything the tile needs in one go: the pieces of all stages, then the tile of C
    {
        const char *ua = reinterpret_cast<const char *>(A_KM ? g.a + (int64_t)k_lo * g.lda + m0 : g.a + (int64_t)m0 * g.lda + k_lo);
        const char *ub = reinterpret_cast<const char *>(B_KN ? g.b + (int64_t)k_lo * g.ldb + n0 : g.b + (int64_t)n0 * g.ldb + k_lo);
        const int64_t inc_a = (A_KM ? (int64_t)BKS * g.lda : (int64_t)BKS) * 4, inc_b = (B_KN ? (int64_t)BKS * g.ldb : (int64_t)BKS) * 4;
        unsigned voff_a[2], voff_b[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int p = wave * 2 + j;   // piece 0..7 of an 8 KiB image
            if constexpr (A_KM) {
                const int k = 4 * p + (lane >> 4), q = (lane & 15) ^ (8 * (k & 1));
                voff_a[j] = (unsigned)(((int64_t)k * g.lda + 4 * q) * 4);
            } else {
                const int row = 8 * p + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
                voff_a[j] = (unsigned)(((int64_t)row * g.lda + 4 * c) * 4);
            }
            if constexpr (B_KN) {
                const int k = 4 * p + (lane >> 4), q = (lane & 15) ^ (8 * (k & 1));
                voff_b[j] = (unsigned)(((int64_t)k * g.ldb + 4 * q) * 4);
            } else {
                const int row = 8 * p + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
                voff_b[j] = (unsigned)(((int64_t)row * g.ldb + 4 * c) * 4);
            }
        }
        for (int st = 0; st < n_steps; ++st) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                __builtin_amdgcn_global_load_lds(GQ_GLOBAL_PTR(ua + voff_a[j]),
                                                 GQ_LDS_PTR(void, smem_raw + st * SS_STAGE_BYTES + (wave * 2 + j) * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds(GQ_GLOBAL_PTR(ub + voff_b[j]),
                                                 GQ_LDS_PTR(void, smem_raw + st * SS_STAGE_BYTES + SS_OP_BYTES + (wave * 2 + j) * 1024), 16, 0, 0);
            }
            ua += inc_a;
            ub += inc_b;
        }
    }
    f32x16 acc;
    if (g.accumulate) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = m0 + wr * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
            const int col = n0 + wc * 32 + li;
            acc[e] = g.c[(int64_t)row * g.ldc + col];
        }
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
    }
    const float sgn = g.negate ? -1.0f : 1.0f;

    const unsigned lds_base = (unsigned)(uintptr_t)GQ_LDS_PTR(char, smem_raw);
    unsigned addr_a[A_KM ? 1 : 8], addr_b[B_KN ? 1 : 8];
    if constexpr (A_KM) {
        const int r = wr * 32 + li;
        addr_a[0] = lds_base + lh * 256 + ((((unsigned)r >> 2) ^ (8u * lh)) << 4) + ((r & 3) << 2);
    } else {
        const int r = wr * 32 + li;
#pragma unroll
        for (int c = 0; c < 8; ++c) addr_a[c] = lds_base + r * 128 + lh * 4 + ((c ^ ((r >> 1) & 7)) << 4);
    }
    if constexpr (B_KN) {
        const int r = wc * 32 + li;
        addr_b[0] = lds_base + SS_OP_BYTES + lh * 256 + ((((unsigned)r >> 2) ^ (8u * lh)) << 4) + ((r & 3) << 2);
    } else {
        const int r = wc * 32 + li;
#pragma unroll
        for (int c = 0; c < 8; ++c) addr_b[c] = lds_base + SS_OP_BYTES + r * 128 + lh * 4 + ((c ^ ((r >> 1) & 7)) << 4);
    }
    float fa[2][2], fb[2][2];   // [set][sub-step of the pair]
    auto frag_read = [&](auto st_c, auto c_c, auto set_c, auto i_c) __attribute__((always_inline)) {
        constexpr int ST = decltype(st_c)::value, C = decltype(c_c)::value, SET = decltype(set_c)::value, I = decltype(i_c)::value;
        constexpr int OP = I >> 1, SUB = I & 1;
        if constexpr (OP == 0) {
            if constexpr (A_KM) sg_read_b32<ST * SS_STAGE_BYTES + (2 * C + SUB) * 512>(fa[SET][SUB], addr_a[0]);
            else sg_read_b32<ST * SS_STAGE_BYTES + SUB * 8>(fa[SET][SUB], addr_a[C]);
        } else {
            if constexpr (B_KN) sg_read_b32<ST * SS_STAGE_BYTES + (2 * C + SUB) * 512>(fb[SET][SUB], addr_b[0]);
            else sg_read_b32<ST * SS_STAGE_BYTES + SUB * 8>(fb[SET][SUB], addr_b[C]);
        }
    };
    using I0 = std::integral_constant<int, 0>;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the DMA pieces (the loads of C are older: they are in too)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (n_steps > 0) {
        sg_for<0, 4>([&](auto i) { frag_read(I0{}, I0{}, I0{}, i); });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
    // stage ST; MORE: another stage follows (its first fragments are read under this stage's last pair)
    auto k_step = [&](auto st_c, auto more_c) __attribute__((always_inline)) {
        constexpr int ST = decltype(st_c)::value;
        constexpr bool MORE = decltype(more_c)::value;
        sg_for<0, 8>([&](auto c_c) {
            constexpr int C = decltype(c_c)::value;
            constexpr int SET = C & 1;
            using NSET = std::integral_constant<int, SET ^ 1>;
            fa[SET][0] *= sgn;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[SET][0], fb[SET][0], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            fa[SET][1] *= sgn;
            sg_for<0, 4>([&](auto i_c) {
                if constexpr (C < 7) frag_read(st_c, std::integral_constant<int, (C + 1) & 7>{}, NSET{}, i_c);
                else if constexpr (MORE) frag_read(std::integral_constant<int, (ST + 1) & 3>{}, I0{}, NSET{}, i_c);
            });
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[SET][1], fb[SET][1], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (C < 7 || MORE) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
        });
    };
    {
        using T = std::true_type;
        using F = std::false_type;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        using I3 = std::integral_constant<int, 3>;
        if (n_steps == 4) { k_step(I0{}, T{}); k_step(I1{}, T{}); k_step(I2{}, T{}); k_step(I3{}, F{}); }
        else if (n_steps == 3) { k_step(I0{}, T{}); k_step(I1{}, T{}); k_step(I2{}, F{}); }
        else if (n_steps == 2) { k_step(I0{}, T{}); k_step(I1{}, F{}); }
        else if (n_steps == 1) { k_step(I0{}, F{}); }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int row = m0 + wr * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        const int col = n0 + wc * 32 + li;
        g.c[(int64_t)row * g.ldc + col] = acc[e];
    }
}

int launch_sgemm_small(const GqSgemmArgs &args, hipStream_t stream)
{
    const int tiles_m = args.m / 64, tiles_n = args.n / 64;
    GQ_REQUIRE(!args.lower_only || tiles_m == tiles_n, "sgemm: lower_only needs a square tile grid");
    const int n_blocks = args.lower_only ? tiles_m * (tiles_m + 1) / 2 : tiles_m * tiles_n;
    void (*kern)(GqSgemmArgs, int, int) = nullptr;
    switch (args.layout) {
    case GQ_A_MK | GQ_B_NK: kern = sgemm_small_kernel<GQ_A_MK | GQ_B_NK>; break;
    case GQ_A_MK | GQ_B_KN: kern = sgemm_small_kernel<GQ_A_MK | GQ_B_KN>; break;
    case GQ_A_KM | GQ_B_NK: kern = sgemm_small_kernel<GQ_A_KM | GQ_B_NK>; break;
    case GQ_A_KM | GQ_B_KN: kern = sgemm_small_kernel<GQ_A_KM | GQ_B_KN>; break;
    default: GQ_REQUIRE(false, "sgemm: bad layout %d", args.layout);
    }
    GQ_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SS_LDS_BYTES));
    hipLaunchKernelGGL(kern, dim3(n_blocks, std::max(args.batch, 1)), dim3(256), SS_LDS_BYTES, stream, args, tiles_m, tiles_n);
    GQ_LAUNCH_CHECK();
    return GQ_OK;
}

int launch_sgemm_dma(const GqSgemmArgs &args, hipStream_t stream)
{
    const int tiles_m = args.m / 128, tiles_n = args.n / 128;
    GQ_REQUIRE(!args.lower_only || tiles_m == tiles_n, "sgemm: lower_only needs a square tile grid");
    const int n_blocks = args.lower_only ? tiles_m * (tiles_m + 1) / 2 : tiles_m * tiles_n;
    void (*kern)(GqSgemmArgs, int, int) = nullptr;
    switch (args.layout) {
    case GQ_A_MK | GQ_B_NK: kern = sgemm_dma_kernel<GQ_A_MK | GQ_B_NK>; break;
    case GQ_A_MK | GQ_B_KN: kern = sgemm_dma_kernel<GQ_A_MK | GQ_B_KN>; break;
    case GQ_A_KM | GQ_B_NK: kern = sgemm_dma_kernel<GQ_A_KM | GQ_B_NK>; break;
    case GQ_A_KM | GQ_B_KN: kern = sgemm_dma_kernel<GQ_A_KM | GQ_B_KN>; break;
    default: GQ_REQUIRE(false, "sgemm: bad layout %d", args.layout);
    }
    GQ_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SG_LDS_BYTES));
    hipLaunchKernelGGL(kern, dim3(n_blocks, std::max(args.batch, 1)), dim3(256), SG_LDS_BYTES, stream, args, tiles_m, tiles_n);
    GQ_LAUNCH_CHECK();
    return GQ_OK;
}

template <int BT>
int launch_sgemm(const GqSgemmArgs &args, hipStream_t stream)
{
    constexpr int LDS_BYTES = 2 * 2 * BT * 33 * 4;
    const int tiles_m = (args.m + BT - 1) / BT, tiles_n = (args.n + BT - 1) / BT;
    GQ_REQUIRE(!args.lower_only || tiles_m == tiles_n, "sgemm: lower_only needs a square tile grid");
    const int n_blocks = args.lower_only ? tiles_m * (tiles_m + 1) / 2 : tiles_m * tiles_n;
    void (*kern)(GqSgemmArgs, int, int) = nullptr;
    switch (args.layout) {
    case GQ_A_MK | GQ_B_NK: kern = sgemm_kernel<GQ_A_MK | GQ_B_NK, BT>; break;
    case GQ_A_MK | GQ_B_KN: kern = sgemm_kernel<GQ_A_MK | GQ_B_KN, BT>; break;
    case GQ_A_KM | GQ_B_NK: kern = sgemm_kernel<GQ_A_KM | GQ_B_NK, BT>; break;
    case GQ_A_KM | GQ_B_KN: kern = sgemm_kernel<GQ_A_KM | GQ_B_KN, BT>; break;
    default: GQ_REQUIRE(false, "sgemm: bad layout %d", args.layout);
    }
    GQ_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    hipLaunchKernelGGL(kern, dim3(n_blocks, std::max(args.batch, 1)), dim3(256), LDS_BYTES, stream, args, tiles_m, tiles_n);
    GQ_LAUNCH_CHECK();
    return GQ_OK;
}

}  // namespace

#ifdef GQ_SG_TRACE
extern "C" int gq_debug_sg_trace(unsigned long long *buf)
{
    GQ_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_sg_trace), &buf, sizeof(buf)));
    return GQ_OK;
}
#endif

static bool g_sgemm_plain = false;
void gq_sgemm_set_plain(int on) { g_sgemm_plain = on != 0; }

int gq_sgemm(const GqSgemmArgs &args, hipStream_t stream)
{
    GQ_REQUIRE(args.a && args.b && args.c, "sgemm: NULL operand");
    GQ_REQUIRE(args.m >= 0 && args.n >= 0 && args.k >= 0, "sgemm: negative dimension");
    GQ_REQUIRE(args.m % 4 == 0 && args.n % 4 == 0 && args.k % 4 == 0 && args.lda % 4 == 0 && args.ldb % 4 == 0 &&
                   args.ldc % 4 == 0,
               "sgemm: m=%d n=%d k=%d lda=%lld ldb=%lld ldc=%lld must all be multiples of 4", args.m, args.n, args.k,
               (long long)args.lda, (long long)args.ldb, (long long)args.ldc);
    GQ_REQUIRE((((uintptr_t)args.a | (uintptr_t)args.b | (uintptr_t)args.c) & 15) == 0, "sgemm: operands must be 16-byte aligned");
    if (args.m == 0 || args.n == 0) return GQ_OK;
    // 128-tiles unless they would occupy fewer than the chip's 256 CUs
    const int64_t tm = (args.m + 127) / 128, tn = (args.n + 127) / 128;
    GQ_REQUIRE(args.batch >= 0 && args.batch <= 65535, "sgemm: batch %d", args.batch);
    GQ_REQUIRE(args.batch <= 1 || (args.stride_a % 4 == 0 && args.stride_b % 4 == 0 && args.stride_c % 4 == 0),
               "sgemm: batch strides must be multiples of 4");
    const int64_t blocks128 = (args.lower_only ? tm * (tm + 1) / 2 : tm * tn) * std::max(args.batch, 1);
    static const bool plain_env = [] { const char *e = getenv("GQ_SGEMM"); return e && e[0] == 'p'; }();
    const bool plain = plain_env || g_sgemm_plain;
    const int64_t span_a = (args.layout & GQ_A_KM) ? 32 * args.lda + 128 : 128 * args.lda + 32;
    const int64_t span_b = (args.layout & GQ_B_KN) ? 32 * args.ldb + 128 : 128 * args.ldb + 32;
    const bool dma_ok = !plain && args.k % BKS == 0 && span_a < ((int64_t)1 << 29) && span_b < ((int64_t)1 << 29);
    if (blocks128 < 256) {
        // short products (K <= 128) of whole 64-tiles: everything of a tile in one memory round trip
        if (dma_ok && args.k <= 128 && args.m % 64 == 0 && args.n % 64 == 0) return launch_sgemm_small(args, stream);
        return launch_sgemm<64>(args, stream);
    }
    // whole 128-tiles, whole k-steps, lane offsets of the DMA within 32 bits: the hand-scheduled kernel
    if (dma_ok && args.m % 128 == 0 && args.n % 128 == 0) return launch_sgemm_dma(args, stream);
    return launch_sgemm<128>(args, stream);
}

// H += X^T X for fp32 activations: the reference's fp32 addmm (gptq_py.py:31-33).  Full product;
// H[i][j] and H[j][i] are the same k-ascending chain of commutative fused multiply-adds, so H comes
// out exactly symmetric.
extern "C" int gq_xtx_accumulate_f32(float *hessian, const float *x, int64_t n_tokens, int64_t hidden, gq_stream_t stream)
{
    GQ_REQUIRE(hessian && (x || n_tokens == 0), "gq_xtx_accumulate_f32: NULL pointer");
    GQ_REQUIRE(n_tokens >= 0 && hidden > 0 && hidden < (1 << 30) && n_tokens < ((int64_t)1 << 31),
               "gq_xtx_accumulate_f32: bad shape tokens=%lld hidden=%lld", (long long)n_tokens, (long long)hidden);
    GQ_REQUIRE(n_tokens % 4 == 0 && hidden % 4 == 0,
               "gq_xtx_accumulate_f32: tokens (%lld) and hidden (%lld) must be multiples of 4 (pad with zero rows)",
               (long long)n_tokens, (long long)hidden);
    if (n_tokens == 0) return GQ_OK;
    GqSgemmArgs g{};
    g.a = x; g.b = x; g.c = hessian;
    g.lda = g.ldb = g.ldc = hidden;
    g.m = g.n = (int)hidden;
    g.k = (int)n_tokens;
    g.layout = GQ_A_KM | GQ_B_KN;
    g.trim = GQ_TRIM_NONE;
    g.negate = 0;
    g.accumulate = 1;
    g.lower_only = 0;
    return gq_sgemm(g, (hipStream_t)stream);
}
