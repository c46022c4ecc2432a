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
    if (blocks128 < 256) return launch_sgemm<64>(args, stream);
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
