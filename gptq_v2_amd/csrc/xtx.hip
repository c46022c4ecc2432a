// H += X^T X  (fp16 / bf16 activations, fp32 Hessian) for gfx950.
//
// Replaces the reference's CUTLASS op (package/gptq/accumulate_hessian_kernel.cu:10-84) and its
// Triton op (gptq_triton/accumulate_hessian.py:55-149): same result contract (fp32 accumulation
// of exact 16-bit products, optional lower-triangle-only compute with mirrored store, exactly
// symmetric output), different construction:
//
//   * X is [tokens][hidden] row-major, so BOTH MFMA operands are "k-strided" (the contraction
//     index is the slow one).  Tiles are staged global -> LDS with 16-byte LDS-DMA loads exactly
//     as they lie in memory ([k][128 columns], 256-byte rows) and the per-lane fragments are
//     read with ds_read_b64_tr_b16, the hardware transposing read, so no transpose pass over X
//     and no register shuffles are needed.  A and B fragments come from the same routine applied
//     to two column ranges of X, hence H[i][j] and H[j][i] see the same products in the same
//     order and diagonal tiles come out exactly symmetric.
//   * 256-byte LDS rows would put the four k-rows of one transposing read on the same banks;
//     the 32-byte column chunk index is XOR-ed with 2*(k & 3).  LDS-DMA writes LDS linearly, so
//     the swizzle is applied to the per-lane global SOURCE address and again on the read.
//   * one 128x128 output tile per 256-thread workgroup, 2x2 waves of 64x64, 32x32x16 MFMA,
//     64-token k-steps, two LDS stages, one barrier per step; the next stage's DMA is issued right
//     after the barrier and lands under this step's LDS reads and MFMAs.
//   * only tiles with tile_row >= tile_col are launched in lower-only mode; blockIdx is remapped
//     so that each XCD (blocks b, b+8, ... share one) walks a contiguous run of the triangular
//     tile order and re-uses X panels from its own L2.
#include <type_traits>

#include "gq_common.h"

namespace {

constexpr int TM = 128;        // output tile edge
constexpr int BK = 64;         // tokens per k-step
constexpr int OP_BYTES = BK * TM * 2;     // one operand tile in LDS: 16 KiB
constexpr int STAGE_BYTES = 2 * OP_BYTES;  // A and B
constexpr int XTX_LDS = 2 * STAGE_BYTES;   // two stages: 64 KiB

// One operand tile (64 tokens x 128 columns of 16-bit) into LDS by LDS-DMA: 16 wave-instructions
// of 1 KiB (= 4 token rows), 4 per wave.  LDS position P (bytes) holds
//   token  k = P / 256,   32-byte chunk (P % 256) / 32  ^  2 * (k & 3),   half (P / 16) & 1.
__device__ __forceinline__ void stage_operand(const unsigned short *__restrict__ x, int64_t ld,
                                              int64_t tok0, int col0, char *lds_op, int wave, int lane)
{
    const int krow = lane >> 4;             // 0..3 == k & 3
    const int s = lane & 15;                // 16-byte slot in the 256-byte row
    const int chunk = (s >> 1) ^ (2 * krow);
    const int coff = chunk * 16 + (s & 1) * 8;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int ii = wave * 4 + it;       // wave-uniform instruction index 0..15
        const unsigned short *src = x + (tok0 + ii * 4 + krow) * ld + col0 + coff;
        __builtin_amdgcn_global_load_lds(GQ_GLOBAL_PTR(src), GQ_LDS_PTR(void, lds_op + ii * 1024), 16, 0, 0);
    }
}

template <int DTYPE>
__device__ __forceinline__ f32x16 mfma16(s16x8 a, s16x8 b, f32x16 c)
{
    if constexpr (DTYPE == GQ_F16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// Transposing LDS read issued from inline asm: hipcc must not see these as LDS loads, or it
// orders them behind the in-flight LDS-DMA of the OTHER stage with a vmcnt(0).  The waits are
// counted by hand below (lgkmcnt), each followed by sched_barrier(0) so no MFMA is hoisted
// above the wait that guards its operands.
template <int OFF>
__device__ __forceinline__ void tr_read(s16x4 &dst, unsigned lds_addr)
{
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(lds_addr), "i"(OFF));
}

// fragments of one 16-token MFMA step: 2 A blocks + 2 B blocks, two transposing reads each
struct Frags {
    s16x4 a[2][2], b[2][2];
};

template <int BASE>
__device__ __forceinline__ void read_step(Frags &f, const unsigned (&addrA)[2], const unsigned (&addrB)[2])
{
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        tr_read<BASE>(f.a[u][0], addrA[u]);
        tr_read<BASE + 4 * 256>(f.a[u][1], addrA[u]);
        tr_read<BASE + OP_BYTES>(f.b[u][0], addrB[u]);
        tr_read<BASE + OP_BYTES + 4 * 256>(f.b[u][1], addrB[u]);
    }
}

template <int DTYPE>
__device__ __forceinline__ void mfma_step(f32x16 (&acc)[2][2], const Frags &f)
{
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const s16x8 fa = __builtin_shufflevector(f.a[a][0], f.a[a][1], 0, 1, 2, 3, 4, 5, 6, 7);
            const s16x8 fb = __builtin_shufflevector(f.b[b][0], f.b[b][1], 0, 1, 2, 3, 4, 5, 6, 7);
            acc[a][b] = mfma16<DTYPE>(fa, fb, acc[a][b]);
        }
}

__device__ __forceinline__ void tile_of_block(int bid, int nt, int lower, int &ti, int &tj)
{
    if (lower) {
        // row-major walk of the lower triangle: t = ti*(ti+1)/2 + tj
        int r = (int)((sqrtf(8.0f * (float)bid + 1.0f) - 1.0f) * 0.5f);
        while ((r + 1) * (r + 2) / 2 <= bid) ++r;
        while (r * (r + 1) / 2 > bid) --r;
        ti = r;
        tj = bid - r * (r + 1) / 2;
    } else {
        ti = bid / nt;
        tj = bid % nt;
    }
}

template <int DTYPE>
__global__ __launch_bounds__(256) void xtx_mfma_kernel(float *__restrict__ hess, const unsigned short *__restrict__ x,
                                                       int hidden, int n_ktiles, int nt, int n_blocks,
                                                       int compute_lower, int mirror)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;

    // XCD-aware remap (bijective for any grid size): blocks with equal blockIdx % 8 share an XCD
    // and get a contiguous range of logical tile ids.
    int bid;
    {
        const int orig = blockIdx.x;
        const int xcd = orig & 7;
        const int q = n_blocks >> 3, r = n_blocks & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    int ti, tj;
    tile_of_block(bid, nt, compute_lower, ti, tj);
    const int m0 = ti * TM, n0 = tj * TM;
    const int64_t ld = hidden;

    const int wr = wave >> 1, wc = wave & 1;
    // per-lane LDS byte offsets of the transposing reads (see header comment)
    const int g = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, p4 = i16 & 3;
    const int g1 = g & 1, kh = g >> 1;
    const unsigned lds_base = (unsigned)(uintptr_t)GQ_LDS_PTR(char, smem);
    unsigned addrA[2], addrB[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int ca = ((wr * 2 + u) * 2 + g1) ^ (2 * q4);
        const int cb = ((wc * 2 + u) * 2 + g1) ^ (2 * q4);
        addrA[u] = lds_base + (8 * kh + q4) * 256 + ca * 32 + p4 * 8;
        addrB[u] = lds_base + (8 * kh + q4) * 256 + cb * 32 + p4 * 8;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.0f;

    stage_operand(x, ld, 0, m0, smem, wave, lane);
    stage_operand(x, ld, 0, n0, smem + OP_BYTES, wave, lane);

    // One barrier per k-step: after it every wave's DMA for stage `cur` has landed (each wave
    // waited for its own) and every wave has finished reading the other stage (it completed its
    // lgkmcnt(0) before arriving), so the other stage is restaged right away and its DMA flies
    // under this step's reads and MFMAs.  Two steps per trip keep all LDS offsets immediates.
    auto step = [&](auto cur_c, int kt) {
        constexpr int cur = decltype(cur_c)::value;
        constexpr int SB = cur * STAGE_BYTES;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + 1 < n_ktiles) {
            char *nxt = smem + (cur ^ 1) * STAGE_BYTES;
            stage_operand(x, ld, (int64_t)(kt + 1) * BK, m0, nxt, wave, lane);
            stage_operand(x, ld, (int64_t)(kt + 1) * BK, n0, nxt + OP_BYTES, wave, lane);
        }
        Frags f0, f1;
        read_step<SB + 0 * 16 * 256>(f0, addrA, addrB);
        read_step<SB + 1 * 16 * 256>(f1, addrA, addrB);
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        mfma_step<DTYPE>(acc, f0);
        __builtin_amdgcn_sched_barrier(0);
        read_step<SB + 2 * 16 * 256>(f0, addrA, addrB);
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        mfma_step<DTYPE>(acc, f1);
        __builtin_amdgcn_sched_barrier(0);
        read_step<SB + 3 * 16 * 256>(f1, addrA, addrB);
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        mfma_step<DTYPE>(acc, f0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        mfma_step<DTYPE>(acc, f1);
    };
    int kt = 0;
    for (; kt + 1 < n_ktiles; kt += 2) {
        step(std::integral_constant<int, 0>{}, kt);
        step(std::integral_constant<int, 1>{}, kt + 1);
    }
    if (kt < n_ktiles) step(std::integral_constant<int, 0>{}, kt);   // odd step count: stage 0 is next

    // epilogue: H tile += acc; strictly-lower tiles are also written to their mirror position
    const bool do_mirror = mirror && (ti != tj);
    const int lrow = 4 * (lane >> 5), lcol = lane & 31;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + (wr * 2 + a) * 32 + (e & 3) + 8 * (e >> 2) + lrow;
                const int col = n0 + (wc * 2 + b) * 32 + lcol;
                float *p = hess + (int64_t)row * ld + col;
                const float v = *p + acc[a][b][e];
                *p = v;
                if (do_mirror) hess[(int64_t)col * ld + row] = v;
            }
}

// Generic path: any hidden size, any token range [tok0, tok1).  64x64 output tile per 256-thread
// workgroup, 4x4 outputs per thread, fp32 fma in token order.  Used for the token tail that does
// not fill a 64-token step and for hidden sizes that are not a multiple of 128.
template <int DTYPE>
__global__ __launch_bounds__(256) void xtx_generic_kernel(float *__restrict__ hess, const unsigned short *__restrict__ x,
                                                          int hidden, int64_t tok0, int64_t tok1, int nt,
                                                          int compute_lower, int mirror)
{
    __shared__ float xs[2][16][64 + 4];
    int ti, tj;
    tile_of_block(blockIdx.x, nt, compute_lower, ti, tj);
    const int m0 = ti * 64, n0 = tj * 64;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    float acc[4][4] = {};
    for (int64_t t0 = tok0; t0 < tok1; t0 += 16) {
        for (int e = threadIdx.x; e < 2 * 16 * 64; e += 256) {
            const int op = e / (16 * 64), rem = e % (16 * 64), kk = rem / 64, cc = rem % 64;
            const int col = (op ? n0 : m0) + cc;
            const int64_t tok = t0 + kk;
            float v = 0.0f;
            if (tok < tok1 && col < hidden) {
                const unsigned short raw = x[tok * hidden + col];
                v = DTYPE == GQ_F16 ? (float)__builtin_bit_cast(_Float16, raw) : gq_bf16_to_f32(raw);
            }
            xs[op][kk][cc] = v;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float av[4], bv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                av[u] = xs[0][kk][ty * 4 + u];
                bv[u] = xs[1][kk][tx * 4 + u];
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = fmaf(av[a], bv[b], acc[a][b]);
        }
        __syncthreads();
    }
    const bool diag = ti == tj;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int row = m0 + ty * 4 + a, col = n0 + tx * 4 + b;
            if (row >= hidden || col >= hidden) continue;
            float *p = hess + (int64_t)row * hidden + col;
            const float v = *p + acc[a][b];
            *p = v;
            if (mirror && !diag) hess[(int64_t)col * hidden + row] = v;
        }
}

template <int DTYPE>
int launch_xtx(float *hess, const unsigned short *x, int64_t n_tokens, int64_t hidden, int save_lower_only,
               int compute_lower_only, hipStream_t stream)
{
    const int mirror = compute_lower_only && !save_lower_only;
    int64_t done = 0;
    if (hidden % TM == 0 && n_tokens >= BK) {
        const int nt = (int)(hidden / TM);
        const int n_ktiles = (int)(n_tokens / BK);
        const int n_blocks = compute_lower_only ? nt * (nt + 1) / 2 : nt * nt;
        GQ_HIP(hipFuncSetAttribute((const void *)xtx_mfma_kernel<DTYPE>, hipFuncAttributeMaxDynamicSharedMemorySize, XTX_LDS));
        hipLaunchKernelGGL(xtx_mfma_kernel<DTYPE>, dim3(n_blocks), dim3(256), XTX_LDS, stream, hess, x, (int)hidden,
                           n_ktiles, nt, n_blocks, compute_lower_only, mirror);
        GQ_LAUNCH_CHECK();
        done = (int64_t)n_ktiles * BK;
    }
    if (done < n_tokens) {
        const int nt = (int)gq_cdiv(hidden, 64);
        const int n_blocks = compute_lower_only ? nt * (nt + 1) / 2 : nt * nt;
        hipLaunchKernelGGL(xtx_generic_kernel<DTYPE>, dim3(n_blocks), dim3(256), 0, stream, hess, x, (int)hidden, done,
                           n_tokens, nt, compute_lower_only, mirror);
        GQ_LAUNCH_CHECK();
    }
    return GQ_OK;
}

}  // namespace

extern "C" int gq_xtx_accumulate(float *hessian, const void *x, int dtype, int64_t n_tokens, int64_t hidden,
                                 int save_lower_only, int compute_lower_only, gq_stream_t stream)
{
    GQ_REQUIRE(hessian && x, "gq_xtx_accumulate: NULL pointer");
    GQ_REQUIRE(dtype == GQ_F16 || dtype == GQ_BF16, "gq_xtx_accumulate: dtype %d is neither fp16 (0) nor bf16 (1)", dtype);
    GQ_REQUIRE(n_tokens >= 0 && hidden > 0 && hidden < (1 << 30), "gq_xtx_accumulate: bad shape tokens=%lld hidden=%lld",
               (long long)n_tokens, (long long)hidden);
    GQ_REQUIRE(compute_lower_only || !save_lower_only, "compute_lower_only must be set when save_lower_only is set");
    GQ_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)hessian & 15) == 0, "gq_xtx_accumulate: pointers must be 16-byte aligned");
    if (n_tokens == 0) return GQ_OK;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GQ_F16) return launch_xtx<GQ_F16>(hessian, (const unsigned short *)x, n_tokens, hidden, save_lower_only, compute_lower_only, s);
    return launch_xtx<GQ_BF16>(hessian, (const unsigned short *)x, n_tokens, hidden, save_lower_only, compute_lower_only, s);
}
