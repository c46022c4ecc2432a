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
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <type_traits>

#include "gq_common.h"

namespace {

constexpr int TM = 128;        // output tile edge

// Tile order: the (lower triangle of the) nt x nt tile grid cut into S x S super-tiles, super-tiles
// row-major, tiles row-major inside each; position `pos` of that order -> (tile row, tile column).
// Computed by every workgroup for itself (a loop over at most nt / S super-rows of scalar arithmetic):
// no table in device memory, so the library allocates nothing.  order = nt | S << 16 | lower << 24.
__host__ __device__ __forceinline__ uint32_t xtx_order(int nt, int super, int lower)
{
    return (uint32_t)nt | ((uint32_t)super << 16) | ((uint32_t)(lower != 0) << 24);
}

__host__ __device__ __forceinline__ void xtx_tile_of(int pos, uint32_t order, int &ti, int &tj)
{
    const int nt = (int)(order & 0xffffu), S = (int)((order >> 16) & 0xffu);
    const bool lower = (order >> 24) != 0;
    const int ns = (nt + S - 1) / S;
    ti = tj = 0;
    for (int si = 0; si < ns; ++si) {
        const int rows = nt - si * S < S ? nt - si * S : S;
        // super-row si: `full` tiles in complete super-tiles left of the diagonal one (all super-tiles but the
        // last of the row when the whole grid is wanted), then the diagonal (or last) super-tile
        const int n_full = lower ? si : ns - 1;
        const int full = n_full * rows * S;
        const int last_cols = lower ? rows : nt - (ns - 1) * S;
        const int last = lower ? rows * (rows + 1) / 2 : rows * last_cols;
        if (pos < full + last) {
            if (pos < full) {
                const int sj = pos / (rows * S), rem = pos - sj * rows * S;
                ti = si * S + rem / S;
                tj = sj * S + rem % S;
            } else {
                pos -= full;
                if (lower) {
                    int r = 0;
                    while ((r + 1) * (r + 2) / 2 <= pos) ++r;
                    ti = si * S + r;
                    tj = si * S + pos - r * (r + 1) / 2;
                } else {
                    ti = si * S + pos / last_cols;
                    tj = n_full * S + pos % last_cols;
                }
            }
            return;
        }
        pos -= full + last;
    }
}

// One operand tile (BKT tokens x 128 columns of 16-bit) into LDS by LDS-DMA: BKT/4 wave-instructions
// of 1 KiB (= 4 token rows), BKT/16 per wave.  LDS position P (bytes) holds
//   token  k = P / 256,   32-byte chunk (P % 256) / 32  ^  2 * (k & 3),   half (P / 16) & 1.
template <int BKT>
__device__ __forceinline__ void stage_operand(const unsigned short *__restrict__ x, int64_t ld,
                                              int64_t tok0, int col0, char *lds_op, int wave, int lane)
{
    const int krow = lane >> 4;             // 0..3 == k & 3
    const int s = lane & 15;                // 16-byte slot in the 256-byte row
    const int chunk = (s >> 1) ^ (2 * krow);
    const int coff = chunk * 16 + (s & 1) * 8;
#pragma unroll
    for (int it = 0; it < BKT / 16; ++it) {
        const int ii = wave * (BKT / 16) + it;       // wave-uniform instruction index
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

template <int BASE, int OPB>
__device__ __forceinline__ void read_step(Frags &f, const unsigned (&addrA)[2], const unsigned (&addrB)[2])
{
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        tr_read<BASE>(f.a[u][0], addrA[u]);
        tr_read<BASE + 4 * 256>(f.a[u][1], addrA[u]);
        tr_read<BASE + OPB>(f.b[u][0], addrB[u]);
        tr_read<BASE + OPB + 4 * 256>(f.b[u][1], addrB[u]);
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

// BKT tokens per k-step, NST LDS stages: the DMA of step k + NST - 1 is issued right after the
// barrier of step k (its buffer was read during step k - 1), so NST - 1 steps of loads are in
// flight under the MFMAs; each wave waits only for its own loads of the step it is about to read
// (counted vmcnt), then the barrier makes every wave's share visible.
template <int DTYPE, int BKT, int NST>
__global__ __launch_bounds__(256) void xtx_mfma_kernel(float *__restrict__ hess, const unsigned short *__restrict__ x,
                                                       int hidden, int n_ktiles, uint32_t tile_order,
                                                       int n_blocks, int mirror)
{
    constexpr int OP_BYTES = BKT * TM * 2;
    constexpr int STAGE_BYTES = 2 * OP_BYTES;
    constexpr int LPS = 2 * (BKT / 16);   // LDS-DMA instructions per wave per step
    constexpr int KK = BKT / 16;          // MFMA k-steps per stage
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;

    // XCD-aware remap (bijective for any grid size): blocks with equal blockIdx % 8 share an XCD
    // and get a contiguous range of the tile order, which lists the tiles super-tile by super-tile
    // (see xtx_tile_of): the ~100 tiles an XCD has in flight then touch ~20 column panels of
    // X instead of all of them, and X streams from HBM once per super-tile instead of once per row.
    int bid;
    {
        const int orig = blockIdx.x;
        const int xcd = orig & 7;
        const int q = n_blocks >> 3, r = n_blocks & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    int ti, tj;
    xtx_tile_of(bid, tile_order, ti, tj);
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

    auto issue = [&](int kt, char *buf) {
        stage_operand<BKT>(x, ld, (int64_t)kt * BKT, m0, buf, wave, lane);
        stage_operand<BKT>(x, ld, (int64_t)kt * BKT, n0, buf + OP_BYTES, wave, lane);
    };
#pragma unroll
    for (int st = 0; st < NST - 1; ++st)
        if (st < n_ktiles) issue(st, smem + st * STAGE_BYTES);

    // All LDS offsets are immediates: the k-loop is unrolled NST times with the stage a constant.
    auto step = [&](auto cur_c, int kt) {
        constexpr int cur = decltype(cur_c)::value;
        constexpr int SB = cur * STAGE_BYTES;
        if (n_ktiles - 1 - kt >= NST - 2) {
            if constexpr (NST == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if constexpr ((NST - 2) * LPS == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if constexpr ((NST - 2) * LPS == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if constexpr ((NST - 2) * LPS == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if constexpr ((NST - 2) * LPS == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + NST - 1 < n_ktiles) issue(kt + NST - 1, smem + ((cur + NST - 1) % NST) * STAGE_BYTES);
        // stage base goes into the address registers (ds offsets are 16-bit)
        const unsigned sA[2] = {addrA[0] + SB, addrA[1] + SB}, sB[2] = {addrB[0] + SB, addrB[1] + SB};
        Frags f0, f1;
        read_step<0, OP_BYTES>(f0, sA, sB);
        if constexpr (KK == 2) {
            read_step<4096, OP_BYTES>(f1, sA, sB);
            asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            mfma_step<DTYPE>(acc, f0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            mfma_step<DTYPE>(acc, f1);
        } else {
            static_assert(KK == 4 || KK == 2, "32 or 64 tokens per step");
            read_step<4096, OP_BYTES>(f1, sA, sB);
            asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            mfma_step<DTYPE>(acc, f0);
            __builtin_amdgcn_sched_barrier(0);
            read_step<2 * 4096, OP_BYTES>(f0, sA, sB);
            asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            mfma_step<DTYPE>(acc, f1);
            __builtin_amdgcn_sched_barrier(0);
            read_step<3 * 4096, OP_BYTES>(f1, sA, sB);
            asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            mfma_step<DTYPE>(acc, f0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            mfma_step<DTYPE>(acc, f1);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    int kt = 0;
    for (; kt + NST <= n_ktiles; kt += NST) {
        step(std::integral_constant<int, 0>{}, kt);
        step(std::integral_constant<int, 1 % NST>{}, kt + 1);
        if constexpr (NST > 2) step(std::integral_constant<int, 2 % NST>{}, kt + 2);
        if constexpr (NST > 3) step(std::integral_constant<int, 3 % NST>{}, kt + 3);
    }
    // remainder: fewer than NST steps left, stages continue from 0
    if (kt < n_ktiles) step(std::integral_constant<int, 0>{}, kt++);
    if (NST > 2 && kt < n_ktiles) step(std::integral_constant<int, 1 % NST>{}, kt++);
    if (NST > 3 && kt < n_ktiles) step(std::integral_constant<int, 2 % NST>{}, kt++);

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

// ---- 256 x 256 tiles -------------------------------------------------------------------------
// Same construction with half the L2 -> LDS bytes per flop: 512 threads, 8 waves as 2 (rows) x 4
// (columns), each wave a 128 x 64 sub-tile of 8 x 4 blocks on v_mfma_f32_16x16x32 (one MFMA spans the
// whole 32-token step, so an accumulator is read and written once per 32 tokens: 3 % faster than the
// same tile on 32 x 32 x 16 blocks, measured at both widths), 128 accumulator registers, 32-token
// steps, 4 LDS stages of 32 KiB (one workgroup per CU, three steps of DMA in flight).
// LDS rows are 512 bytes (256 columns); the 32-byte chunk index is XOR-ed with 2 * (k & 3) as in
// the 128-column image (only the low three bits matter for the banks).
//
// Schedule.  A step is two segments separated by workgroup barriers: R (the 24 transposing LDS
// reads of the step) and M (its 32 MFMAs).  Waves 4..7 run one segment behind waves 0..3 (one
// extra barrier in front of their loop, waves 0..3 take theirs at the end); waves w and w + 4
// share a SIMD, so one of them is always in M while the other is in R: the matrix pipe has work
// all the time and the partner's LDS traffic runs beside it.  The LDS-DMA loads of step t + 3 are
// issued from inside M(t), one between every eight MFMAs: there the wave only waits for the matrix
// pipe, so issuing costs nothing and the address unit sees an even trickle (issued at the head of
// a read segment the same loads delay the reads behind them by hundreds of cycles).
// Ordering: stage (t+3) % 4 held step t-1, whose last reads (R(t-1) of waves 4..7) ended before a
// barrier every wave has passed when it enters its M(t).  Every wave ends each R(s) with
// vmcnt(4): all its DMA but the 4 youngest (step s+2) have landed, i.e. its share of step s+1 is in
// LDS one barrier before waves 0..3 read it (their R(s+1)) and two before waves 4..7 do.
constexpr int T2 = 256;
constexpr int BK2 = 32;
constexpr int NST2 = 4;
constexpr int OP2_BYTES = BK2 * T2 * 2;        // 16 KiB
constexpr int STAGE2_BYTES = 2 * OP2_BYTES;    // 32 KiB
constexpr int XTX2_LDS = NST2 * STAGE2_BYTES;  // 128 KiB

// wave-instruction `ii` (0..15) of one operand: 1 KiB = 2 token rows of 512 bytes
__device__ __forceinline__ void stage_one2(const unsigned short *__restrict__ x, int64_t ld, int64_t tok0, int col0,
                                           char *lds_op, int ii, int lane)
{
    const int s = lane & 31;   // 16-byte slot in the row
    const int k = ii * 2 + (lane >> 5);
    const int chunk = (s >> 1) ^ (2 * (k & 3));
    const unsigned short *src = x + (tok0 + k) * ld + col0 + chunk * 16 + (s & 1) * 8;
    __builtin_amdgcn_global_load_lds(GQ_GLOBAL_PTR(src), GQ_LDS_PTR(void, lds_op + ii * 1024), 16, 0, 0);
}

#ifdef GQ_XTX_TRACE
// Debug build only (tools/trace_xtx.py): per-wave cycle stamps of one workgroup.
__device__ unsigned long long *g_xtx_trace = nullptr;
#define XT_TR(i) ts[i] = __builtin_readcyclecounter()
#else
#define XT_TR(i)
#endif

template <int DTYPE>
__global__ __launch_bounds__(512) void xtx256_kernel(float *__restrict__ hess, const unsigned short *__restrict__ x,
                                                     int hidden, int n_ktiles_all, uint32_t tile_order,
                                                     int n_blocks, int mirror, int n_tiles, int n_splits, int64_t part_stride,
                                                     int n_full, int tail_splits, float *__restrict__ tail_buf)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef GQ_XTX_TRACE
    unsigned long long ts[12] = {};
    XT_TR(0);
    const unsigned long long tr0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    int bid;
    {
        // workgroups go to the XCDs round-robin; each XCD takes one contiguous stretch of the order.  With a tail
        // split the whole tiles and the tail runs are two such ranges (n_full is a multiple of 8), so that every
        // XCD gets its share of the short workgroups
        const int base = (tail_splits > 0 && (int)blockIdx.x >= n_full) ? n_full : 0;
        const int count = tail_splits > 0 ? (base ? n_blocks - n_full : n_full) : n_blocks;
        const int orig = blockIdx.x - base;
        const int xcd = orig & 7;
        const int q = count >> 3, r = count & 7;
        bid = base + (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    // token split: workgroup (tile, split) adds the products of its share of the k-steps into partial
    // Hessian `split` (part_stride floats apart); the caller sums the parts.
    // Tail split (tail_splits > 0, n_splits == 1): the first n_full tiles of the order -- whole rounds of one
    // workgroup per CU -- are taken whole; the tiles behind them, which would fill only part of a last round, are
    // cut into tail_splits token runs each, workgroups of one run next to each other, and accumulate into tile-sized
    // buffers of their own (tail_buf[run][tile]); gq_xtx_fold_tail adds those into H.
    int split, n_sp, tile_idx;
    const bool tail = tail_splits > 0 && bid >= n_full;
    if (tail) {
        const int n_tail = n_tiles - n_full, idx = bid - n_full;
        split = idx / n_tail;
        tile_idx = n_full + (idx - split * n_tail);
        n_sp = tail_splits;
    } else {
        split = bid / n_tiles;
        tile_idx = bid - split * n_tiles;
        n_sp = n_splits;
    }
    int ti, tj;
    xtx_tile_of(tile_idx, tile_order, ti, tj);
    const int m0 = ti * T2, n0 = tj * T2;
    const int64_t ld = hidden;
    const int k_begin = (int)((int64_t)n_ktiles_all * split / n_sp);
    const int n_ktiles = (int)((int64_t)n_ktiles_all * (split + 1) / n_sp) - k_begin;
    if (n_ktiles == 0) return;   // uniform over the workgroup
    x += (int64_t)k_begin * BK2 * ld;
    // where the tile's sums go: into H at (m0, n0), or into the tail buffer's own 256 x 256 piece
    float *__restrict__ out = tail ? tail_buf + ((int64_t)split * (n_tiles - n_full) + (tile_idx - n_full)) * (T2 * T2)
                                   : hess + (int64_t)split * part_stride;
    const int64_t out_ld = tail ? T2 : ld;
    const int om0 = tail ? 0 : m0, on0 = tail ? 0 : n0;

    // 16 x 16 x 32 blocks: the 16 lanes of group g hold the 16 columns of a block for tokens 8 g .. 8 g + 7
    // of the 32-token step (two transposing reads of 4 tokens each)
    const int wr = wave >> 2, wc = wave & 3;
    const int g = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, p4 = i16 & 3;
    const unsigned lds_base = (unsigned)(uintptr_t)GQ_LDS_PTR(char, smem);
    unsigned addrA[8], addrB[4];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int ca = (wr * 8 + u) ^ (2 * q4);
        addrA[u] = lds_base + (8 * g + q4) * 512 + ca * 32 + p4 * 8;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int cb = (wc * 4 + u) ^ (2 * q4);
        addrB[u] = lds_base + (8 * g + q4) * 512 + cb * 32 + p4 * 8;
    }

    f32x4 acc[8][4];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // this wave's 4 DMA instructions of step kt: 2 of the row-panel operand, 2 of the column-panel one
    auto issue_one = [&](int kt, char *buf, int i) {
        const int ii = wave * 2 + (i & 1);
        if (i < 2) stage_one2(x, ld, (int64_t)kt * BK2, m0, buf, ii, lane);
        else stage_one2(x, ld, (int64_t)kt * BK2, n0, buf + OP2_BYTES, ii, lane);
    };
#pragma unroll
    for (int st = 0; st < NST2 - 1; ++st)
        if (st < n_ktiles) {
#pragma unroll
            for (int i = 0; i < 4; ++i) issue_one(st, smem + st * STAGE2_BYTES, i);
        }
    // step 0 in LDS for everyone before the first read
    if (n_ktiles >= 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (n_ktiles == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const bool late = wave >= 4;   // wave-uniform
    if (late) __builtin_amdgcn_s_barrier();

    auto seg_barrier = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    auto step = [&](auto cur_c, int kt) {
        constexpr int cur = decltype(cur_c)::value;
        constexpr int SB = cur * STAGE2_BYTES;
        // ---- R
        // stage offsets beyond the 16-bit immediate of the read go into the address
        constexpr int SB_HI = SB & ~0xFFFF, SB_LO = SB & 0xFFFF;
        s16x4 fa[8][2], fb[4][2];
        XT_TR(2);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            tr_read<SB_LO>(fa[u][0], addrA[u] + SB_HI);
            tr_read<SB_LO + 4 * 512>(fa[u][1], addrA[u] + SB_HI);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            tr_read<SB_LO + OP2_BYTES>(fb[u][0], addrB[u] + SB_HI);
            tr_read<SB_LO + OP2_BYTES + 4 * 512>(fb[u][1], addrB[u] + SB_HI);
        }
        XT_TR(3);
        if (kt + 2 < n_ktiles) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        XT_TR(4);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        XT_TR(5);
        seg_barrier();
        XT_TR(6);
        // ---- M
        const bool more = kt + NST2 - 1 < n_ktiles;
        char *nbuf = smem + ((cur + NST2 - 1) % NST2) * STAGE2_BYTES;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const s16x8 va = __builtin_shufflevector(fa[a][0], fa[a][1], 0, 1, 2, 3, 4, 5, 6, 7);
                const s16x8 vb = __builtin_shufflevector(fb[b][0], fb[b][1], 0, 1, 2, 3, 4, 5, 6, 7);
                if constexpr (DTYPE == GQ_F16)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, va), __builtin_bit_cast(f16x8, vb), acc[a][b], 0, 0, 0);
                else
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, va), __builtin_bit_cast(bf16x8, vb), acc[a][b], 0, 0, 0);
                if (b == 3 && (a & 1) == 0) {   // behind MFMA 4, 12, 20, 28: one DMA instruction each
                    __builtin_amdgcn_sched_barrier(0);
                    if (more) issue_one(kt + NST2 - 1, nbuf, a >> 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        __builtin_amdgcn_s_setprio(0);
        XT_TR(7);
        seg_barrier();
        XT_TR(8);
    };
    XT_TR(1);
    int kt = 0;
    for (; kt + NST2 <= n_ktiles; kt += NST2) {
        step(std::integral_constant<int, 0>{}, kt);
        step(std::integral_constant<int, 1>{}, kt + 1);
        step(std::integral_constant<int, 2>{}, kt + 2);
        step(std::integral_constant<int, 3>{}, kt + 3);
    }
    if (kt < n_ktiles) step(std::integral_constant<int, 0>{}, kt++);
    if (kt < n_ktiles) step(std::integral_constant<int, 1>{}, kt++);
    if (kt < n_ktiles) step(std::integral_constant<int, 2>{}, kt++);
    if (!late) __builtin_amdgcn_s_barrier();
    XT_TR(9);

    // H[tile] += acc.  Register e of a lane is row 4 * (lane >> 4) + e, column lane & 15 of the 16 x 16
    // block, so in the mirrored tile the four registers are four consecutive floats of one row: one
    // 16-byte store.  The loads of four blocks are issued together.
    const bool do_mirror = mirror && (ti != tj) && !tail;   // (a tail tile's mirror image is written by the fold)
    const int lrow = 4 * (lane >> 4), lcol = lane & 15;
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        float old[4][4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const float *p0 = out + (int64_t)(om0 + wr * 128 + a * 16 + lrow) * out_ld + on0 + wc * 64 + b * 16 + lcol;
#pragma unroll
            for (int e = 0; e < 4; ++e) old[b][e] = p0[(int64_t)e * out_ld];
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int row0 = om0 + wr * 128 + a * 16 + lrow, col = on0 + wc * 64 + b * 16 + lcol;
            float *p0 = out + (int64_t)row0 * out_ld + col;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                old[b][e] += acc[a][b][e];
                p0[(int64_t)e * out_ld] = old[b][e];
            }
            if (do_mirror)
                *reinterpret_cast<f32x4 *>(out + (int64_t)col * out_ld + row0) = f32x4{old[b][0], old[b][1], old[b][2], old[b][3]};
        }
    }
#ifdef GQ_XTX_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    XT_TR(10);
    const unsigned long long tr1 = __builtin_amdgcn_s_memrealtime();
    if (g_xtx_trace && blockIdx.x == 40 && lane == 0) {
#pragma unroll
        for (int i = 0; i < 11; ++i) g_xtx_trace[wave * 16 + i] = ts[i];
        g_xtx_trace[wave * 16 + 11] = tr0;
        g_xtx_trace[wave * 16 + 12] = tr1;
    }
#endif
}

#ifdef GQ_XTX_TRACE
}  // namespace
extern "C" int gq_debug_xtx_trace(unsigned long long *buf)
{
    GQ_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_xtx_trace), &buf, sizeof(buf)));
    return GQ_OK;
}
namespace {
#endif

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

// super-tile edge of the tile order (experiment knob: GQ_XTX_SUPERTILE) and the number of tiles
int xtx_super_tile(int by_default)
{
    static const int forced = [] {
        const char *e = getenv("GQ_XTX_SUPERTILE");
        return e ? std::max(1, std::min(255, atoi(e))) : 0;
    }();
    return forced ? forced : by_default;
}

template <int DTYPE, int BKT, int NST>
int launch_mfma(float *hess, const unsigned short *x, int64_t n_tokens, int64_t hidden, int compute_lower_only,
                int mirror, hipStream_t stream, int64_t &done)
{
    constexpr int LDS = NST * 2 * BKT * TM * 2;
    const int nt = (int)(hidden / TM);
    const int n_ktiles = (int)(n_tokens / BKT);
    GQ_REQUIRE(nt < 65536, "gq_xtx_accumulate: hidden size too large");
    const int n_blocks = compute_lower_only ? nt * (nt + 1) / 2 : nt * nt;
    const uint32_t table = xtx_order(nt, xtx_super_tile(12), compute_lower_only);
    GQ_HIP(hipFuncSetAttribute((const void *)xtx_mfma_kernel<DTYPE, BKT, NST>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    hipLaunchKernelGGL((xtx_mfma_kernel<DTYPE, BKT, NST>), dim3(n_blocks), dim3(256), LDS, stream, hess, x, (int)hidden,
                       n_ktiles, table, n_blocks, mirror);
    GQ_LAUNCH_CHECK();
    done = (int64_t)n_ktiles * BKT;
    return GQ_OK;
}

// The tail split of a hidden size: tiles = n_full + n_tail, the n_tail tiles of the partly filled last round in
// `splits` token runs each (0: no tail split).  Measured: a launch costs its ROUNDS of one workgroup per CU, not its
// tiles (32768 tokens: hidden 9216 = 666 tiles 2.63 ms, 9728 = 741 tiles 2.57 ms, 11008 = 946 tiles 3.35 ms, 11264 =
// 990 tiles 3.40 ms: 0.84-0.87 ms per started round), so 178 tiles in a fourth round cost as much as 256; cut into 4
// runs they are 712 workgroups of a quarter of the length, three short rounds: 3.75 rounds instead of 4.
struct XtxTail { int n_full, n_tail, splits; };

XtxTail xtx_tail_plan(int64_t hidden, int compute_lower_only)
{
    static const int off = [] { const char *e = getenv("GQ_XTX_TAIL"); return e && e[0] == '0' ? 1 : 0; }();
    XtxTail none{0, 0, 0};
    if (off || hidden % T2 || hidden < 3584) return none;
    const int64_t nt = hidden / T2, tiles = compute_lower_only ? nt * (nt + 1) / 2 : nt * nt;
    if (tiles <= 256 || tiles % 256 == 0 || nt >= 65536) return none;
    const int n_tail = (int)(tiles % 256);
    int best = 0;
    double best_cost = 0.9;   // of a full round; every extra run costs one more epilogue per tile (~2 % of a tile)
    for (int sp = 2; sp <= 8; ++sp) {
        const double cost = (double)gq_cdiv((int64_t)n_tail * sp, 256) / sp + 0.02 * (sp - 1);
        if (cost < best_cost) {
            best = sp;
            best_cost = cost;
        }
    }
    // worth its buffers only where the last round is a visible share of the launch: 3 % of the rounds at least
    // (C = 28672: 24.7 rounds, predicted 0.8 %, measured 22.9 ms either way)
    if (!best || (1.0 - best_cost) < 0.03 * (double)gq_cdiv(tiles, 256)) return none;
    return XtxTail{(int)(tiles - n_tail), n_tail, best};
}

template <int DTYPE>
int launch_mfma256(float *hess, const unsigned short *x, int64_t n_tokens, int64_t hidden, int compute_lower_only,
                   int mirror, hipStream_t stream, int64_t &done, int n_splits = 1, int64_t part_stride = 0,
                   float *tail_buf = nullptr)
{
    const int nt = (int)(hidden / T2);
    const int n_ktiles = (int)(n_tokens / BK2);
    GQ_REQUIRE(nt < 65536, "gq_xtx_accumulate: hidden size too large");
    const int n_blocks = compute_lower_only ? nt * (nt + 1) / 2 : nt * nt;
    if (tail_buf) {
        const XtxTail plan = xtx_tail_plan(hidden, compute_lower_only);
        GQ_REQUIRE(plan.splits > 0 && n_splits == 1, "gq_xtx_accumulate_tail: no tail split for hidden size %lld", (long long)hidden);
        // a short call (fewer than 64 k-steps per run) leaves the tail tiles whole: one run, still into the buffer
        const int runs = std::max(1, std::min(plan.splits, n_ktiles / 64));
        const uint32_t table = xtx_order(nt, xtx_super_tile(6), compute_lower_only);
        auto kern = xtx256_kernel<DTYPE>;
        const int grid = plan.n_full + plan.n_tail * runs;
        GQ_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, XTX2_LDS));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), XTX2_LDS, stream, hess, x, (int)hidden, n_ktiles, table, grid, mirror,
                           n_blocks, 1, (int64_t)0, plan.n_full, runs, tail_buf);
        GQ_LAUNCH_CHECK();
        done = (int64_t)n_ktiles * BK2;
        return GQ_OK;
    }
    // 6 x 6 super-tiles: the ~32 tiles an XCD runs at a time (one per CU) then span about 12 column
    // panels of X instead of 64, and the XCD's L2 serves the rest (measured: fewest memory-side bytes)
    const uint32_t table = xtx_order(nt, xtx_super_tile(6), compute_lower_only);
    auto kern = xtx256_kernel<DTYPE>;
    GQ_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, XTX2_LDS));
    hipLaunchKernelGGL(kern, dim3(n_blocks * n_splits), dim3(512), XTX2_LDS, stream, hess, x, (int)hidden, n_ktiles, table,
                       n_blocks * n_splits, mirror, n_blocks, n_splits, part_stride, n_blocks, 0, (float *)nullptr);
    GQ_LAUNCH_CHECK();
    done = (int64_t)n_ktiles * BK2;
    return GQ_OK;
}

// experiment knob: GQ_XTX_VARIANT = "<tokens per step>x<stages>" (128-tiles) or "256"; unset = by shape
int xtx_variant()
{
    static int v = -1;
    if (v == -1) {
        const char *e = getenv("GQ_XTX_VARIANT");
        v = -2;   // -2: not forced, choose by shape
        if (e) {
            if (!strcmp(e, "64x2")) v = 0;
            else if (!strcmp(e, "32x4")) v = 1;
            else if (!strcmp(e, "32x3")) v = 2;
            else if (!strcmp(e, "64x3")) v = 3;
            else if (!strcmp(e, "32x2")) v = 4;
            else if (!strcmp(e, "256")) v = 5;
        }
    }
    return v;
}

template <int DTYPE>
int launch_xtx(float *hess, const unsigned short *x, int64_t n_tokens, int64_t hidden, int save_lower_only,
               int compute_lower_only, hipStream_t stream)
{
    const int mirror = compute_lower_only && !save_lower_only;
    int64_t done = 0;
    if (hidden % TM == 0 && n_tokens >= 64) {
        int rc;
        int variant = xtx_variant();
        if (variant < 0) {
            // 256 x 256 tiles at one workgroup per CU, or 128 x 128 tiles at two.  Measured per 32768-token call
            // (tools/bench_xtx_variants.py, ms, 256-tiles | 128-tiles): hidden 768..3072 0.61-0.64 | 0.46-0.49 (at most
            // 78 large tiles: most CUs idle), 3584 0.70 | 0.74, 4096 0.70 | 0.82, 5120 0.89 | 1.55, 8192 2.36 | 2.84,
            // 10240 3.30 | 4.99, 13824 5.27 | 8.95, 28672 23.1 | 43.9.  (A round-count model weighted with round times
            // measured on an earlier kernel chose the small tiles at 8192-11520 and lost 17-35 % there.)
            variant = (hidden % T2 == 0 && hidden >= 3584) ? 5 : 4;
        }
        switch (variant) {
        case 0: rc = launch_mfma<DTYPE, 64, 2>(hess, x, n_tokens, hidden, compute_lower_only, mirror, stream, done); break;
        case 1: rc = launch_mfma<DTYPE, 32, 4>(hess, x, n_tokens, hidden, compute_lower_only, mirror, stream, done); break;
        case 2: rc = launch_mfma<DTYPE, 32, 3>(hess, x, n_tokens, hidden, compute_lower_only, mirror, stream, done); break;
        case 3: rc = launch_mfma<DTYPE, 64, 3>(hess, x, n_tokens, hidden, compute_lower_only, mirror, stream, done); break;
        case 5:
            if (hidden % T2 == 0) {
                rc = launch_mfma256<DTYPE>(hess, x, n_tokens, hidden, compute_lower_only, mirror, stream, done);
                break;
            }
            [[fallthrough]];
        default: rc = launch_mfma<DTYPE, 32, 2>(hess, x, n_tokens, hidden, compute_lower_only, mirror, stream, done); break;
        }
        if (rc) return rc;
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
    GQ_REQUIRE(hessian && (x || n_tokens == 0), "gq_xtx_accumulate: NULL pointer");   // an empty batch has no data pointer
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

// ---- token split over partial Hessians ---------------------------------------------------------
// With few output tiles (C = 4096: 136 lower tiles of 256 x 256 on 256 CUs) one workgroup per tile
// leaves CUs idle.  The tokens of a call are then cut into n_parts runs and workgroup (tile, run)
// adds its products into partial Hessian `run`; the parts are summed in index order when H is
// needed (gq_xtx_fold), so the result does not depend on the order in which workgroups finish and
// stays exactly symmetric (every part is, and the sums of mirrored elements add the same numbers in
// the same order).

extern "C" int gq_xtx_split_hint(int64_t hidden, int64_t n_tokens, int compute_lower_only)
{
    if (hidden <= 0 || hidden % T2 || n_tokens < 64 * BK2) return 1;
    const int64_t nt = hidden / T2, tiles = compute_lower_only ? nt * (nt + 1) / 2 : nt * nt;
    // time in units of one full-token tile per CU: rounds of workgroups over 256 CUs, each 1 / parts long,
    // plus the read-modify-write of one more partial Hessian per extra part (about 5 % of a tile's time at
    // 32768 tokens per call, proportionally more for shorter calls)
    const double rmw = 0.05 * 32768.0 / (double)n_tokens * std::min(1.0, (double)tiles / 256.0);
    int best = 1;
    double best_cost = (double)gq_cdiv(tiles, 256);
    for (int parts = 2; parts <= 4; ++parts) {
        if (n_tokens / BK2 < 64 * parts) break;
        const double cost = (double)gq_cdiv(tiles * parts, 256) / parts + rmw * (parts - 1);
        if (cost < 0.9 * best_cost) {
            best = parts;
            best_cost = cost;
        }
    }
    return best;
}

extern "C" int gq_xtx_accumulate_split(float *parts, int n_parts, int64_t part_stride, const void *x, int dtype,
                                       int64_t n_tokens, int64_t hidden, int save_lower_only, int compute_lower_only,
                                       gq_stream_t stream)
{
    const char *who = "gq_xtx_accumulate_split";
    GQ_REQUIRE(parts && (x || n_tokens == 0), "%s: NULL pointer", who);
    GQ_REQUIRE(dtype == GQ_F16 || dtype == GQ_BF16, "%s: dtype %d is neither fp16 (0) nor bf16 (1)", who, dtype);
    GQ_REQUIRE(n_tokens >= 0 && hidden > 0 && hidden < (1 << 30), "%s: bad shape tokens=%lld hidden=%lld", who,
               (long long)n_tokens, (long long)hidden);
    GQ_REQUIRE(n_parts >= 1 && n_parts <= 16, "%s: n_parts must be 1..16, got %d", who, n_parts);
    GQ_REQUIRE(n_parts == 1 || (part_stride >= hidden * hidden && part_stride % 4 == 0),
               "%s: part_stride %lld must be a multiple of 4 floats and hold a %lld x %lld matrix", who, (long long)part_stride,
               (long long)hidden, (long long)hidden);
    GQ_REQUIRE(compute_lower_only || !save_lower_only, "compute_lower_only must be set when save_lower_only is set");
    GQ_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)parts & 15) == 0, "%s: pointers must be 16-byte aligned", who);
    if (n_tokens == 0) return GQ_OK;
    hipStream_t s = (hipStream_t)stream;
    const int64_t n_ktiles = n_tokens / BK2;
    if (n_parts == 1 || hidden % T2 || n_ktiles < n_parts) {   // nothing to split: everything into part 0
        if (dtype == GQ_F16) return launch_xtx<GQ_F16>(parts, (const unsigned short *)x, n_tokens, hidden, save_lower_only, compute_lower_only, s);
        return launch_xtx<GQ_BF16>(parts, (const unsigned short *)x, n_tokens, hidden, save_lower_only, compute_lower_only, s);
    }
    const int mirror = compute_lower_only && !save_lower_only;
    int64_t done = 0;
    int rc = dtype == GQ_F16
                 ? launch_mfma256<GQ_F16>(parts, (const unsigned short *)x, n_tokens, hidden, compute_lower_only, mirror, s, done, n_parts, part_stride)
                 : launch_mfma256<GQ_BF16>(parts, (const unsigned short *)x, n_tokens, hidden, compute_lower_only, mirror, s, done, n_parts, part_stride);
    if (rc) return rc;
    if (done < n_tokens) {   // the tokens that do not fill a k-step: generic kernel, into part 0
        const int nt = (int)gq_cdiv(hidden, 64);
        const int n_blocks = compute_lower_only ? nt * (nt + 1) / 2 : nt * nt;
        if (dtype == GQ_F16)
            hipLaunchKernelGGL(xtx_generic_kernel<GQ_F16>, dim3(n_blocks), dim3(256), 0, s, parts, (const unsigned short *)x, (int)hidden, done,
                               n_tokens, nt, compute_lower_only, mirror);
        else
            hipLaunchKernelGGL(xtx_generic_kernel<GQ_BF16>, dim3(n_blocks), dim3(256), 0, s, parts, (const unsigned short *)x, (int)hidden, done,
                               n_tokens, nt, compute_lower_only, mirror);
        GQ_LAUNCH_CHECK();
    }
    return GQ_OK;
}

namespace {
// parts[0][i] += parts[1][i] + ... (in index order); parts[1..] = 0
__global__ void xtx_fold_kernel(float *__restrict__ parts, int n_parts, int64_t part_stride, int64_t n4)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 *p0 = reinterpret_cast<f32x4 *>(parts) + i;
        f32x4 acc = *p0;
        for (int p = 1; p < n_parts; ++p) {
            f32x4 *pp = reinterpret_cast<f32x4 *>(parts + (int64_t)p * part_stride) + i;
            acc += *pp;
            *pp = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        *p0 = acc;
    }
}
}  // namespace

namespace {
// H[tile] += tail_buf[0][t] + tail_buf[1][t] + ... (in index order), the buffers cleared, and the tile's mirror image
// written from the result: one workgroup per tail tile, 64 x 64 pieces through LDS for the transposed write
__global__ __launch_bounds__(256) void xtx_fold_tail_kernel(float *__restrict__ hess, float *__restrict__ tail_buf, int hidden,
                                                            uint32_t tile_order, int n_full, int n_tail, int splits, int mirror)
{
    __shared__ float piece[64][65];
    const int t = blockIdx.x;
    int ti, tj;
    xtx_tile_of(n_full + t, tile_order, ti, tj);
    const int64_t ld = hidden;
    const int64_t m0 = (int64_t)ti * T2, n0 = (int64_t)tj * T2;
    const bool do_mirror = mirror && ti != tj;
    const int r = threadIdx.x >> 4, c4 = (threadIdx.x & 15) * 4;   // 16 rows x 16 float4 per turn
    for (int sr = 0; sr < 4; ++sr)
        for (int sc = 0; sc < 4; ++sc) {
            if (do_mirror) __syncthreads();
#pragma unroll
            for (int turn = 0; turn < 4; ++turn) {
                const int row = sr * 64 + turn * 16 + r, col = sc * 64 + c4;
                float *ph = hess + (m0 + row) * ld + n0 + col;
                f32x4 v = *reinterpret_cast<const f32x4 *>(ph);
                for (int p = 0; p < splits; ++p) {
                    f32x4 *pb = reinterpret_cast<f32x4 *>(tail_buf + ((int64_t)p * n_tail + t) * (T2 * T2) + row * T2 + col);
                    v += *pb;
                    *pb = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                *reinterpret_cast<f32x4 *>(ph) = v;
                if (do_mirror) {
                    piece[turn * 16 + r][c4 + 0] = v[0];
                    piece[turn * 16 + r][c4 + 1] = v[1];
                    piece[turn * 16 + r][c4 + 2] = v[2];
                    piece[turn * 16 + r][c4 + 3] = v[3];
                }
            }
            if (do_mirror) {
                __syncthreads();
#pragma unroll
                for (int turn = 0; turn < 4; ++turn) {
                    const int mr = turn * 16 + r;   // row of the mirrored piece = column of the piece
                    const f32x4 v = f32x4{piece[c4 + 0][mr], piece[c4 + 1][mr], piece[c4 + 2][mr], piece[c4 + 3][mr]};
                    *reinterpret_cast<f32x4 *>(hess + (n0 + sc * 64 + mr) * ld + m0 + sr * 64 + c4) = v;
                }
            }
        }
}
}  // namespace

extern "C" size_t gq_xtx_tail_workspace(int64_t hidden)
{
    const XtxTail plan = xtx_tail_plan(hidden, 1);
    return plan.splits ? (size_t)plan.splits * plan.n_tail * T2 * T2 * sizeof(float) : 0;
}

extern "C" int gq_xtx_accumulate_tail(float *hessian, float *tail_ws, const void *x, int dtype, int64_t n_tokens, int64_t hidden,
                                      gq_stream_t stream)
{
    const char *who = "gq_xtx_accumulate_tail";
    GQ_REQUIRE(hessian && tail_ws && (x || n_tokens == 0), "%s: NULL pointer", who);
    GQ_REQUIRE(dtype == GQ_F16 || dtype == GQ_BF16, "%s: dtype %d is neither fp16 (0) nor bf16 (1)", who, dtype);
    GQ_REQUIRE(n_tokens >= 0 && hidden > 0 && hidden < (1 << 30), "%s: bad shape", who);
    GQ_REQUIRE(gq_xtx_tail_workspace(hidden) > 0, "%s: hidden size %lld has no tail split (gq_xtx_tail_workspace is 0)", who, (long long)hidden);
    GQ_REQUIRE((((uintptr_t)x | (uintptr_t)hessian | (uintptr_t)tail_ws) & 15) == 0, "%s: pointers must be 16-byte aligned", who);
    if (n_tokens == 0) return GQ_OK;
    hipStream_t s = (hipStream_t)stream;
    int64_t done = 0;
    if (n_tokens >= BK2) {
        int rc = dtype == GQ_F16
                     ? launch_mfma256<GQ_F16>(hessian, (const unsigned short *)x, n_tokens, hidden, 1, 1, s, done, 1, 0, tail_ws)
                     : launch_mfma256<GQ_BF16>(hessian, (const unsigned short *)x, n_tokens, hidden, 1, 1, s, done, 1, 0, tail_ws);
        if (rc) return rc;
    }
    if (done < n_tokens) {   // the tokens that do not fill a k-step: generic kernel, straight into H
        const int nt = (int)gq_cdiv(hidden, 64);
        const int n_blocks = nt * (nt + 1) / 2;
        if (dtype == GQ_F16)
            hipLaunchKernelGGL(xtx_generic_kernel<GQ_F16>, dim3(n_blocks), dim3(256), 0, s, hessian, (const unsigned short *)x, (int)hidden, done,
                               n_tokens, nt, 1, 1);
        else
            hipLaunchKernelGGL(xtx_generic_kernel<GQ_BF16>, dim3(n_blocks), dim3(256), 0, s, hessian, (const unsigned short *)x, (int)hidden, done,
                               n_tokens, nt, 1, 1);
        GQ_LAUNCH_CHECK();
    }
    return GQ_OK;
}

extern "C" int gq_xtx_fold_tail(float *hessian, float *tail_ws, int64_t hidden, gq_stream_t stream)
{
    GQ_REQUIRE(hessian && tail_ws && hidden > 0, "gq_xtx_fold_tail: bad arguments");
    const XtxTail plan = xtx_tail_plan(hidden, 1);
    GQ_REQUIRE(plan.splits > 0, "gq_xtx_fold_tail: hidden size %lld has no tail split", (long long)hidden);
    const int nt = (int)(hidden / T2);
    const uint32_t table = xtx_order(nt, xtx_super_tile(6), 1);
    hipLaunchKernelGGL(xtx_fold_tail_kernel, dim3(plan.n_tail), dim3(256), 0, (hipStream_t)stream, hessian, tail_ws, (int)hidden, table,
                       plan.n_full, plan.n_tail, plan.splits, 1);
    GQ_LAUNCH_CHECK();
    return GQ_OK;
}

extern "C" int gq_xtx_fold(float *parts, int n_parts, int64_t part_stride, int64_t hidden, gq_stream_t stream)
{
    GQ_REQUIRE(parts && n_parts >= 1 && n_parts <= 16 && hidden > 0, "gq_xtx_fold: bad arguments");
    GQ_REQUIRE(n_parts == 1 || (part_stride >= hidden * hidden && part_stride % 4 == 0), "gq_xtx_fold: bad part_stride");
    GQ_REQUIRE(((uintptr_t)parts & 15) == 0 && (hidden * hidden) % 4 == 0, "gq_xtx_fold: 16-byte alignment");
    if (n_parts == 1) return GQ_OK;
    const int64_t n4 = hidden * hidden / 4;
    hipLaunchKernelGGL(xtx_fold_kernel, dim3((unsigned)std::min<int64_t>(gq_cdiv(n4, 256), 256 * 16)), dim3(256), 0, (hipStream_t)stream,
                       parts, n_parts, part_stride, n4);
    GQ_LAUNCH_CHECK();
    return GQ_OK;
}
