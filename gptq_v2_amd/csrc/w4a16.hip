// y[M][R] = x[M][C] * W^T with W held as packed 4-bit values + per-group fp16 scale and uint8 zero
// (fills the slot of the dense fp16 nn.Linear the reference runs on a reconstructed weight,
// llama.py:172-177; the reference has no packed GEMM).
//
// Prefill-shaped (M >= 2048), so this is an MFMA-bound fp16 GEMM whose B operand is produced on
// the fly: a 128(n) x 64(k) tile of W is 4 KiB of packed nibbles, read with one 16-byte load per
// thread, expanded in registers to fp16 with the exact arithmetic of the reference's dequantise
// ((q - z) is formed exactly by the 0x6400 exponent trick, then one rounded fp16 multiply by the
// scale) and stored to LDS; x tiles go global -> LDS by LDS-DMA.  Both LDS images are
// [128 rows][64 halves] with the 16-byte chunk index XOR-ed with (row >> 1) & 7, which makes the
// ds_read_b128 fragment reads conflict-free (the swizzle is applied to the DMA's source address
// for x and to the ds_write address for W).  128x128 output tile, 2x2 waves of 64x64,
// v_mfma_f32_32x32x16_f16, two LDS stages.
#include <type_traits>

#include <cstdlib>

#include <atomic>

#include "gq_common.h"

namespace {

constexpr int WT = 128;   // tile edge (both M and N)
constexpr int WK = 64;    // k-step
constexpr int W_OP_BYTES = WT * WK * 2;       // 16 KiB per operand image
constexpr int W_STAGE_BYTES = 2 * W_OP_BYTES;
constexpr int W_LDS = 2 * W_STAGE_BYTES;       // 64 KiB

template <int OFF>
__device__ __forceinline__ void lds_read_b128(u32x4 &dst, unsigned lds_addr)
{
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(lds_addr), "i"(OFF));
}

// x tile (128 rows x 64 halves) by LDS-DMA: 16 wave-instructions of 1 KiB (8 rows), 4 per wave.
// LDS slot (row, s) holds logical chunk s ^ ((row >> 1) & 7).
__device__ __forceinline__ void stage_x(const unsigned short *__restrict__ x, int64_t ld, int64_t m_total, int64_t m0,
                                        int64_t k0, char *lds_a, int wave, int lane)
{
    const int rloc = lane >> 3, s = lane & 7;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int ii = wave * 4 + it;
        const int row = ii * 8 + rloc;
        const int chunk = s ^ ((row >> 1) & 7);
        int64_t m = m0 + row;
        m = m < m_total ? m : m_total - 1;   // rows past the end shadow the last row; never stored
        const unsigned short *src = x + m * ld + k0 + chunk * 8;
        __builtin_amdgcn_global_load_lds(GQ_GLOBAL_PTR(src), GQ_LDS_PTR(void, lds_a + ii * 1024), 16, 0, 0);
    }
}

struct WRegs {
    u32x4 q;          // 32 packed values of one W row
    unsigned s_raw;   // scale (fp16 bits) as loaded
    unsigned z_raw;   // zero point (0..255) as loaded; converted where it is used, not where it is
};                    // loaded, so that the load's latency is not waited for early

// Packed operand layout (GQ_PACK_W4_KBLOCK): qw[C/64][R][8 words] -- the 64 values of one row's
// k-step are 32 contiguous bytes and the rows of a k-step follow each other, so the 128 or 256 rows
// of a tile are one contiguous 4 or 8 KiB piece per k-step; scale and zero are [G][R], so a group's
// entries for the rows of a tile are contiguous too.
//
// Per-thread read position: the 32 values (half a k-step) of one W row, and the scale / zero point
// of the group they belong to (a group is a multiple of 32 values, so the 32 values never straddle
// two groups).  Advancing by one k-step is a few adds and compares; no division in the main loop.
struct WCursor {
    const uint32_t *q;
    const unsigned short *s;
    const uint8_t *z;
    int rem;          // offset of the thread's 32 values inside their group
    int group_size;
    int64_t r_total;
};

__device__ __forceinline__ WCursor w_cursor(const uint32_t *__restrict__ qw, const unsigned short *__restrict__ scale,
                                            const uint8_t *__restrict__ zero, int64_t r_total, int group_size, int64_t n0,
                                            int tid)
{
    int64_t n = n0 + (tid >> 1);
    n = n < r_total ? n : r_total - 1;   // rows past the end shadow the last row; never stored
    const int half = tid & 1;
    WCursor c;
    c.q = qw + n * 8 + half * 4;
    const int g = (half * 32) / group_size;   // 0, or 1 when the group is 32 wide
    c.s = scale + g * r_total + n;
    c.z = zero + g * r_total + n;
    c.rem = half * 32 - g * group_size;
    c.group_size = group_size;
    c.r_total = r_total;
    return c;
}

// loads the thread's share of the current k-step and moves the cursor to the next one
__device__ __forceinline__ void load_w(WRegs &w, WCursor &c)
{
    w.q = *reinterpret_cast<const u32x4 *>(c.q);
    w.s_raw = *c.s;
    w.z_raw = *c.z;
    c.q += c.r_total * 8;
    c.rem += WK;
#pragma unroll
    for (int i = 0; i < 2; ++i)   // WK = 64 <= 2 * group_size
        if (c.rem >= c.group_size) {
            c.rem -= c.group_size;
            c.s += c.r_total;
            c.z += c.r_total;
        }
}

__device__ __forceinline__ void w_scale_zero(const WRegs &w, f16x2 &s2, f16x2 &z2)
{
    const _Float16 s = __builtin_bit_cast(_Float16, (unsigned short)w.s_raw);
    const _Float16 zb = (_Float16)(1024.0f + (float)w.z_raw);   // exact: <= 1279
    s2 = f16x2{s, s};
    z2 = f16x2{zb, zb};
}

// expand 32 nibbles to fp16 and store them as 4 chunks of the W image
__device__ __forceinline__ void store_w(const WRegs &w, char *lds_b, int tid)
{
    const int row = tid >> 1, half = tid & 1;
    f16x2 s2, z2;
    w_scale_zero(w, s2, z2);
#pragma unroll
    for (int wi = 0; wi < 4; ++wi) {
        const unsigned word = w.q[wi];
        u32x4 out;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const unsigned bits = ((word >> (4 * p)) & 0x000F000Fu) | 0x64006400u;   // (1024 + v0, 1024 + v1)
            f16x2 v = __builtin_bit_cast(f16x2, bits);
            v = v - z2;   // exact: q - z
            v = v * s2;   // one rounding, as the reference's (q - z) * scale in fp16
            out[p] = __builtin_bit_cast(unsigned, v);
        }
        const int chunk = (half * 4 + wi) ^ ((row >> 1) & 7);
        *reinterpret_cast<u32x4 *>(lds_b + row * 128 + chunk * 16) = out;
    }
}

// bf16: q as fp32 (byte conversions of the nibble-split word), one fma per value with -z * s as the
// addend -- q * s and z * s are exact in fp32 (4 or 8 bits times 8 bits) and so is their difference
// (q - z) * s, which is then rounded to bf16 once, the rounding of the reference's
// (int16 q - int16 z) * bf16 scale.  Two values per v_pk_fma_f32.
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct DeqConstB {
    f32x2 s2;
    f32x2 nzs2;
};

template <class W>
__device__ __forceinline__ DeqConstB deq_const_bf16(const W &w)
{
    DeqConstB d;
    const float s = __uint_as_float(w.s_raw << 16);
    const float nzs = -(float)w.z_raw * s;   // exact
    d.s2 = f32x2{s, s};
    d.nzs2 = f32x2{nzs, nzs};
    return d;
}

__device__ __forceinline__ u32x4 expand_word(unsigned word, const DeqConstB &d, unsigned, unsigned)
{
    const unsigned even = word & 0x0F0F0F0Fu;          // nibbles 0, 2, 4, 6 in bytes 0..3
    const unsigned odd = (word >> 4) & 0x0F0F0F0Fu;    // nibbles 1, 3, 5, 7
    u32x4 out;
    // out[p] holds the values of nibbles p and p + 4
    auto pair = [&](unsigned src, int byte_lo) {
        f32x2 q{(float)((src >> (8 * byte_lo)) & 0xffu), (float)((src >> (8 * (byte_lo + 2))) & 0xffu)};
        const f32x2 v = __builtin_elementwise_fma(q, d.s2, d.nzs2);
        const unsigned lo = __builtin_bit_cast(unsigned short, (__bf16)v[0]), hi = __builtin_bit_cast(unsigned short, (__bf16)v[1]);
        return lo | (hi << 16);
    };
    out[0] = pair(even, 0);
    out[1] = pair(odd, 0);
    out[2] = pair(even, 1);
    out[3] = pair(odd, 1);
    return out;
}

// bf16 form of store_w
__device__ __forceinline__ void store_w_bf16(const WRegs &w, char *lds_b, int tid)
{
    const int row = tid >> 1, half = tid & 1;
    const DeqConstB d = deq_const_bf16(w);
#pragma unroll
    for (int wi = 0; wi < 4; ++wi) {
        const u32x4 out = expand_word(w.q[wi], d, 0u, 0u);
        const int chunk = (half * 4 + wi) ^ ((row >> 1) & 7);
        *reinterpret_cast<u32x4 *>(lds_b + row * 128 + chunk * 16) = out;
    }
}

// DT: GQ_F16 or GQ_BF16 -- the type of x, y and the scales, and of the MFMA operands.
template <int DT>
__global__ __launch_bounds__(256) void w4a16_kernel(unsigned short *__restrict__ y, const unsigned short *__restrict__ x,
                                                    const uint32_t *__restrict__ qw, const unsigned short *__restrict__ scale,
                                                    const uint8_t *__restrict__ zero, int64_t m_total, int64_t r_total,
                                                    int64_t c_total, int group_size, int tiles_n, int n_blocks)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    // XCD-aware, bijective remap: blocks sharing an XCD walk consecutive tiles (same x row panel)
    int bid;
    {
        const int orig = blockIdx.x, xcd = orig & 7;
        const int q = n_blocks >> 3, r = n_blocks & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int64_t m0 = (int64_t)(bid / tiles_n) * WT, n0 = (int64_t)(bid % tiles_n) * WT;
    const int n_ktiles = (int)(c_total / WK);

    const int wr = wave >> 1, wc = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const unsigned lds_base = (unsigned)(uintptr_t)GQ_LDS_PTR(char, smem);
    // fragment address for k-step kk: row*128 + ((2*kk + lh) ^ ((row >> 1) & 7)) * 16; the XOR keeps
    // bit 0 of the chunk apart from kk only for lh, so precompute per kk (4 values) per block row
    unsigned addrA[2][4], addrB[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int ra = (wr * 2 + u) * 32 + li, rb = (wc * 2 + u) * 32 + li;
            addrA[u][kk] = lds_base + ra * 128 + (((2 * kk + lh) ^ ((ra >> 1) & 7)) << 4);
            addrB[u][kk] = lds_base + W_OP_BYTES + rb * 128 + (((2 * kk + lh) ^ ((rb >> 1) & 7)) << 4);
        }

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.0f;

    WRegs wregs;
    WCursor wcsr = w_cursor(qw, scale, zero, r_total, group_size, n0, tid);
    stage_x(x, c_total, m_total, m0, 0, smem, wave, lane);
    load_w(wregs, wcsr);
    auto expand = [&](char *lds_b) {
        if constexpr (DT == GQ_F16) store_w(wregs, lds_b, tid);
        else store_w_bf16(wregs, lds_b, tid);
    };
    expand(smem + W_OP_BYTES);

    auto step = [&](auto cur_c, int kt) {
        constexpr int cur = decltype(cur_c)::value;
        constexpr int SB = cur * W_STAGE_BYTES;
        __syncthreads();   // x tile of this stage landed (vmcnt drained), W tile written, other stage free
        const bool more = kt + 1 < n_ktiles;
        if (more) {
            stage_x(x, c_total, m_total, m0, (int64_t)(kt + 1) * WK, smem + (cur ^ 1) * W_STAGE_BYTES, wave, lane);
            load_w(wregs, wcsr);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            u32x4 fa[2], fb[2];
            lds_read_b128<SB>(fa[0], addrA[0][kk]);
            lds_read_b128<SB>(fa[1], addrA[1][kk]);
            lds_read_b128<SB>(fb[0], addrB[0][kk]);
            lds_read_b128<SB>(fb[1], addrB[1][kk]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    if constexpr (DT == GQ_F16)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[a]),
                                                                           __builtin_bit_cast(f16x8, fb[b]), acc[a][b], 0, 0, 0);
                    else
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[a]),
                                                                            __builtin_bit_cast(bf16x8, fb[b]), acc[a][b], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more) expand(smem + (cur ^ 1) * W_STAGE_BYTES + W_OP_BYTES);
    };
    int kt = 0;
    for (; kt + 1 < n_ktiles; kt += 2) {
        step(std::integral_constant<int, 0>{}, kt);
        step(std::integral_constant<int, 1>{}, kt + 1);
    }
    if (kt < n_ktiles) step(std::integral_constant<int, 0>{}, kt);

#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t row = m0 + (wr * 2 + a) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                const int64_t col = n0 + (wc * 2 + b) * 32 + li;
                if (row < m_total && col < r_total) {
                    if constexpr (DT == GQ_F16) y[row * r_total + col] = __builtin_bit_cast(unsigned short, (_Float16)acc[a][b][e]);
                    else y[row * r_total + col] = __builtin_bit_cast(unsigned short, (__bf16)acc[a][b][e]);
                }
            }
}

// ---- 256 x 256 tiles -------------------------------------------------------------------------
// Half the L2 -> LDS bytes per flop of the 128 x 128 kernel: 512 threads, 8 waves as 2 (M) x 4 (N),
// each wave a 128 x 64 sub-tile (4 x 2 MFMA blocks, 128 accumulator registers), 64-deep k-steps,
// two LDS stages of 64 KiB (x image 32 KiB + W image 32 KiB), one workgroup per CU.  Same images,
// swizzle and dequantisation as above; every thread expands the 32 nibbles of half a W row.
// The two waves of each SIMD run half a k-step apart (see the main loop).
// ---- helpers of the 256 x 256 kernel -----------------------------------------------------------
// Expansion constants of one thread's 32 values (one group): the two exponent-trick biases with
// the zero point folded in, and the scale, each in both halves of a packed pair.
struct DeqConst {
    f16x2 zb0;   // 1024 + z : bias of a nibble sitting in bits 0..3 of a half   (0x6400 | q)
    f16x2 zb1;   //   64 + z : bias of a nibble sitting in bits 4..7 of a half   (0x5400 | q << 4)
    f16x2 s2;
};

template <class W>
__device__ __forceinline__ DeqConst deq_const(const W &w)
{
    DeqConst d;
    const _Float16 s = __builtin_bit_cast(_Float16, (unsigned short)w.s_raw);
    const _Float16 zb = (_Float16)(1024.0f + (float)w.z_raw);   // exact: <= 1279
    d.s2 = f16x2{s, s};
    d.zb0 = f16x2{zb, zb};
    d.zb1 = d.zb0 - f16x2{(_Float16)960.0f, (_Float16)960.0f};  // exact
    return d;
}

template <int DT, class W>
__device__ __forceinline__ auto deq_const_dt(const W &w)
{
    if constexpr (DT == GQ_F16) return deq_const(w);
    else return deq_const_bf16(w);
}

// (x & mask) | magic in one instruction; the mask rides in an SGPR (one literal per VOP3 at most)
__device__ __forceinline__ unsigned and_or_b32(unsigned x, unsigned mask, unsigned magic)
{
    unsigned r;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "s"(mask), "v"(magic));
    return r;
}

// The 8 values of one packed word as 4 fp16 pairs (q - z) * s, exactly as the reference's
// dequantisation rounds them: nibbles in bits 0..3 of a half become 1024 + q, nibbles in bits 4..7
// become 64 + q without a shift, the upper byte of each half is handled the same way after one
// shift by 8.  13 VALU instructions per word.
__device__ __forceinline__ u32x4 expand_word(unsigned word, const DeqConst &d, unsigned magic0, unsigned magic1)
{
    const unsigned hi = word >> 8;
    f16x2 v0 = __builtin_bit_cast(f16x2, and_or_b32(word, 0x000F000Fu, magic0)) - d.zb0;
    f16x2 v1 = __builtin_bit_cast(f16x2, and_or_b32(word, 0x00F000F0u, magic1)) - d.zb1;
    f16x2 v2 = __builtin_bit_cast(f16x2, and_or_b32(hi, 0x000F000Fu, magic0)) - d.zb0;
    f16x2 v3 = __builtin_bit_cast(f16x2, and_or_b32(hi, 0x00F000F0u, magic1)) - d.zb1;
    u32x4 out;
    out[0] = __builtin_bit_cast(unsigned, v0 * d.s2);
    out[1] = __builtin_bit_cast(unsigned, v1 * d.s2);
    out[2] = __builtin_bit_cast(unsigned, v2 * d.s2);
    out[3] = __builtin_bit_cast(unsigned, v3 * d.s2);
    return out;
}

// Wave-uniform read position of the 256-row tile in the packed weights (group sizes that are
// multiples of the 64-deep k-step: all 64 values of a row's k-step share one group).  Lives in
// SGPRs; each thread adds its own constant offsets.
struct W2Cursor {
    const char *q;             // packed words of the tile's first row at the current k-step
    const unsigned short *s;   // scale of the tile's first row, current group
    const uint8_t *z;
    int rem;                   // k offset inside the current group
};

#ifdef GQ_W4_NO_PRIO
#define GQ_PRIO(p)
#else
#define GQ_PRIO(p) __builtin_amdgcn_s_setprio(p)
#endif
#ifdef GQ_W4_TRACE
// Debug build only (tools/trace_w4.py): per-wave cycle stamps of the last steady-state k-step.
__device__ unsigned long long *g_w4_trace = nullptr;
#define GQ_TR(i) if constexpr (MORE2) ts[i] = __builtin_readcyclecounter()
#else
#define GQ_TR(i)
#endif
constexpr int W2T = 256;
constexpr int W2_OP_BYTES = W2T * WK * 2;        // 32 KiB
constexpr int W2_STAGE_BYTES = 2 * W2_OP_BYTES;  // 64 KiB
constexpr int W2_LDS = 2 * W2_STAGE_BYTES;       // 128 KiB

// x tile (256 rows x 64 halves) by LDS-DMA: 32 wave-instructions of 1 KiB (8 rows each), issued by
// `n_waves` waves (`w` = index of this wave among them).
// one of the 32 wave-instructions of the x tile (ii = 0..31)
__device__ __forceinline__ void stage_x2_one(const unsigned short *__restrict__ x, int64_t ld, int64_t m_total, int64_t m0,
                                             int64_t k0, char *lds_a, int ii, int lane)
{
    const int rloc = lane >> 3, s = lane & 7;
    const int row = ii * 8 + rloc;
    const int chunk = s ^ ((row >> 1) & 7);
    int64_t m = m0 + row;
    m = m < m_total ? m : m_total - 1;
    const unsigned short *src = x + m * ld + k0 + chunk * 8;
    __builtin_amdgcn_global_load_lds(GQ_GLOBAL_PTR(src), GQ_LDS_PTR(void, lds_a + ii * 1024), 16, 0, 0);
}

template <int N_WAVES>
__device__ __forceinline__ void stage_x2(const unsigned short *__restrict__ x, int64_t ld, int64_t m_total, int64_t m0,
                                         int64_t k0, char *lds_a, int w, int lane)
{
    const int rloc = lane >> 3, s = lane & 7;
    constexpr int PER_WAVE = 32 / N_WAVES;
#pragma unroll
    for (int it = 0; it < PER_WAVE; ++it) {
        const int ii = w * PER_WAVE + it;
        const int row = ii * 8 + rloc;
        const int chunk = s ^ ((row >> 1) & 7);
        int64_t m = m0 + row;
        m = m < m_total ? m : m_total - 1;
        const unsigned short *src = x + m * ld + k0 + chunk * 8;
        __builtin_amdgcn_global_load_lds(GQ_GLOBAL_PTR(src), GQ_LDS_PTR(void, lds_a + ii * 1024), 16, 0, 0);
    }
}

// NB = 32-column MFMA blocks per wave: 2 is the 256 x 256 tile described above; 1 is a 256 (M) x 128 (N) tile
// with the same x image, schedule and waves (each 128 x 32), for shapes whose 256 x 256 tiles would leave CUs
// without a workgroup (M = 2048 against 4096 rows: 128 tiles on 256 CUs; 256 x 128 gives exactly 256).
template <int WORDS>
struct W2Regs {
    unsigned q __attribute__((ext_vector_type(WORDS)));   // WORDS * 8 packed values of one W row
    unsigned s_raw;
    unsigned z_raw;
};

template <int DT, int NB>
__global__ __launch_bounds__(512) void w4a16_256_kernel(unsigned short *__restrict__ y, const unsigned short *__restrict__ x,
                                                        const uint32_t *__restrict__ qw, const unsigned short *__restrict__ scale,
                                                        const uint8_t *__restrict__ zero, int64_t m_total, int64_t r_total,
                                                        int64_t c_total, int group_size, int tiles_n, int n_blocks, int band)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef GQ_W4_TRACE
    const unsigned long long tk0 = __builtin_readcyclecounter(), tr0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    int bid;
    {
        const int orig = blockIdx.x, xcd = orig & 7;
        const int q = n_blocks >> 3, r = n_blocks & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    // tiles in bands of `band` tile rows, column-major inside a band: the 32 workgroups an XCD runs
    // at a time cover band x (32 / band) tiles and share that many x panels and W tiles in its L2
    int tm, tn;
    {
        const int per_band = band * tiles_n;
        const int b = bid / per_band, within = bid - b * per_band;
        const int tiles_m = n_blocks / tiles_n;
        const int rows = min(band, tiles_m - b * band);
        tn = within / rows;
        tm = b * band + (within - tn * rows);
    }
    constexpr int TN = 128 * NB;          // tile width (rows of W)
    constexpr int WPT = 2 * NB;           // packed words a thread expands per k-step
    constexpr int TPR = 8 / WPT;          // threads per W row
    const int64_t m0 = (int64_t)tm * W2T, n0 = (int64_t)tn * TN;
    const int n_ktiles = (int)(c_total / WK);

    const int wr = wave >> 2, wc = wave & 3;
    const int li = lane & 31, lh = lane >> 5;
    const unsigned lds_base = (unsigned)(uintptr_t)GQ_LDS_PTR(char, smem);
    // row bases of the wave's MFMA blocks; the chunk of k-step kk is (2*kk + lh) ^ sw with
    // sw = (row >> 1) & 7 = (li >> 1) & 7 for every block (block offsets are multiples of 32 rows)
    const int sw = (li >> 1) & 7;
    unsigned rowA[4], rowB[NB], choff[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) rowA[u] = lds_base + ((wr * 4 + u) * 32 + li) * 128;
#pragma unroll
    for (int u = 0; u < NB; ++u) rowB[u] = lds_base + W2_OP_BYTES + ((wc * NB + u) * 32 + li) * 128;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) choff[kk] = (unsigned)(((2 * kk + lh) ^ sw) << 4);

    f32x16 acc[4][NB];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.0f;

    // Each k-step is four segments separated by workgroup barriers:
    //   R1 (LDS reads of k 0..31, next step's DMA issue, half of the next W tile's expansion),
    //   M1 (16 MFMAs), R2 (reads of k 32..63, other half of the expansion), M2 (16 MFMAs).
    // Waves 4..7 run one segment behind waves 0..3 (one extra barrier in front, waves 0..3 take it
    // at the end).  Waves w and w + 4 share a SIMD, so while one of them is in an MFMA segment the
    // other is in a read segment: the matrix pipe always has work and the LDS / VALU / DMA work of
    // the partner runs beside it.  Every LDS access of a segment is complete (lgkmcnt(0), and
    // vmcnt for this wave's DMA) before the barrier that ends it, so the barrier hands the data over.
    // W registers run two steps ahead: the packed words of step k+1 are in registers when step k
    // starts (loaded during step k-1).
    using WR = W2Regs<WPT>;
    WR wcur, wnext{};
    W2Cursor csr;
    csr.q = reinterpret_cast<const char *>(qw + n0 * 8);
    csr.s = scale + n0;
    csr.z = zero + n0;
    csr.rem = 0;
    unsigned voff_q, voff_g;   // this thread's row (clamped to the last one) and half inside the tile
    {
        int64_t nl = tid / TPR;
        nl = n0 + nl < r_total ? nl : r_total - 1 - n0;   // rows past the end shadow the last row; never stored
        voff_q = (unsigned)(nl * 32 + (tid % TPR) * (WPT * 4));
        voff_g = (unsigned)nl;
    }
    const int64_t kstep_bytes = r_total * 32;
    auto load_w2 = [&](WR &w) {
        // Issued as asm so that hipcc neither widens the 16- and 8-bit results right behind the loads
        // (which drags a vmcnt(0) to the next barrier) nor counts them: the consumer waits by hand.
        if constexpr (WPT == 4) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(w.q) : "v"(csr.q + voff_q) : "memory");
        else asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(w.q) : "v"(csr.q + voff_q) : "memory");
        asm volatile("global_load_ushort %0, %1, off" : "=v"(w.s_raw) : "v"(csr.s + voff_g) : "memory");
        asm volatile("global_load_ubyte %0, %1, off" : "=v"(w.z_raw) : "v"(csr.z + voff_g) : "memory");
        csr.q += kstep_bytes;
        csr.rem += WK;
        if (csr.rem >= group_size) {   // group_size is a multiple of WK here
            csr.rem = 0;
            csr.s += r_total;
            csr.z += r_total;
        }
    };
    unsigned magic0 = 0x64006400u, magic1 = 0x54005400u;
    asm volatile("" : "+v"(magic0), "+v"(magic1));   // keep them in VGPRs (operands of v_and_or_b32)
    const int wrow = tid / TPR, wpart = tid % TPR;
    using DQ = std::conditional_t<DT == GQ_F16, DeqConst, DeqConstB>;
    auto store_part = [&](const WR &w, const DQ &d, int wi, char *lds_b) {
        const u32x4 out = expand_word(w.q[wi], d, magic0, magic1);
        const int chunk = (wpart * WPT + wi) ^ ((wrow >> 1) & 7);
        *reinterpret_cast<u32x4 *>(lds_b + wrow * 128 + chunk * 16) = out;
    };
    const bool late = wave >= 4;   // wave-uniform (readfirstlane above)
    stage_x2<8>(x, c_total, m_total, m0, 0, smem, wave, lane);
    load_w2(wcur);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(wcur.q), "+v"(wcur.s_raw), "+v"(wcur.z_raw)::"memory");
    {
        const DQ d = deq_const_dt<DT>(wcur);
#pragma unroll
        for (int wi = 0; wi < WPT; ++wi) store_part(wcur, d, wi, smem + W2_OP_BYTES);
    }
    if (n_ktiles > 1) load_w2(wcur);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(wcur.q), "+v"(wcur.s_raw), "+v"(wcur.z_raw)::"memory");
    __syncthreads();
    // x tile of step 1: waited for (vmcnt(0)) by every wave before the last barrier of step 0
    if (n_ktiles > 1) stage_x2<8>(x, c_total, m_total, m0, WK, smem + W2_STAGE_BYTES, wave, lane);
    if (late) __builtin_amdgcn_s_barrier();

#ifdef GQ_W4_TRACE
    unsigned long long ts[13] = {};
#endif
    u32x4 fa[2][4], fb[2][NB];
    auto read_frags = [&](int slot, unsigned sb, int kk) {
#pragma unroll
        for (int u = 0; u < 4; ++u) lds_read_b128<0>(fa[slot][u], rowA[u] + sb + choff[kk]);
#pragma unroll
        for (int u = 0; u < NB; ++u) lds_read_b128<0>(fb[slot][u], rowB[u] + sb + choff[kk]);
    };
    // 8 MFMAs of one 16-deep k-chunk; `between(i)` runs after the MFMAs 1 and 5 of the chunk (one
    // global load per four MFMAs keeps the address unit far from saturation while the pipe is busy)
    auto mfmas = [&](int slot, auto between) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                if constexpr (DT == GQ_F16)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[slot][a]),
                                                                       __builtin_bit_cast(f16x8, fb[slot][b]), acc[a][b], 0, 0, 0);
                else
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[slot][a]),
                                                                        __builtin_bit_cast(bf16x8, fb[slot][b]), acc[a][b], 0, 0, 0);
                if (b == NB - 1 && (a == 0 || a == 2)) {
                    __builtin_amdgcn_sched_barrier(0);
                    between(a >> 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
    };
    auto seg_barrier = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };

    // MORE: a next step exists (expand its W tile from `wuse`);
    // MORE2: a step after that exists (fetch its packed W words and its x tile).
    //
    // Global loads are issued from inside the MFMA segments, one between every four MFMAs: there
    // the issuing wave only waits for the matrix pipe, so the issue costs nothing, and the loads
    // are spread evenly over time.  Issued at the head of a read segment they delay the LDS reads
    // and the expansion behind them by several hundred cycles; issued in a burst after the MFMAs
    // they hold up the barrier the partner is waiting at.
    //   x tile of step k+2 (it replaces the tile of step k in LDS, so every read of step k -- the
    //   R2 of both halves -- must lie before a barrier the issuing wave has passed):
    //     waves 4..7 issue rows 0..127 from their M2 of step k   (wait: end of their R2 of step k+1)
    //     waves 0..3 issue rows 128..255 from their M1 of step k+1 (wait: end of their M2 of step k+1)
    //   packed W words / scale / zero of step k+2: every wave, at the end of its M1 of step k,
    //   consumed (moved to `wuse`) at the end of its M2 of step k.
    // All of these loads are invisible to hipcc's wait insertion (DMA has no register result, the W
    // loads are asm), so every vmcnt below is counted by hand in program order per wave half:
    //   waves 0..3:  M1(k): [4 DMA step k+1 -> if k > 0] 3 W   | M2(k) end: vmcnt(0)
    //   waves 4..7:  M1(k): 3 W | M2(k): 4 DMA, then vmcnt(4) for the W words | R2(k+1) end: vmcnt(3)
    //                (the 3 W loads of M1(k+1) are younger than the DMA)
    auto body = [&](auto more_c, auto more2_c, auto first_c, int kt, WR &wuse, WR &wload) {
        constexpr bool MORE = decltype(more_c)::value, MORE2 = decltype(more2_c)::value;
        constexpr bool FIRST = decltype(first_c)::value;   // step 0: its successor's x tile was staged in the prologue
        const unsigned sb = (unsigned)(kt & 1) * W2_STAGE_BYTES;
        char *cur = smem + (kt & 1) * W2_STAGE_BYTES;
        char *nxt = smem + ((kt + 1) & 1) * W2_STAGE_BYTES;
        char *nb = nxt + W2_OP_BYTES;
        auto nothing = [](int) {};
        // ---- R1
        GQ_TR(0);
        read_frags(0, sb, 0);
        read_frags(1, sb, 1);
        GQ_TR(10);
        __builtin_amdgcn_sched_barrier(0);   // reads first: their latency runs under the expansion
        DQ dq;
        if constexpr (MORE) {
            dq = deq_const_dt<DT>(wuse);
#pragma unroll
            for (int wi = 0; wi < WPT / 2; ++wi) store_part(wuse, dq, wi, nb);
        }
        GQ_TR(11);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        GQ_TR(1);
        seg_barrier();
        // ---- M1
        GQ_TR(2);
        GQ_PRIO(1);
        if constexpr (MORE && !FIRST) {
            // waves 0..3: lower half of the x tile of step kt+1
            auto dma_lo = [&](int i) {
                if (!late) stage_x2_one(x, c_total, m_total, m0, (int64_t)(kt + 1) * WK, nxt, 16 + wave * 4 + i, lane);
            };
            auto dma_hi = [&](int i) {
                if (!late) stage_x2_one(x, c_total, m_total, m0, (int64_t)(kt + 1) * WK, nxt, 16 + wave * 4 + 2 + i, lane);
            };
            mfmas(0, dma_lo);
            mfmas(1, dma_hi);
        } else {
            mfmas(0, nothing);
            mfmas(1, nothing);
        }
        GQ_PRIO(0);
        GQ_TR(12);
        if constexpr (MORE2) {
            __builtin_amdgcn_sched_barrier(0);
            load_w2(wload);
        }
        GQ_TR(3);
        seg_barrier();
        // ---- R2
        GQ_TR(4);
        read_frags(0, sb, 2);
        read_frags(1, sb, 3);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MORE) {
#pragma unroll
            for (int wi = WPT / 2; wi < WPT; ++wi) store_part(wuse, dq, wi, nb);
        }
        // waves 4..7: their half of the next x tile (DMA from the previous M2; from the prologue in
        // step 0) must have landed before this barrier; the 3 younger W loads of M1 may stay in flight
        if constexpr (MORE) {
            if (late) {
                if constexpr (MORE2) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        GQ_TR(5);
        seg_barrier();
        // ---- M2
        GQ_TR(6);
        GQ_PRIO(1);
        if constexpr (MORE2) {
            // waves 4..7: upper half of the x tile of step kt+2
            auto dma_lo = [&](int i) {
                if (late) stage_x2_one(x, c_total, m_total, m0, (int64_t)(kt + 2) * WK, cur, (wave - 4) * 4 + i, lane);
            };
            auto dma_hi = [&](int i) {
                if (late) stage_x2_one(x, c_total, m_total, m0, (int64_t)(kt + 2) * WK, cur, (wave - 4) * 4 + 2 + i, lane);
            };
            mfmas(0, dma_lo);
            mfmas(1, dma_hi);
        } else {
            mfmas(0, nothing);
            mfmas(1, nothing);
        }
        GQ_PRIO(0);
        GQ_TR(7);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MORE2) {
            // W words of step kt+2 (loaded at the end of M1) move to `wuse`; waves 0..3 also collect
            // their DMA of M1 here, waves 4..7 leave the 4 DMA just issued in flight
            if (late) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            wuse = wload;
            asm volatile("" : "+v"(wuse.q), "+v"(wuse.s_raw), "+v"(wuse.z_raw)::"memory");
        } else if constexpr (MORE) {
            if (!late) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // DMA of M1 (of the prologue in step 0)
        }
        GQ_TR(9);
        seg_barrier();
        GQ_TR(8);
    };
    using T = std::true_type;
    using F = std::false_type;
#ifdef GQ_W4_TRACE
    const unsigned long long tk1 = __builtin_readcyclecounter();
#endif
    int kt = 0;
    if (n_ktiles > 2) {
        body(T{}, T{}, T{}, 0, wcur, wnext);
        for (kt = 1; kt + 2 < n_ktiles; ++kt) body(T{}, T{}, F{}, kt, wcur, wnext);
        body(T{}, F{}, F{}, kt, wcur, wnext);
        ++kt;
        body(F{}, F{}, F{}, kt, wcur, wnext);
    } else if (n_ktiles == 2) {
        body(T{}, F{}, T{}, 0, wcur, wnext);
        body(F{}, F{}, F{}, 1, wcur, wnext);
    } else {
        body(F{}, F{}, T{}, 0, wcur, wnext);
    }
    if (!late) __builtin_amdgcn_s_barrier();
#ifdef GQ_W4_TRACE
    const unsigned long long tk2 = __builtin_readcyclecounter();
    if (g_w4_trace && blockIdx.x == 8 && lane == 0) {
#pragma unroll
        for (int i = 0; i < 13; ++i) g_w4_trace[wave * 32 + i] = ts[i];
    }
#endif

#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t row = m0 + (wr * 4 + a) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                const int64_t col = n0 + (wc * NB + b) * 32 + li;
                if (row < m_total && col < r_total) {
                    if constexpr (DT == GQ_F16) y[row * r_total + col] = __builtin_bit_cast(unsigned short, (_Float16)acc[a][b][e]);
                    else y[row * r_total + col] = __builtin_bit_cast(unsigned short, (__bf16)acc[a][b][e]);
                }
            }
#ifdef GQ_W4_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long tk3 = __builtin_readcyclecounter(), tr3 = __builtin_amdgcn_s_memrealtime();
    if (g_w4_trace && blockIdx.x == 8 && lane == 0) {
        unsigned long long *o = g_w4_trace + wave * 32 + 13;
        o[0] = tk0; o[1] = tk1; o[2] = tk2; o[3] = tk3; o[4] = tr0; o[5] = tr3;
    }
#endif
}

}  // namespace

#ifdef GQ_W4_TRACE
extern "C" int gq_debug_w4_trace(unsigned long long *buf)
{
    GQ_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_w4_trace), &buf, sizeof(buf)));
    return GQ_OK;
}
#endif

namespace {

std::atomic<int> g_w4_tile{-1};   // -1: not set yet (GQ_W4_TILE decides, else automatic)

int w4a16_launch(void *y, const void *x, const uint32_t *qweight_packed, const void *scale, const uint8_t *qzero, int64_t m,
                 int64_t r, int64_t c, int group_size, int dtype, gq_stream_t stream, const char *who)
{
    GQ_REQUIRE(((y && x) || m == 0) && qweight_packed && scale && qzero, "%s: NULL pointer", who);   // no rows: no data pointers
    GQ_REQUIRE(dtype == GQ_F16 || dtype == GQ_BF16, "%s: bad dtype %d", who, dtype);
    GQ_REQUIRE(m >= 0 && r > 0 && c > 0, "%s: bad shape", who);
    GQ_REQUIRE(c % WK == 0, "%s: in_features %lld must be a multiple of %d", who, (long long)c, WK);
    GQ_REQUIRE(group_size >= 32 && group_size % 32 == 0 && c % group_size == 0,
               "%s: group size %d must be a multiple of 32 dividing in_features", who, group_size);
    GQ_REQUIRE((((uintptr_t)x | (uintptr_t)qweight_packed | (uintptr_t)y) & 15) == 0, "%s: pointers must be 16-byte aligned", who);
    if (m == 0) return GQ_OK;
    // Tile shape: gq_set_w4_tile / GQ_W4_TILE force one (128 = 128 x 128, 256 = 256 x 256, 384 = 256 x 128) where the
    // shape allows it; automatic otherwise:
    //  * 256 x 256 tiles unless they would occupy at most half of the chip's 256 CUs: four times as many 128 x 128
    //    tiles then finish sooner (measured at M = 2048: 717 against 642 TFLOP/s for 4096 x 4096);
    //  * 256 (M) x 128 tiles where they need less time in rounds of one workgroup per CU.  Measured (fp16, K = 4096): a
    //    full round of theirs takes 0.62 of a 256 x 256 round (M = 16384: 869 against 1083 TFLOP/s), and at M = 2048
    //      4096 rows   (128 wide tiles, half a round | 256 narrow, one round): 97 us (the 128 x 128 kernel) -> 86 us
    //      4096 rows, K = 11008:                                               235 us -> 218 us
    //      11008 rows  (344 wide = 2 rounds | 688 narrow = 3 rounds):          225 us -> 257 us: stays wide
    //    so a narrow round is costed at 0.7 of a wide one (3 x 0.7 > 2); with at most 128 narrow tiles the 128 x 128
    //    kernel still has one workgroup per CU or fewer and its smaller tiles are done sooner.
    int forced = g_w4_tile.load();
    if (forced < 0) {
        const char *e = getenv("GQ_W4_TILE");
        forced = e ? atoi(e) : 0;
        int expect = -1;
        g_w4_tile.compare_exchange_strong(expect, forced);
    }
    const int64_t t_wide = gq_cdiv(m, W2T) * gq_cdiv(r, W2T), t_narrow = gq_cdiv(m, W2T) * gq_cdiv(r, 128);
    const bool few_tiles = t_wide <= 128;
    const bool can256 = m >= W2T && group_size % WK == 0;
    const bool narrow = forced ? forced == 384 : t_narrow > 128 && (double)gq_cdiv(t_narrow, 256) * 0.7 < (double)gq_cdiv(t_wide, 256);
    if (can256 && r >= 128 && narrow) {
        const int tiles_m = (int)gq_cdiv(m, W2T), tiles_n = (int)gq_cdiv(r, 128);
        const int n_blocks = tiles_m * tiles_n;
        auto kern1 = dtype == GQ_F16 ? w4a16_256_kernel<GQ_F16, 1> : w4a16_256_kernel<GQ_BF16, 1>;
        GQ_HIP(hipFuncSetAttribute((const void *)kern1, hipFuncAttributeMaxDynamicSharedMemorySize, W2_LDS));
        hipLaunchKernelGGL(kern1, dim3(n_blocks), dim3(512), W2_LDS, (hipStream_t)stream, (unsigned short *)y,
                           (const unsigned short *)x, qweight_packed, (const unsigned short *)scale, qzero, m, r, c,
                           group_size, tiles_n, n_blocks, 2);
        GQ_LAUNCH_CHECK();
        return GQ_OK;
    }
    if (can256 && r >= W2T && forced != 128 && (!few_tiles || forced == 256)) {
        const int tiles_m = (int)gq_cdiv(m, W2T), tiles_n = (int)gq_cdiv(r, W2T);
        const int n_blocks = tiles_m * tiles_n;
        static const int band_env = [] {
            const char *e = getenv("GQ_W4_BAND");
            return e ? atoi(e) : 0;
        }();
        const int band = band_env > 0 ? band_env : 2;   // memory-side reads of the 11008 x 4096 shape: -32 % against 1; run time unchanged
        auto kern2 = dtype == GQ_F16 ? w4a16_256_kernel<GQ_F16, 2> : w4a16_256_kernel<GQ_BF16, 2>;
        GQ_HIP(hipFuncSetAttribute((const void *)kern2, hipFuncAttributeMaxDynamicSharedMemorySize, W2_LDS));
        hipLaunchKernelGGL(kern2, dim3(n_blocks), dim3(512), W2_LDS, (hipStream_t)stream, (unsigned short *)y,
                           (const unsigned short *)x, qweight_packed, (const unsigned short *)scale, qzero, m, r, c,
                           group_size, tiles_n, n_blocks, band);
        GQ_LAUNCH_CHECK();
        return GQ_OK;
    }
    const int tiles_m = (int)gq_cdiv(m, WT), tiles_n = (int)gq_cdiv(r, WT);
    const int n_blocks = tiles_m * tiles_n;
    auto kern = dtype == GQ_F16 ? w4a16_kernel<GQ_F16> : w4a16_kernel<GQ_BF16>;
    GQ_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS));
    hipLaunchKernelGGL(kern, dim3(n_blocks), dim3(256), W_LDS, (hipStream_t)stream, (unsigned short *)y,
                       (const unsigned short *)x, qweight_packed, (const unsigned short *)scale, qzero, m, r, c, group_size,
                       tiles_n, n_blocks);
    GQ_LAUNCH_CHECK();
    return GQ_OK;
}

}  // namespace

int gq_wq_stream_launch(void *y, const void *x, const uint32_t *qweight_packed, const void *scale, const uint8_t *qzero,
                        int64_t m, int64_t r, int64_t c, int group_size, int bits, int dtype, hipStream_t stream);

namespace {
std::atomic<int> g_wq_variant{-1};   // -1: not set yet (GQ_WQ_KERNEL=stream / split decides, else automatic)
}

extern "C" int gq_set_w4_tile(int tile)
{
    GQ_REQUIRE(tile == 0 || tile == 128 || tile == 256 || tile == 384,
               "gq_set_w4_tile: 0 (automatic), 128 (128 x 128), 256 (256 x 256) or 384 (256 x 128), got %d", tile);
    const int before = g_w4_tile.exchange(tile);
    return before < 0 ? 0 : before;
}

extern "C" int gq_set_wq_kernel(int variant)
{
    GQ_REQUIRE(variant >= 0 && variant <= 2, "gq_set_wq_kernel: variant must be 0 (automatic), 1 (stream) or 2 (split), got %d", variant);
    const int before = g_wq_variant.exchange(variant);
    return before < 0 ? 0 : before;
}

extern "C" int gq_wq_gemm(void *y, const void *x, const uint32_t *qweight_packed, const void *scale, const uint8_t *qzero,
                          int64_t m, int64_t r, int64_t c, int group_size, int bits, int dtype, gq_stream_t stream)
{
    const char *who = "gq_wq_gemm";
    GQ_REQUIRE(bits == 3 || bits == 4, "%s: bits must be 3 or 4, got %d", who, bits);
    int variant = g_wq_variant.load();
    if (variant < 0) {
        const char *e = getenv("GQ_WQ_KERNEL");
        variant = !e ? 0 : (e[0] == 's' && e[1] == 't') ? 1 : 2;
        int expect = -1;
        g_wq_variant.compare_exchange_strong(expect, variant);
    }
    // 4 bits, measured at M = 16384 on Llama shapes: fp16 -- split 1150-1200 TFLOP/s, stream 980-1130; bf16 (whose
    // expansion is the longer one) -- split 995-1030, stream 1055-1165, and 761 against 694 at M = 2048 once
    // there are more tiles than CUs.  So: the split kernel, except bf16 with more than 256 tiles.
    const bool stream4 = variant == 1 || (variant == 0 && dtype == GQ_BF16 && gq_cdiv(m, W2T) * gq_cdiv(r, W2T) > 256);
    if (bits == 4 && !stream4) return w4a16_launch(y, x, qweight_packed, scale, qzero, m, r, c, group_size, dtype, stream, who);
    GQ_REQUIRE(((y && x) || m == 0) && qweight_packed && scale && qzero, "%s: NULL pointer", who);
    GQ_REQUIRE(dtype == GQ_F16 || dtype == GQ_BF16, "%s: bad dtype %d", who, dtype);
    GQ_REQUIRE(m >= 0 && r > 0 && c > 0, "%s: bad shape", who);
    GQ_REQUIRE(c % WK == 0 && c < (1 << 21) && r < (1 << 22), "%s: in_features %lld must be a multiple of %d below 2^21, out_features below 2^22",
               who, (long long)c, WK);
    GQ_REQUIRE(group_size >= 32 && group_size % 32 == 0 && c % group_size == 0,
               "%s: group size %d must be a multiple of 32 dividing in_features", who, group_size);
    GQ_REQUIRE((((uintptr_t)x | (uintptr_t)qweight_packed | (uintptr_t)y) & 15) == 0, "%s: pointers must be 16-byte aligned", who);
    if (m == 0) return GQ_OK;
    return gq_wq_stream_launch(y, x, qweight_packed, scale, qzero, m, r, c, group_size, bits, dtype, (hipStream_t)stream);
}

extern "C" int gq_w4a16_gemm(void *y, const void *x, const uint32_t *qweight_packed, const void *scale,
                             const uint8_t *qzero, int64_t m, int64_t r, int64_t c, int group_size, gq_stream_t stream)
{
    return w4a16_launch(y, x, qweight_packed, scale, qzero, m, r, c, group_size, GQ_F16, stream, "gq_w4a16_gemm");
}

extern "C" int gq_w4a16_gemm_dt(void *y, const void *x, const uint32_t *qweight_packed, const void *scale,
                                const uint8_t *qzero, int64_t m, int64_t r, int64_t c, int group_size, int dtype,
                                gq_stream_t stream)
{
    return w4a16_launch(y, x, qweight_packed, scale, qzero, m, r, c, group_size, dtype, stream, "gq_w4a16_gemm_dt");
}
