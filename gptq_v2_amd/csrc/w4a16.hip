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

__device__ __forceinline__ void load_w(WRegs &w, const uint32_t *__restrict__ qw, const unsigned short *__restrict__ scale,
                                       const uint8_t *__restrict__ zero, int64_t r_total, int64_t words_per_row,
                                       int n_groups, int group_size, int64_t n0, int64_t k0, int tid)
{
    int64_t n = n0 + (tid >> 1);
    n = n < r_total ? n : r_total - 1;
    const int half = tid & 1;
    w.q = *reinterpret_cast<const u32x4 *>(qw + n * words_per_row + k0 / 8 + half * 4);
    const int g = (int)((k0 + half * 32) / group_size);
    w.s_raw = scale[n * n_groups + g];
    w.z_raw = zero[n * n_groups + g];
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

// one quarter of store_w: the 8 values of word `wi`, placed between MFMA groups by the 256-tile kernel
template <int WI>
__device__ __forceinline__ void store_w_part(const WRegs &w, char *lds_b, int tid)
{
    const int row = tid >> 1, half = tid & 1;
    f16x2 s2, z2;
    w_scale_zero(w, s2, z2);
    const unsigned word = w.q[WI];
    u32x4 out;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const unsigned bits = ((word >> (4 * p)) & 0x000F000Fu) | 0x64006400u;
        f16x2 v = __builtin_bit_cast(f16x2, bits);
        v = v - z2;
        v = v * s2;
        out[p] = __builtin_bit_cast(unsigned, v);
    }
    const int chunk = (half * 4 + WI) ^ ((row >> 1) & 7);
    *reinterpret_cast<u32x4 *>(lds_b + row * 128 + chunk * 16) = out;
}

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
    const int64_t words_per_row = c_total / 8;
    const int n_groups = (int)(c_total / group_size);
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
    stage_x(x, c_total, m_total, m0, 0, smem, wave, lane);
    load_w(wregs, qw, scale, zero, r_total, words_per_row, n_groups, group_size, n0, 0, tid);
    store_w(wregs, smem + W_OP_BYTES, tid);

    auto step = [&](auto cur_c, int kt) {
        constexpr int cur = decltype(cur_c)::value;
        constexpr int SB = cur * W_STAGE_BYTES;
        __syncthreads();   // x tile of this stage landed (vmcnt drained), W tile written, other stage free
        const bool more = kt + 1 < n_ktiles;
        if (more) {
            stage_x(x, c_total, m_total, m0, (int64_t)(kt + 1) * WK, smem + (cur ^ 1) * W_STAGE_BYTES, wave, lane);
            load_w(wregs, qw, scale, zero, r_total, words_per_row, n_groups, group_size, n0, (int64_t)(kt + 1) * WK, tid);
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
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[a]),
                                                                       __builtin_bit_cast(f16x8, fb[b]), acc[a][b], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more) store_w(wregs, smem + (cur ^ 1) * W_STAGE_BYTES + W_OP_BYTES, tid);
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
                if (row < m_total && col < r_total)
                    y[row * r_total + col] = __builtin_bit_cast(unsigned short, (_Float16)acc[a][b][e]);
            }
}

// ---- 256 x 256 tiles -------------------------------------------------------------------------
// Half the L2 -> LDS bytes per flop of the 128 x 128 kernel: 512 threads, 8 waves as 2 (M) x 4 (N),
// each wave a 128 x 64 sub-tile (4 x 2 MFMA blocks, 128 accumulator registers), 64-deep k-steps,
// two LDS stages of 64 KiB (x image 32 KiB + W image 32 KiB), one workgroup per CU.  Same images,
// swizzle and dequantisation as above; every thread expands the 32 nibbles of half a W row.
// The two waves of each SIMD run half a k-step apart (see the main loop).
constexpr int W2T = 256;
constexpr int W2_OP_BYTES = W2T * WK * 2;        // 32 KiB
constexpr int W2_STAGE_BYTES = 2 * W2_OP_BYTES;  // 64 KiB
constexpr int W2_LDS = 2 * W2_STAGE_BYTES;       // 128 KiB

__device__ __forceinline__ void stage_x2(const unsigned short *__restrict__ x, int64_t ld, int64_t m_total, int64_t m0,
                                         int64_t k0, char *lds_a, int wave, int lane)
{
    const int rloc = lane >> 3, s = lane & 7;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int ii = wave * 4 + it;       // 32 wave-instructions of 8 rows
        const int row = ii * 8 + rloc;
        const int chunk = s ^ ((row >> 1) & 7);
        int64_t m = m0 + row;
        m = m < m_total ? m : m_total - 1;
        const unsigned short *src = x + m * ld + k0 + chunk * 8;
        __builtin_amdgcn_global_load_lds(GQ_GLOBAL_PTR(src), GQ_LDS_PTR(void, lds_a + ii * 1024), 16, 0, 0);
    }
}

__global__ __launch_bounds__(512) void w4a16_256_kernel(unsigned short *__restrict__ y, const unsigned short *__restrict__ x,
                                                        const uint32_t *__restrict__ qw, const unsigned short *__restrict__ scale,
                                                        const uint8_t *__restrict__ zero, int64_t m_total, int64_t r_total,
                                                        int64_t c_total, int group_size, int tiles_n, int n_blocks)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    int bid;
    {
        const int orig = blockIdx.x, xcd = orig & 7;
        const int q = n_blocks >> 3, r = n_blocks & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int64_t m0 = (int64_t)(bid / tiles_n) * W2T, n0 = (int64_t)(bid % tiles_n) * W2T;
    const int64_t words_per_row = c_total / 8;
    const int n_groups = (int)(c_total / group_size);
    const int n_ktiles = (int)(c_total / WK);

    const int wr = wave >> 2, wc = wave & 3;
    const int li = lane & 31, lh = lane >> 5;
    const unsigned lds_base = (unsigned)(uintptr_t)GQ_LDS_PTR(char, smem);
    // row bases of the wave's MFMA blocks; the chunk of k-step kk is (2*kk + lh) ^ sw with
    // sw = (row >> 1) & 7 = (li >> 1) & 7 for every block (block offsets are multiples of 32 rows)
    const int sw = (li >> 1) & 7;
    unsigned rowA[4], rowB[2], choff[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) rowA[u] = lds_base + ((wr * 4 + u) * 32 + li) * 128;
#pragma unroll
    for (int u = 0; u < 2; ++u) rowB[u] = lds_base + W2_OP_BYTES + ((wc * 2 + u) * 32 + li) * 128;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) choff[kk] = (unsigned)(((2 * kk + lh) ^ sw) << 4);

    f32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
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
    WRegs wcur, wnext;
    stage_x2(x, c_total, m_total, m0, 0, smem, wave, lane);
    load_w(wcur, qw, scale, zero, r_total, words_per_row, n_groups, group_size, n0, 0, tid);
    store_w(wcur, smem + W2_OP_BYTES, tid);
    if (n_ktiles > 1) load_w(wcur, qw, scale, zero, r_total, words_per_row, n_groups, group_size, n0, WK, tid);
    __syncthreads();
    const bool late = wave >= 4;   // wave-uniform (readfirstlane above)
    if (late) __builtin_amdgcn_s_barrier();

    u32x4 fa[2][4], fb[2][2];
    auto read_frags = [&](int slot, unsigned sb, int kk) {
#pragma unroll
        for (int u = 0; u < 4; ++u) lds_read_b128<0>(fa[slot][u], rowA[u] + sb + choff[kk]);
#pragma unroll
        for (int u = 0; u < 2; ++u) lds_read_b128<0>(fb[slot][u], rowB[u] + sb + choff[kk]);
    };
    auto mfmas = [&](int slot) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[slot][a]),
                                                                   __builtin_bit_cast(f16x8, fb[slot][b]), acc[a][b], 0, 0, 0);
    };
    auto seg_barrier = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };

    // MORE: a next step exists (stage its x tile, expand its W tile from `wuse`);
    // MORE2: a step after that exists (fetch its packed W words into `wload`).
    // The steady-state loop runs with both flags constant, so hipcc's own load counting stays exact
    // (no branches between a load and its use).
    auto body = [&](auto more_c, auto more2_c, int kt, WRegs &wuse, WRegs &wload) {
        constexpr bool MORE = decltype(more_c)::value, MORE2 = decltype(more2_c)::value;
        const unsigned sb = (unsigned)(kt & 1) * W2_STAGE_BYTES;
        char *nxt = smem + ((kt + 1) & 1) * W2_STAGE_BYTES;
        char *nb = nxt + W2_OP_BYTES;
        // ---- R1
        // Use `wuse` once BEFORE any DMA is issued: with an LDS-DMA in flight hipcc waits vmcnt(0)
        // at the first use of an ordinary load's result, which here would drain the DMA just issued;
        // at this point only the (old) loads of `wuse` itself can be outstanding.
        if constexpr (MORE) asm volatile("" ::"v"(wuse.q), "v"(wuse.s_raw), "v"(wuse.z_raw));
        if constexpr (MORE) stage_x2(x, c_total, m_total, m0, (int64_t)(kt + 1) * WK, nxt, wave, lane);
        if constexpr (MORE2)
            load_w(wload, qw, scale, zero, r_total, words_per_row, n_groups, group_size, n0, (int64_t)(kt + 2) * WK, tid);
        read_frags(0, sb, 0);
        read_frags(1, sb, 1);
        if constexpr (MORE) {
            store_w_part<0>(wuse, nb, tid);
            store_w_part<1>(wuse, nb, tid);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        seg_barrier();
        // ---- M1
        __builtin_amdgcn_s_setprio(1);
        mfmas(0);
        mfmas(1);
        __builtin_amdgcn_s_setprio(0);
        seg_barrier();
        // ---- R2
        read_frags(0, sb, 2);
        read_frags(1, sb, 3);
        if constexpr (MORE) {
            store_w_part<2>(wuse, nb, tid);
            store_w_part<3>(wuse, nb, tid);
        }
        // this wave's DMA of the next x tile must have landed before the barrier that frees it for
        // reading; the 3 younger loads (W words, scale, zero of step k+2) may stay in flight
        if constexpr (MORE2) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        seg_barrier();
        // ---- M2
        __builtin_amdgcn_s_setprio(1);
        mfmas(0);
        mfmas(1);
        __builtin_amdgcn_s_setprio(0);
        seg_barrier();
    };
    using T = std::true_type;
    using F = std::false_type;
    int kt = 0;
    for (; kt + 2 < n_ktiles; ++kt) {
        body(T{}, T{}, kt, wcur, wnext);
        wcur = wnext;   // the loads into wnext are a whole step old here
    }
    if (kt + 1 < n_ktiles) {
        body(T{}, F{}, kt, wcur, wnext);
        ++kt;
    }
    if (kt < n_ktiles) body(F{}, F{}, kt, wcur, wnext);
    if (!late) __builtin_amdgcn_s_barrier();

#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t row = m0 + (wr * 4 + a) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                const int64_t col = n0 + (wc * 2 + b) * 32 + li;
                if (row < m_total && col < r_total)
                    y[row * r_total + col] = __builtin_bit_cast(unsigned short, (_Float16)acc[a][b][e]);
            }
}

}  // namespace

extern "C" int gq_w4a16_gemm(void *y, const void *x, const uint32_t *qweight_packed, const void *scale,
                             const uint8_t *qzero, int64_t m, int64_t r, int64_t c, int group_size, gq_stream_t stream)
{
    GQ_REQUIRE(y && x && qweight_packed && scale && qzero, "gq_w4a16_gemm: NULL pointer");
    GQ_REQUIRE(m >= 0 && r > 0 && c > 0, "gq_w4a16_gemm: bad shape");
    GQ_REQUIRE(c % WK == 0, "gq_w4a16_gemm: in_features %lld must be a multiple of %d", (long long)c, WK);
    GQ_REQUIRE(group_size >= 32 && group_size % 32 == 0 && c % group_size == 0,
               "gq_w4a16_gemm: group size %d must be a multiple of 32 dividing in_features", group_size);
    GQ_REQUIRE((((uintptr_t)x | (uintptr_t)qweight_packed | (uintptr_t)y) & 15) == 0, "gq_w4a16_gemm: pointers must be 16-byte aligned");
    if (m == 0) return GQ_OK;
    if (m >= W2T && r >= W2T) {
        const int tiles_m = (int)gq_cdiv(m, W2T), tiles_n = (int)gq_cdiv(r, W2T);
        const int n_blocks = tiles_m * tiles_n;
        GQ_HIP(hipFuncSetAttribute((const void *)w4a16_256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, W2_LDS));
        hipLaunchKernelGGL(w4a16_256_kernel, dim3(n_blocks), dim3(512), W2_LDS, (hipStream_t)stream, (unsigned short *)y,
                           (const unsigned short *)x, qweight_packed, (const unsigned short *)scale, qzero, m, r, c,
                           group_size, tiles_n, n_blocks);
        GQ_LAUNCH_CHECK();
        return GQ_OK;
    }
    const int tiles_m = (int)gq_cdiv(m, WT), tiles_n = (int)gq_cdiv(r, WT);
    const int n_blocks = tiles_m * tiles_n;
    GQ_HIP(hipFuncSetAttribute((const void *)w4a16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS));
    hipLaunchKernelGGL(w4a16_kernel, dim3(n_blocks), dim3(256), W_LDS, (hipStream_t)stream, (unsigned short *)y,
                       (const unsigned short *)x, qweight_packed, (const unsigned short *)scale, qzero, m, r, c, group_size,
                       tiles_n, n_blocks);
    GQ_LAUNCH_CHECK();
    return GQ_OK;
}
