// y[M][N] = x[M][K] * w[N][K]^T for fp16 / bf16 operands, fp32 accumulation, 16-bit output:
// the dense nn.Linear the reference runs on a reconstructed weight when it replays the
// calibration batches through a freshly quantised layer (llama.py:172-177, quant.py:262-279).
// At 32768 tokens per launch the packed form of the weight saves nothing (every 256-row band of x
// would expand the same nibbles again), so the replay reads the dense weight that
// reconstruct_nn_linear has just produced.
//
// Both operands are K-contiguous rows, so both go global -> LDS by LDS-DMA as they lie in memory
// and are read back with ds_read_b128.  256 x 256 output tile, 8 waves as 2 (M) x 4 (N), 64-deep
// k-steps, two LDS stages of 64 KiB (x image 32 KiB + w image 32 KiB), one workgroup per CU.  LDS
// images are [256 rows][64 halves] with the 16-byte chunk index XOR-ed with (row >> 1) & 7
// (conflict-free ds_read_b128 for the 16 x 16 x 32 fragment shape as well as the 32 x 32 x 16 one);
// the swizzle is applied to the DMA's source address.
//
// Schedule: every wave runs ONE stream in which the fragment reads of the next 16 MFMAs and the
// wave's DMA pieces are issued between the MFMAs of the current 16, into the other of two fragment
// register sets.  The two waves of a SIMD both issue MFMAs and the matrix pipe takes them in turn,
// so neither the LDS latency of one wave nor its DMA issue leaves the pipe without work.  The only
// workgroup barrier of a k-step sits behind its third quarter: by then every wave has read the last
// fragments of the step's stage (the fourth quarter's are read during the third), so the stage is
// free for the DMA of step t+2, and every wave has waited for its own pieces of step t+1, so the
// other stage is complete for the reads that the fourth quarter issues.
//
// Measured at 32768 tokens (MI355X, synthetic activations): 2280-2320 cycles per k-step against 2048
// of pure MFMA issue, at a clock of 1.35-1.7 GHz -- the launch is limited by the power the chip may
// draw, not by issue slots.  The same loop with its loads, fragment reads and waits all removed (MFMAs
// only, on one wave per SIMD) reaches 1385-1395 TFLOP/s over back-to-back launches, 0.56 of the
// nominal peak: that is the ceiling for this instruction under the cap; this kernel holds 1250-1340
// there, the library's GEMM 1340-1390.  The matrix pipe is busy 0.67-0.76 of the cycles here and
// 0.78-0.89 in the library's kernel, which runs at a correspondingly lower clock (1.5-1.6 against
// 1.7-1.85 GHz): idle cycles are not free, they move the chip to a higher, less efficient voltage.
// What moved the speed: v_mfma_f32_16x16x32 in place of 32x32x16 at the same wave tile gave 5-7 %,
// immediate offsets in place of address arithmetic for the fragment reads 1-3 %, bands of 3 tile
// rows instead of 2 0.5-3 % (bands of 4 need the XCDs to start at different k to match that: without
// the stagger all eight fetch the same w panels at once and lose 5 %).  What did not: a four-barrier read / MFMA role split of the two waves
// of a SIMD (4-8 % slower); non-temporal stores of y (3-7 % slower) and non-temporal loads of either
// operand (10-18 % slower: the re-reads of x and w are served by the caches the hint bypasses); four
// waves of 128 x 128 with the accumulators in 256 hand-assigned AGPRs (the library's shape: a third
// fewer LDS bytes per flop), as a persistent grid of one workgroup per CU that fetches the next
// tile's first two k-steps during the last two of the current one: equal to this kernel within 1 %
// when launches are interleaved with other work, 8 % slower back to back, and the persistent grid
// changed neither its cycles nor its pipe occupancy (0.71-0.74) -- prologue and epilogue are not what
// is lost.  In that kernel, dropping the DMA issue alone gained 13-18 %, dropping the fragment reads
// alone 9 %, dropping the wait for the DMA nothing: the cost of the loads is the power their traffic
// draws, not their latency.  (6.3 GB are fetched past the L2 per 12288-row launch with bands of 2 and
// the library's tile order fetches 36 % less, but bands of 4 or 8 with the stagger, which fetch a third
// less too, change nothing measurable, in isolation or inside the whole quantisation step.)
// (For such experiments: accumulators bound to inline-asm MFMAs as C++ variables are moved between
// registers by hipcc, which does not know that the asm is a matrix instruction still in flight, and
// come out wrong; through the builtin every result gets a register of its own and three moves per
// MFMA put them back.  Hard-coded a[N:N+3] operands with a clobber list of all AGPRs work.)
#include <algorithm>
#include <type_traits>

#include <cstdlib>

#include "gq_common.h"

namespace {

constexpr int NT = 256;                       // tile edge
constexpr int NK = 64;                        // k-step
constexpr int N_OP_BYTES = NT * NK * 2;       // 32 KiB per operand image
constexpr int N_STAGE_BYTES = 2 * N_OP_BYTES; // 64 KiB
constexpr int N_LDS = 2 * N_STAGE_BYTES;      // 128 KiB

// Fragment read issued from inline asm: hipcc must not see it as an LDS load, or it orders it behind
// the in-flight LDS-DMA of the other stage with a vmcnt(0).  The waits are placed by hand
// (lgkmcnt(0), each followed by sched_barrier(0) so that no MFMA is hoisted above it).
__device__ __forceinline__ void nt_read_b128(u32x4 &dst, unsigned lds_addr)
{
    asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(lds_addr));
}

template <int OFF>
__device__ __forceinline__ void nt_read_b128_off(u32x4 &dst, unsigned lds_addr)
{
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(lds_addr), "i"(OFF));
}

#ifdef GQ_NT_TRACE
// Debug build only (tools/trace_nt.py): per-workgroup time stamps (100 MHz real-time counter) and
// the hardware ids of the CU it ran on.
__device__ unsigned long long *g_nt_trace = nullptr;
#endif

// The wave's 128 x 64 sub-tile is 8 (x rows) x 4 (w rows) blocks of 16 x 16; a k-step is two 32-deep
// chunks, a chunk two halves of 16 MFMAs (4 x-row blocks each against the chunk's 4 w-row blocks).
// A operand = w rows, B operand = x rows, so a lane of the result holds 4 consecutive output
// features of one token: one 8-byte store.  Fragment registers: two sets of w fragments (by chunk
// parity) and two sets of x fragments (by half); each set is loaded while the other is in use.
typedef float f32x4v __attribute__((ext_vector_type(4)));

template <int DT>
__global__ __launch_bounds__(512) void gemm_nt_kernel(unsigned short *__restrict__ y, const unsigned short *__restrict__ x,
                                                       const unsigned short *__restrict__ w, int64_t m_total, int64_t n_total,
                                                       int64_t k_total, int tiles_n, int n_blocks, int band)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef GQ_NT_TRACE
    const unsigned long long tr0 = __builtin_amdgcn_s_memrealtime();
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
    int tm, tn;
    {
        const int per_band = band * tiles_n;
        const int b = bid / per_band, within = bid - b * per_band;
        const int tiles_m = n_blocks / tiles_n;
        const int rows = min(band, tiles_m - b * band);
        tn = within / rows;
        tm = b * band + (within - tn * rows);
    }
    const int64_t m0 = (int64_t)tm * NT, n0 = (int64_t)tn * NT;
    const int n_ktiles = (int)(k_total / NK);

    const int wr = wave >> 2, wc = wave & 3;
    const int l15 = lane & 15, lg = lane >> 4;
    const unsigned lds_base = (unsigned)(uintptr_t)GQ_LDS_PTR(char, smem);
    // fragment of block u, chunk c: row (block base + l15), 16-byte chunk (4 c + lg) ^ ((row >> 1) & 7);
    // block bases are multiples of 16 rows, so the swizzle term is ((l15 >> 1) & 7) for every block
    const int sw = (l15 >> 1) & 7;
    const unsigned xrow = lds_base + (wr * 128 + l15) * 128;               // + u * 2048, u = 0..7
    const unsigned wrow = lds_base + N_OP_BYTES + (wc * 64 + l15) * 128;   // + u * 2048, u = 0..3
    unsigned choff[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) choff[c] = (unsigned)(((4 * c + lg) ^ sw) << 4);

    f32x4v acc[4][8];   // [w block][x block]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = f32x4v{0.f, 0.f, 0.f, 0.f};

    // DMA source addressing: a wave-uniform 64-bit base (the tile's first row at the current k) plus one
    // 32-bit byte offset per piece and lane, fixed for the whole tile (row, clamped to the last row of
    // the operand, and swizzled chunk); the host checks that 256 rows stay below 2^31 bytes
    const unsigned short *xt = x + m0 * k_total, *wt = w + n0 * k_total;
    unsigned voff[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = (wave * 4 + (i & 3)) * 8 + (lane >> 3);
        const int64_t lim = (i < 4 ? m_total - m0 : n_total - n0) - 1;
        const int rc = (int)(row < lim ? row : lim);
        voff[i] = (unsigned)rc * (unsigned)(k_total * 2) + (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) << 4);
    }
    auto dma = [&](int kt, char *stage, int i) {
        const char *base = reinterpret_cast<const char *>((i < 4 ? xt : wt) + (int64_t)kt * NK);
        char *dst = stage + (i < 4 ? 0 : N_OP_BYTES) + (wave * 4 + (i & 3)) * 1024;
        __builtin_amdgcn_global_load_lds(GQ_GLOBAL_PTR(base + voff[i]), GQ_LDS_PTR(void, dst), 16, 0, 0);
    };
    u32x4 fw[2][4], fx[2][4];   // fw[chunk parity][w block], fx[half][x block of the half]
    // block u of an image is u * 2048 bytes behind block 0: an immediate offset of the read
    auto read_w = [&](int set, unsigned sb, int c, int u) __attribute__((always_inline)) {
        const unsigned addr = wrow + sb + choff[c];
        switch (u) {
        case 0: nt_read_b128_off<0>(fw[set][0], addr); break;
        case 1: nt_read_b128_off<2048>(fw[set][1], addr); break;
        case 2: nt_read_b128_off<4096>(fw[set][2], addr); break;
        default: nt_read_b128_off<6144>(fw[set][3], addr); break;
        }
    };
    auto read_x = [&](int half, unsigned sb, int c, int u) __attribute__((always_inline)) {
        const unsigned addr = xrow + sb + choff[c] + half * 8192;
        switch (u) {
        case 0: nt_read_b128_off<0>(fx[half][0], addr); break;
        case 1: nt_read_b128_off<2048>(fx[half][1], addr); break;
        case 2: nt_read_b128_off<4096>(fx[half][2], addr); break;
        default: nt_read_b128_off<6144>(fx[half][3], addr); break;
        }
    };
    // 16 MFMAs: x blocks of `half` against the w blocks of set `ws`; hook(i) runs behind MFMA i
    auto half_step = [&](int ws, int half, auto hook) {
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                if constexpr (DT == GQ_F16)
                    acc[a][half * 4 + b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fw[ws][a]),
                                                                                  __builtin_bit_cast(f16x8, fx[half][b]),
                                                                                  acc[a][half * 4 + b], 0, 0, 0);
                else
                    acc[a][half * 4 + b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fw[ws][a]),
                                                                                   __builtin_bit_cast(bf16x8, fx[half][b]),
                                                                                   acc[a][half * 4 + b], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                hook(b * 4 + a);
                __builtin_amdgcn_sched_barrier(0);
            }
    };
    auto reads_done = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };

#pragma unroll
    for (int i = 0; i < 8; ++i) dma(0, smem, i);
    if (n_ktiles > 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) dma(1, smem + N_STAGE_BYTES, i);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 4; ++u) read_w(0, 0u, 0, u);
#pragma unroll
    for (int u = 0; u < 4; ++u) read_x(0, 0u, 0, u);
    reads_done();
#ifdef GQ_NT_TRACE
    const unsigned long long tr1 = __builtin_amdgcn_s_memrealtime(), tc1 = __builtin_readcyclecounter();
#endif

    auto body = [&](auto more_c, auto more2_c, auto first_c, int kt) {
        constexpr bool MORE = decltype(more_c)::value, MORE2 = decltype(more2_c)::value;
        constexpr bool FIRST = decltype(first_c)::value;
        const unsigned sb = (unsigned)(kt & 1) * N_STAGE_BYTES, sbn = (unsigned)((kt + 1) & 1) * N_STAGE_BYTES;
        char *cur = smem + (kt & 1) * N_STAGE_BYTES;
        char *nxt = smem + ((kt + 1) & 1) * N_STAGE_BYTES;
        // (chunk 0, half 0): x fragments of (chunk 0, half 1); second half of this wave's pieces of step kt+1
        half_step(0, 0, [&](int i) {
            if ((i & 1) == 0 && i < 8) read_x(1, sb, 0, i >> 1);
            if constexpr (MORE && !FIRST) {
                if ((i & 3) == 3) dma(kt + 1, nxt, 4 + (i >> 2));
            }
        });
        reads_done();
        // (chunk 0, half 1): w fragments of chunk 1, x fragments of (chunk 1, half 0)
        half_step(0, 1, [&](int i) {
            if ((i & 1) == 0) {
                if (i < 8) read_w(1, sb, 1, i >> 1);
                else read_x(0, sb, 1, (i >> 1) - 4);
            }
        });
        reads_done();
        // (chunk 1, half 0): x fragments of (chunk 1, half 1): the last reads of this stage
        half_step(1, 0, [&](int i) { if ((i & 1) == 0 && i < 8) read_x(1, sb, 1, i >> 1); });
        reads_done();
        if constexpr (MORE) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of step kt+1
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        // (chunk 1, half 1): fragments of (chunk 0, half 0) of step kt+1; first half of the pieces of step kt+2
        half_step(1, 1, [&](int i) {
            if constexpr (MORE) {
                if ((i & 1) == 0) {
                    if (i < 8) read_w(0, sbn, 0, i >> 1);
                    else read_x(0, sbn, 0, (i >> 1) - 4);
                }
            }
            if constexpr (MORE2) {
                if ((i & 3) == 3) dma(kt + 2, cur, i >> 2);
            }
        });
        if constexpr (MORE) reads_done();
    };
    using T = std::true_type;
    using F = std::false_type;
    int kt = 0;
    if (n_ktiles > 2) {
        body(T{}, T{}, T{}, 0);
        for (kt = 1; kt + 2 < n_ktiles; ++kt) body(T{}, T{}, F{}, kt);
        body(T{}, F{}, F{}, kt);
        ++kt;
        body(F{}, F{}, F{}, kt);
    } else if (n_ktiles == 2) {
        body(T{}, F{}, T{}, 0);
        body(F{}, F{}, F{}, 1);
    } else {
        body(F{}, F{}, T{}, 0);
    }
#ifdef GQ_NT_TRACE
    const unsigned long long tr2 = __builtin_amdgcn_s_memrealtime(), tc2 = __builtin_readcyclecounter();
#endif

    // lane: token l15 of x block b, output features 4 lg .. 4 lg + 3 of w block a; 32-bit offsets from
    // the wave's corner of the tile
    {
        unsigned short *yw = y + (m0 + wr * 128) * n_total + n0 + wc * 64;
        const int rows_left = (int)min((int64_t)128, m_total - (m0 + wr * 128));
        const int cols_left = (int)min((int64_t)64, n_total - (n0 + wc * 64));
        const bool wide = (n_total & 3) == 0;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const int row = b * 16 + l15;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int col = a * 16 + 4 * lg;
                unsigned short h[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if constexpr (DT == GQ_F16) h[e] = __builtin_bit_cast(unsigned short, (_Float16)acc[a][b][e]);
                    else h[e] = __builtin_bit_cast(unsigned short, (__bf16)acc[a][b][e]);
                }
                if (row >= rows_left) continue;
                unsigned short *dst = yw + (unsigned)row * (unsigned)n_total + (unsigned)col;
                if (wide && col + 3 < cols_left) {
                    *reinterpret_cast<u32x2 *>(dst) = u32x2{(unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16)};
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (col + e < cols_left) dst[e] = h[e];
                }
            }
        }
    }
#ifdef GQ_NT_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long tr3 = __builtin_amdgcn_s_memrealtime();
    if (g_nt_trace && tid == 0) {
        unsigned long long *o = g_nt_trace + (size_t)blockIdx.x * 8;
        o[0] = tr0; o[1] = tr1; o[2] = tr2; o[3] = tr3;
        o[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_ID
        o[5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // XCC_ID
        o[6] = (unsigned long long)bid;
        o[7] = tc2 - tc1;
    }
#endif
}



}  // namespace

#ifdef GQ_NT_TRACE
extern "C" int gq_debug_nt_trace(unsigned long long *buf)
{
    GQ_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_nt_trace), &buf, sizeof(buf)));
    return GQ_OK;
}
#endif

extern "C" int gq_gemm_nt(void *y, const void *x, const void *w, int64_t m, int64_t n, int64_t k, int dtype,
                          gq_stream_t stream)
{
    GQ_REQUIRE(((y && x) || m == 0) && w, "gq_gemm_nt: NULL pointer");
    GQ_REQUIRE(dtype == GQ_F16 || dtype == GQ_BF16, "gq_gemm_nt: bad dtype %d", dtype);
    GQ_REQUIRE(m >= 0 && n > 0 && k > 0, "gq_gemm_nt: bad shape");
    GQ_REQUIRE(k % NK == 0, "gq_gemm_nt: in_features %lld must be a multiple of %d", (long long)k, NK);
    GQ_REQUIRE((((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) & 15) == 0, "gq_gemm_nt: pointers must be 16-byte aligned");
    GQ_REQUIRE(gq_cdiv(m, NT) * gq_cdiv(n, NT) < (1 << 30), "gq_gemm_nt: too many tiles");
    GQ_REQUIRE(k < (1 << 21) && n < (1 << 22), "gq_gemm_nt: at most 2^21 input and 2^22 output features");
    if (m == 0) return GQ_OK;
    const int tiles_m = (int)gq_cdiv(m, NT), tiles_n = (int)gq_cdiv(n, NT);
    const int n_blocks = tiles_m * tiles_n;
    static const int band_env = [] {   // experiment knob: tile rows per band of the tile order
        const char *e = getenv("GQ_NT_BAND");
        return e ? atoi(e) : 0;
    }();
    // tile rows per band, measured over back-to-back launches at 32768 tokens: 3 beats 2 by 0.5-1 % on the
    // wide shapes (and 4, 5, 6, 8 lose 2-7 % there); the 16-tile-wide, 4096-deep launch gains 6 % from 4
    const int band = band_env > 0 ? band_env : (tiles_n <= 16 && k <= 4096 ? 4 : 3);
    auto kern = dtype == GQ_F16 ? gemm_nt_kernel<GQ_F16> : gemm_nt_kernel<GQ_BF16>;
    GQ_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, N_LDS));
    hipLaunchKernelGGL(kern, dim3(n_blocks), dim3(512), N_LDS, (hipStream_t)stream, (unsigned short *)y,
                       (const unsigned short *)x, (const unsigned short *)w, m, n, k, tiles_n, n_blocks, band);
    GQ_LAUNCH_CHECK();
    return GQ_OK;
}
