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
// immediate offsets in place of address arithmetic for the fragment reads 1-3 %, the persistent grid
// with the next tile's loads in flight over the epilogue (and buffer loads whose tile and k-step are
// scalar operands, in place of a 64-bit address per piece and lane) 1.2-3 %, bands of 3 tile
// rows instead of 2 0.5-3 % (bands of 4 need the XCDs to start at different k to match that: without
// the stagger all eight fetch the same w panels at once and lose 5 %).  What did not: a four-barrier read / MFMA role split of the two waves
// of a SIMD (4-8 % slower); non-temporal stores of y (3-7 % slower) and non-temporal loads of either
// operand (10-18 % slower: the re-reads of x and w are served by the caches the hint bypasses); four
// waves of 128 x 128 with the accumulators in 256 hand-assigned AGPRs (the library's shape: a third
// fewer LDS bytes per flop), as a persistent grid of one workgroup per CU that fetches the next
// tile's first two k-steps during the last two of the current one: equal to this kernel within 1 %
// when launches are interleaved with other work, 8 % slower back to back (one wave per SIMD exposes
// every wait of its own; its pipe occupancy stayed at 0.71-0.74 with or without the persistent grid).
// In that kernel, dropping the DMA issue alone gained 13-18 %, dropping the fragment reads
// alone 9 %, dropping the wait for the DMA nothing: the cost of the loads is the power their traffic
// draws, not their latency.  (6.3 GB are fetched past the L2 per 12288-row launch with bands of 2 and
// the library's tile order fetches 36 % less, but bands of 4 or 8 with the stagger, which fetch a third
// less too, change nothing measurable, in isolation or inside the whole quantisation step.)
// The same switches on this kernel (back-to-back 12288 x 4096 / 4096 x 11008 launches, results wrong of
// course): without any of the waits -- all four lgkmcnt(0), the vmcnt(0), the barrier -- 1220-1225 /
// 1295-1318 TFLOP/s against 1224 / 1300 as built: synchronisation costs nothing measurable.  Without the DMA
// issue 1441 / 1580, without the fragment reads 1472 / 1495, with neither 1738 / 1876 (the operands then
// never change, which also lowers the power of the MFMAs themselves).  What the kernel pays for is moving
// bytes -- 64 KiB per k-step into the LDS and 192 KiB out of it per CU (24 fragment reads of 1 KiB per wave) -- under one power budget with the
// matrix pipe, not waiting for them.
// A 128-token x 256-feature tile on four waves (one per SIMD, the same per-wave stream) for products whose
// 256-token tiles cover only half of the CUs was built and measured slower than leaving those CUs idle: 2048
// tokens x 4096 features x 4096 deep 0.110 against 0.098 ms, x 11008 deep 0.264 against 0.224 ms.  Half of the
// CUs at the clock the power budget then allows do as much as all of them at the lower one, and one wave per
// SIMD has every wait of its own exposed.
// The 4096 x 4096 launch (1146 TFLOP/s against 1350 on the wider shapes): starting every other workgroup 5-80 us
// late so that the tiles' epilogues do not leave in one burst only added the delay (1112 ... 1076).
// Rotating the w panels from band to band, so that the XCDs (which work on different bands) never read the same w
// panel at the same time (with bands of 4 all eight start a 4096 x 4096 launch on panels 0-7): 1075-1078 against
// 1068-1082 TFLOP/s there, nothing on 12288 x 4096, 11008 x 4096 and 4096 x 11008, +1 % on 22016 x 4096 (which
// two launches of 11008 rows beat by 6 % anyway): the shared panels are not what holds the short launch back.
// Cycle stamps per phase of a k-step (one wave of each SIMD pair, 12288 x 4096): the two waves of a SIMD do
// not alternate.  After the barrier the older wave (0-3) wins the matrix pipe whenever it is ready, runs
// its next three quarters nearly alone and then waits ~780 of the step's ~2300 cycles at the barrier,
// while the younger one (4-7) runs its quarters alone with its own waits exposed (~150 cycles a step)
// and arrives last.  Issuing all eight DMA pieces of a step in one quarter instead of four in each of
// two costs 5-6 %; reading the eight fragments of the two wide quarters back to back instead of one
// per two MFMAs gains 0.4-0.8 % (within box noise, not kept).
// The four-wave kernel again, as a stream with buffer-load DMA: with one asm statement per MFMA and the hooks of
// this kernel it ran 12 % behind (1131 against 1283 TFLOP/s at 12288 x 4096) -- one wave per SIMD issues one
// instruction per four cycles, so a 16-cycle MFMA leaves three slots, and hipcc's s_nop behind every asm
// statement, a fragment read and the three instructions of a DMA piece overran them.  With at most two
// instructions behind any MFMA (the DMA piece split over three MFMAs: scalar offset, M0, load) it reached this
// kernel's speed, not more: 1310-1314 / 1326-1347 / 1346-1349 / 1366 against 1299-1304 / 1344-1346 / 1348-1351
// / 1377-1379 on the four shapes.  Cycle stamps around its waits: lgkmcnt 40 cycles (the stamp's own latency),
// vmcnt 9, barrier 28 -- nothing is waited for -- while the 64 MFMAs of a chunk take 1390-1545 cycles, 22-24
// per MFMA where an unthrottled pipe takes 16 (tools/src/mfma_rate.hip: 1950-2150 TFLOP/s on operands that
// never change).  The limiter holds back MFMA issue itself; SQ_WAIT_ANY is 0.25-0.29 of the wave cycles in
// both kernels and 0.05 in the library's, whose pipe occupancy times clock is 8-10 % higher on every shape.
// What makes its flops cheaper was not found: not the LDS bytes (equal in the four-wave kernel), not the tile
// order (bands of 1 fetch 2.4 x as much past the L2 as bands of 3 and cost 5 %, so its 36 % less is worth
// about 1 %), not the operand order of consecutive MFMAs (holding src0 instead of src1 fixed: -0.7 %).
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

// The wave's 128 x 64 sub-tile is 8 (x rows) x 4 (w rows) blocks of 16 x 16; a k-step is two 32-deep
// chunks, a chunk two halves of 16 MFMAs (4 x-row blocks each against the chunk's 4 w-row blocks).
// A operand = w rows, B operand = x rows, so a lane of the result holds 4 consecutive output
// features of one token: one 8-byte store.  Fragment registers: two sets of w fragments (by chunk
// parity) and two sets of x fragments (by half); each set is loaded while the other is in use.
typedef float f32x4v __attribute__((ext_vector_type(4)));

// Persistent grid: one workgroup per CU walks its share of the tiles as ONE stream of k-steps.  The
// pieces of the next tile's first two k-steps are fetched during the last two of the current one, so a
// tile's epilogue and the next tile's first loads overlap and the LDS pipeline never drains.  The DMA
// goes through buffer loads (bounds-checked: rows past the end of an operand read as zero instead of
// being clamped per lane, so the lane offsets are the same for every tile; the tile and the k-step are
// scalar operands of the load).
typedef __amdgpu_buffer_rsrc_t nt_rsrc_t;

// EPI = 0: y = x w^T in the operands' 16-bit type.
// EPI = 1: c32[m][n] -= sum_k x[m][k] w[n][k] on an fp32 matrix (leading dimension ldc), tiles on and below the diagonal
//          only (tile list = the lower triangle, row by row; m_total == n_total): the symmetric rank-k update of the
//          Cholesky factorisation with its fp32 operands split into bf16 pieces (see gq_syrk_sub_split below).
template <int DT, int EPI>
__global__ __launch_bounds__(512) void gemm_nt_kernel(unsigned short *__restrict__ y, const unsigned short *__restrict__ x,
                                                        const unsigned short *__restrict__ w, int64_t m_total, int64_t n_total,
                                                        int64_t k_total, int tiles_n, int n_blocks, int band,
                                                        float *__restrict__ c32, int64_t ldc)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    // workgroup `orig` sits on XCD orig & 7 (round-robin dispatch) in slot orig >> 3.  An XCD owns a
    // contiguous run of the band-ordered tile list and its slots walk that run round-robin, so the tiles
    // in flight on an XCD at any time are neighbours that share x and w panels in its L2.
    int tile0, tile_end, tile_step;
    {
        const int orig = blockIdx.x, xcd = orig & 7, slot = orig >> 3;
        const int q = n_blocks >> 3, r = n_blocks & 7;
        const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        tile_step = ((int)gridDim.x + 7 - xcd) >> 3;
        tile0 = start + slot;
        tile_end = start + q + (xcd < r ? 1 : 0);
    }
    if (tile0 >= tile_end) return;   // the whole workgroup leaves: no barrier is left waiting
    const int tiles_m = EPI == 1 ? tiles_n : n_blocks / tiles_n;
    auto tile_origin = [&](int bid, int64_t &m0, int64_t &n0) __attribute__((always_inline)) {
        if constexpr (EPI == 1) {
            // Lower triangle in bands of four tile rows, column by column inside a band: the 32 tiles an XCD works on
            // at a time are then about four rows by eight columns, twelve operand panels whose current k-slices its L2
            // holds, where a row-by-row order has 33 (one x panel, 32 w panels) and every k-step of every tile goes
            // past the L2.  Band r (rows 4r .. 4r + 3) holds 16 r + 10 tiles, the bands before it 8 r^2 + 2 r.
            const int t_all = tiles_n;                 // tile rows = tile columns
            const int full = t_all >> 2;               // complete bands
            int r = (int)((sqrtf(1.0f + 8.0f * (float)bid) - 1.0f) * 0.125f);
            while (8 * (r + 1) * (r + 1) + 2 * (r + 1) <= bid) ++r;
            while (8 * r * r + 2 * r > bid) --r;
            r = min(r, full);                          // the last, partial band takes what is left
            const int base = 4 * r, nrows = min(4, t_all - base);
            int within = bid - (8 * r * r + 2 * r);
            int ti, tj;
            if (within < nrows * base) {
                tj = within / nrows;
                ti = base + (within - tj * nrows);
            } else {
                within -= nrows * base;
                int c = 0;
                while (within >= nrows - c) {
                    within -= nrows - c;
                    ++c;
                }
                tj = base + c;
                ti = base + c + within;
            }
            m0 = (int64_t)ti * NT;
            n0 = (int64_t)tj * NT;
            return;
        }
        const int per_band = band * tiles_n;
        const int b = bid / per_band, within = bid - b * per_band;
        const int rows = min(band, tiles_m - b * band);
        const int tn = within / rows;
        m0 = (int64_t)(b * band + (within - tn * rows)) * NT;
        n0 = (int64_t)tn * NT;
    };
    const int n_ktiles = (int)(k_total / NK);
    const int my_tiles = (tile_end - tile0 + tile_step - 1) / tile_step;
    const int n_steps = my_tiles * n_ktiles;   // < 2^31: checked by the launcher

    const int wr = wave >> 2, wc = wave & 3;
    const int l15 = lane & 15, lg = lane >> 4;
    const unsigned lds_base = (unsigned)(uintptr_t)GQ_LDS_PTR(char, smem);
    const int sw = (l15 >> 1) & 7;
    const unsigned xrow = lds_base + (wr * 128 + l15) * 128;               // + u * 2048, u = 0..7
    const unsigned wrow = lds_base + N_OP_BYTES + (wc * 64 + l15) * 128;   // + u * 2048, u = 0..3
    unsigned choff[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) choff[c] = (unsigned)(((4 * c + lg) ^ sw) << 4);

    f32x4v acc[4][8];   // [w block][x block]
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 8; ++b) acc[a][b] = f32x4v{0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();

    // DMA source: a buffer descriptor per operand and tile (base = the tile's first row, extent = the rest
    // of the operand), one 32-bit byte offset per piece and lane (row within the tile, swizzled chunk) and
    // the k-step as the scalar offset
    unsigned voff[4];   // piece j of either operand: rows (wave * 4 + j) * 8 + (lane >> 3)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (wave * 4 + j) * 8 + (lane >> 3);
        voff[j] = (unsigned)row * (unsigned)(k_total * 2) + (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) << 4);
    }
    auto rsrc_of = [&](const unsigned short *base, int64_t first_row, int64_t rows_total) __attribute__((always_inline)) {
        const int64_t bytes = (rows_total - first_row) * k_total * 2;
        const unsigned recs = (unsigned)(bytes < (int64_t)0xFFFFFFFFll ? bytes : (int64_t)0xFFFFFFFFll);
        return __builtin_amdgcn_make_buffer_rsrc((void *)(base + first_row * k_total), 0, (int)recs, 0x00020000);
    };
    auto dma_piece = [&](nt_rsrc_t rs, int kt, char *dst, int j) __attribute__((always_inline)) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, GQ_LDS_PTR(void, dst), 16, (int)voff[j], kt * (NK * 2), 0, 0);
    };
    // x pieces 0..3 of a k-step, then its w pieces 0..3
    auto dma_x = [&](nt_rsrc_t rx, int kt, char *stage, int j) __attribute__((always_inline)) {
        dma_piece(rx, kt, stage + (wave * 4 + j) * 1024, j);
    };
    auto dma_w = [&](nt_rsrc_t rw, int kt, char *stage, int j) __attribute__((always_inline)) {
        dma_piece(rw, kt, stage + N_OP_BYTES + (wave * 4 + j) * 1024, j);
    };

    u32x4 fw[2][4], fx[2][4];   // fw[chunk parity][w block], fx[half][x block of the half]
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
    auto half_step = [&](int ws, int half, auto hook) __attribute__((always_inline)) {
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
    auto reads_done = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };

    // fetch cursor: the (tile, k-step) two steps ahead of the one being computed
    int f_tile = tile0, f_kt = 0;
    nt_rsrc_t f_rx, f_rw;
    auto fetch_tile = [&](int t) __attribute__((always_inline)) {
        int64_t m0, n0;
        tile_origin(t, m0, n0);
        f_rx = rsrc_of(x, m0, m_total);
        f_rw = rsrc_of(w, n0, n_total);
    };
    auto fetch_advance = [&]() __attribute__((always_inline)) {
        if (++f_kt == n_ktiles) {
            f_kt = 0;
            f_tile += tile_step;
            if (f_tile < tile_end) fetch_tile(f_tile);
        }
    };
    fetch_tile(f_tile);
#pragma unroll
    for (int j = 0; j < 4; ++j) dma_x(f_rx, 0, smem, j);
#pragma unroll
    for (int j = 0; j < 4; ++j) dma_w(f_rw, 0, smem, j);
    fetch_advance();
    if (n_steps > 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) dma_x(f_rx, f_kt, smem + N_STAGE_BYTES, j);
#pragma unroll
        for (int j = 0; j < 4; ++j) dma_w(f_rw, f_kt, smem + N_STAGE_BYTES, j);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // w descriptor and k-step of the step after the one being computed (the fetch cursor's previous value)
    nt_rsrc_t p_rw = f_rw;
    int p_kt = f_kt;
    fetch_advance();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 4; ++u) read_w(0, 0u, 0, u);
#pragma unroll
    for (int u = 0; u < 4; ++u) read_x(0, 0u, 0, u);
    reads_done();

    // One k-step g of the stream.  MORE: step g+1 exists (its first fragments are read here and the w
    // pieces of it are fetched here unless the prologue did); MORE2: step g+2 exists (its x pieces are
    // fetched here).
    auto body = [&](auto more_c, auto more2_c, auto first_c, int g) __attribute__((always_inline)) {
        constexpr bool MORE = decltype(more_c)::value, MORE2 = decltype(more2_c)::value;
        constexpr bool FIRST = decltype(first_c)::value;
        const unsigned sb = (unsigned)(g & 1) * N_STAGE_BYTES, sbn = (unsigned)((g + 1) & 1) * N_STAGE_BYTES;
        char *cur = smem + (g & 1) * N_STAGE_BYTES;
        char *nxt = smem + ((g + 1) & 1) * N_STAGE_BYTES;
        // (chunk 0, half 0): x fragments of (chunk 0, half 1); the w pieces of step g+1
        half_step(0, 0, [&](int i) __attribute__((always_inline)) {
            if ((i & 1) == 0 && i < 8) read_x(1, sb, 0, i >> 1);
            if constexpr (MORE && !FIRST) {
                if ((i & 3) == 3) dma_w(p_rw, p_kt, nxt, i >> 2);
            }
        });
        reads_done();
        // (chunk 0, half 1): w fragments of chunk 1, x fragments of (chunk 1, half 0)
        half_step(0, 1, [&](int i) __attribute__((always_inline)) {
            if ((i & 1) == 0) {
                if (i < 8) read_w(1, sb, 1, i >> 1);
                else read_x(0, sb, 1, (i >> 1) - 4);
            }
        });
        reads_done();
        // (chunk 1, half 0): x fragments of (chunk 1, half 1): the last reads of this stage
        half_step(1, 0, [&](int i) __attribute__((always_inline)) { if ((i & 1) == 0 && i < 8) read_x(1, sb, 1, i >> 1); });
        reads_done();
        if constexpr (MORE) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of step g+1
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        // (chunk 1, half 1): fragments of (chunk 0, half 0) of step g+1; the x pieces of step g+2
        half_step(1, 1, [&](int i) __attribute__((always_inline)) {
            if constexpr (MORE) {
                if ((i & 1) == 0) {
                    if (i < 8) read_w(0, sbn, 0, i >> 1);
                    else read_x(0, sbn, 0, (i >> 1) - 4);
                }
            }
            if constexpr (MORE2) {
                if ((i & 3) == 3) dma_x(f_rx, f_kt, cur, i >> 2);
            }
        });
        if constexpr (MORE) reads_done();
        if constexpr (MORE2) {
            p_rw = f_rw;
            p_kt = f_kt;
            fetch_advance();
        }
    };

    // lane: token l15 of x block b, output features 4 lg .. 4 lg + 3 of w block a; 32-bit offsets from
    // the wave's corner of the tile
    auto store_tile = [&](int t) __attribute__((always_inline)) {
        int64_t m0, n0;
        tile_origin(t, m0, n0);
        if constexpr (EPI == 1) {
            // The extents are multiples of 128 (the factorisation's panel width), so a wave's 128 x 64 part of the tile is
            // inside the matrix as a whole or not at all: no per-element bounds, and the four 16-byte pieces a lane owns
            // in one row are loaded together, then written together -- one round trip to memory per row block instead of
            // one per piece (with a bounds test around every piece the compiler kept load, wait, store strictly in turn:
            // 32 dependent round trips per tile, as long as the tile's 48 k-steps).
            float *cw = c32 + (m0 + wr * 128) * ldc + n0 + wc * 64;
            const bool inside = m0 + wr * 128 < m_total && n0 + wc * 64 < n_total;   // wave-uniform
            if (inside) {
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    float *pr = cw + (int64_t)(b * 16 + l15) * ldc + 4 * lg;
                    f32x4v v[4];
#pragma unroll
                    for (int a = 0; a < 4; ++a) v[a] = *reinterpret_cast<const f32x4v *>(pr + a * 16);
#pragma unroll
                    for (int a = 0; a < 4; ++a) *reinterpret_cast<f32x4v *>(pr + a * 16) = v[a] - acc[a][b];
                }
            }
            return;
        }
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
    };
    int c_tile = tile0, c_kt = 0;   // the tile and k-step being computed
    auto step_done = [&]() __attribute__((always_inline)) {
        if (++c_kt == n_ktiles) {
            store_tile(c_tile);
            zero_acc();
            c_kt = 0;
            c_tile += tile_step;
        }
    };
    using T = std::true_type;
    using F = std::false_type;
    if (n_steps > 2) {
        body(T{}, T{}, T{}, 0);
        step_done();
        int g = 1;
        for (; g + 2 < n_steps; ++g) {
            body(T{}, T{}, F{}, g);
            step_done();
        }
        body(T{}, F{}, F{}, g);
        step_done();
        body(F{}, F{}, F{}, g + 1);
        step_done();
    } else if (n_steps == 2) {
        body(T{}, F{}, T{}, 0);
        step_done();
        body(F{}, F{}, F{}, 1);
        step_done();
    } else {
        body(F{}, F{}, T{}, 0);
        step_done();
    }
}

}  // namespace

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
    static const int grid_env = [] {   // experiment knob: workgroups of the persistent grid
        const char *e = getenv("GQ_NT_GRID");
        return e ? atoi(e) : 0;
    }();
    int dev = 0, cus = 0;
    GQ_HIP(hipGetDevice(&dev));
    GQ_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int grid = std::min(n_blocks, grid_env > 0 ? grid_env : std::max(cus, 8));
    GQ_REQUIRE((int64_t)gq_cdiv(n_blocks, grid) * (k / NK) < (1ll << 30), "gq_gemm_nt: too many k-steps per workgroup");
    auto kern = dtype == GQ_F16 ? gemm_nt_kernel<GQ_F16, 0> : gemm_nt_kernel<GQ_BF16, 0>;
    GQ_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, N_LDS));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), N_LDS, (hipStream_t)stream, (unsigned short *)y,
                       (const unsigned short *)x, (const unsigned short *)w, m, n, k, tiles_n, n_blocks, band, (float *)nullptr,
                       (int64_t)0);
    GQ_LAUNCH_CHECK();
    return GQ_OK;
}

// c[n][n] (fp32, leading dimension ldc, lower tiles) -= x w^T for bf16 x, w [n][k] (rows k-contiguous, k a multiple
// of 64, n of 128): the matrix-core product of gq_gemm_nt with an fp32 read-modify-write of c as its epilogue.
// Library-internal (csrc/hessian.hip).
int gq_gemm_nt_sub_lower_f32(float *c, int64_t ldc, const void *x, const void *w, int64_t n, int64_t k, hipStream_t stream)
{
    GQ_REQUIRE(c && x && w && n > 0 && k > 0 && k % NK == 0 && n % 128 == 0 && ldc % 4 == 0,
               "gq_gemm_nt_sub_lower_f32: n must be a multiple of 128, k of 64, ldc of 4");
    GQ_REQUIRE((((uintptr_t)x | (uintptr_t)w | (uintptr_t)c) & 15) == 0, "gq_gemm_nt_sub_lower_f32: pointers must be 16-byte aligned");
    GQ_REQUIRE(k < (1 << 21) && n < (1 << 22), "gq_gemm_nt_sub_lower_f32: at most 2^21 input and 2^22 output features");
    const int tiles = (int)gq_cdiv(n, NT);
    const int n_blocks = tiles * (tiles + 1) / 2;
    int dev = 0, cus = 0;
    GQ_HIP(hipGetDevice(&dev));
    GQ_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int grid = std::min(n_blocks, std::max(cus, 8));
    auto kern = gemm_nt_kernel<GQ_BF16, 1>;
    GQ_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, N_LDS));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), N_LDS, stream, (unsigned short *)nullptr, (const unsigned short *)x,
                       (const unsigned short *)w, n, n, k, tiles, n_blocks, 1, c, ldc);
    GQ_LAUNCH_CHECK();
    return GQ_OK;
}

extern "C" int gq_gemm_nt_sub_lower(float *c, int64_t ldc, const void *x, const void *w, int64_t n, int64_t k, gq_stream_t stream)
{
    return gq_gemm_nt_sub_lower_f32(c, ldc, x, w, n, k, (hipStream_t)stream);
}
