// The GPTQ column loop on the device (reference: gptq_py.py:69-251, gptq_cpp.cpp:33-84,
// gptq_cpp_kernel.cu:21-137) and the grid search that feeds it (quant.py:89-118).
//
// Working layout: everything is kept transposed, W^T as [C][R] fp32 ("one row per input
// feature"), as the reference's native loop does.  One thread owns one output feature r, so
// every access to column c of W is a coalesced read of row c of W^T and rows of W never
// interact.  A block of up to 128 consecutive input features is processed per launch:
//   * the thread loads its <=128 weights into registers (fully unrolled, static indices),
//   * walks the columns in order: quantise, dequantise, error, and immediately applies the
//     rank-1 update to the not-yet-quantised registers; the needed entries U[c][c'] are the same
//     for every thread and come in through scalar loads,
//   * nothing is written back but the outputs (q, dequantised value, scaled error).
// The update of the columns behind the block is an fp32 MFMA GEMM (sgemm.hip) whose per-element
// operation order (k ascending, fused multiply-add, starting from the old weight) is the same as
// the in-block updates', so the result does not depend on how columns are cut into blocks.
#include <stdlib.h>

#include <type_traits>
#include <vector>

#include "sgemm.h"

namespace {

constexpr int QB = 128;   // max columns per block launch (registers per thread)
struct QuantBlockArgs {
    float *wt;               // [C][R] weights (read; rows a..a+bs-1 optionally written back)
    float *quant_t;          // [C][R] or NULL
    float *error_t;          // [C][R]
    void *q_out;             // [C][R] int16 or uint8, or NULL
    const float *u;          // [C][C]
    float *scale;            // [R] grid of the group this block lies in (read, or written if dynamic)
    float *zero;             // [R]
    int64_t c, r;
    int a, bs;
    float maxq;
    int q_is_u8;
    int write_back;          // store the (updated) weights of the block back to wt
    int dynamic;             // compute the grid here, from the block's weights as they are now
    int sym;
};

__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fmaxf(lo, fminf(hi, x)); }

__device__ __forceinline__ void grid_from_minmax(float lo, float hi, float amax, float maxq, int sym, float &scale,
                                                 float &zero)
{
    if (sym) {
        scale = amax * (2.0f / maxq) + 1e-12f;
        zero = rintf((maxq + 1.0f) * 0.5f);
    } else {
        const float xmax = fmaxf(hi, 0.0f), xmin = fminf(lo, 0.0f);
        scale = (xmax - xmin) / maxq + 1e-12f;
        zero = rintf(-xmin / scale);
    }
}

// Correctly rounded a / b from rb = RN(1 / b): q0 = RN(a * rb) is within one ulp of the quotient,
// the residual a - b * q0 is exact in one fma, and one more fma lands on the correctly rounded
// quotient (the classic fma division; three dependent instructions instead of the ~14 of the
// IEEE sequence hipcc emits for `/`).  Its proof needs a correctly rounded reciprocal and no
// underflow in the residual, so: the reciprocals are taken in double precision and rounded once
// (exact for a 24-bit divisor, see recip_rn), and a sub-block that met an operand outside
// 2^-97 .. 2^93 -- or a divisor outside 2^-60 .. 2^60, flagged by rb = NaN -- is walked again with
// the plain division (its inputs are still untouched in LDS).
__device__ __forceinline__ float recip_rn(float b)
{
    // RN(1/b): the double quotient carries 53 bits; it cannot sit within 2^-53 of a 25-bit midpoint
    // m unless b * m is a power of two, so rounding it again to 24 bits never double-rounds.
    const unsigned e = (__float_as_uint(b) >> 23) & 0xffu;
    if (e < 67u || e > 187u) return __uint_as_float(0x7fc00000u);
    return (float)(1.0 / (double)b);
}

// fast form, no branch.  The operands the proof does not cover -- a dividend that is neither zero nor within
// 2^-97 .. 2^93, a reciprocal that is NaN -- are found afterwards from three running extrema of the operands'
// magnitudes taken as integers (four cheap instructions per division beside the chain; a comparison of the
// exponent field with a zero test per operand cost twelve).
struct DivRange {
    unsigned mx = 0u;            // largest |a|
    unsigned mn1 = 0xffffffffu;  // smallest |a| - 1 (unsigned: a zero wraps to the top and never wins)
    unsigned rb = 0u;            // largest |rb| (a NaN lies above infinity)
    __device__ __forceinline__ bool bad() const
    {
        return mx >= (221u << 23) || mn1 < (30u << 23) - 1u || rb > 0x7f800000u;
    }
};

__device__ __forceinline__ float div_rn_fast(float a, float b, float rb, DivRange &seen)
{
    const unsigned ai = __float_as_uint(a) & 0x7fffffffu;
    seen.mx = max(seen.mx, ai);
    seen.mn1 = min(seen.mn1, ai - 1u);
    seen.rb = max(seen.rb, __float_as_uint(rb) & 0x7fffffffu);
    const float q0 = __fmul_rn(a, rb);
    const float rem = fmaf(-b, q0, a);
    return fmaf(rem, rb, q0);
}

// SEM == GQ_SEM_KERNEL: fp16 dequantisation, fused update.  GQ_SEM_PYTHON: fp32, mul then sub.
//
// One 256-thread workgroup owns 64 output features (lane = feature) and one block of <= 128
// columns.  LDS holds the block's weights wl[column][lane] (one 4-byte column per lane: conflict-
// free) and the block's square of U, ul[c][c'], whose reads are the same address in every lane
// (broadcast).  SUB columns at a time are pulled into registers and walked sequentially by ALL
// four waves redundantly (the other SIMDs would idle anyway, and every wave then owns the SUB
// scaled errors without any exchange); the rank-SUB update of the rest of the block is split
// over the waves SUB columns at a time.  Every weight still sees its updates in ascending column
// order, one fma each, so the cut into sub-blocks does not change any bit.
constexpr int ULD = QB + 4;   // stride of the U square in LDS
// (Shrinking this to 96 KiB -- no padding, the reciprocals in registers -- so that a workgroup fits on a CU beside
// one 64 KiB workgroup of another slice's update GEMM was measured: column loops of a block 16.7 -> 17.8 ms.)
constexpr int QB_LDS = (QB * 64 + QB * ULD + QB) * 4;   // weights, U square, reciprocals of its diagonal

#ifdef GQ_QB_TRACE
// Debug build only (tools/trace_qb.py): cycle stamps of workgroup 0.
__device__ unsigned long long *g_qb_trace = nullptr;
#define QB_TR(i) ts[i] = __builtin_readcyclecounter()
#else
#define QB_TR(i)
#endif

template <int SEM, int SUB>
__global__ __launch_bounds__(256) void quant_block_kernel(
    float *__restrict__ wt, float *__restrict__ quant_t, float *__restrict__ error_t, void *__restrict__ q_out,
    const float *__restrict__ u, float *__restrict__ scale, float *__restrict__ zero, int64_t ldu, int64_t r_total,
    int a, int bs, float maxq, int q_is_u8, int write_back, int dynamic, int sym)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
#ifdef GQ_QB_TRACE
    unsigned long long ts[40] = {};
    QB_TR(0);
#endif
    float (*wl)[64] = reinterpret_cast<float (*)[64]>(smem_raw);
    float *ul = reinterpret_cast<float *>(smem_raw) + QB * 64;
    float *rdl = ul + QB * ULD;   // RN(1 / U[c][c]) of the block's columns
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t r = (int64_t)blockIdx.x * 64 + lane;
    const bool live = r < r_total;
    const int64_t rr = live ? r : r_total - 1;   // dead lanes shadow the last row and store nothing
    // Everything the block needs, fetched with all loads in flight at once (a load-store loop would
    // pay one memory latency per iteration): the wave's 32 weight columns, the thread's 16 quads
    // of the U square, the grid.
    // Addresses are a wave-uniform 64-bit base plus a 32-bit per-lane offset (the host checks that
    // 128 columns of R rows and 128 rows of U stay below 2^30 elements): 64-bit multiplies per
    // load cost more than the loads.
    float s = 0.0f, z = 0.0f;
    if (!dynamic) {
        s = scale[rr];
        z = zero[rr];
    }
    const unsigned rt = (unsigned)r_total, lu = (unsigned)ldu;
    const float *__restrict__ wbase = wt + (int64_t)a * r_total;
    const float *__restrict__ ubase = u + (int64_t)a * ldu + a;
    {
        // (U first: behind the weight loads hipcc forms these addresses with an instruction that names a
        // register still being loaded, and waits for it)
        // The loads of the U square are unconditional, from addresses clamped into the square, and
        // masked afterwards: a load under a divergent branch makes hipcc drain every older load first
        // (vmcnt(0) in front of each), which turned these 16 loads into 16 round trips.
        const int trow = tid >> 5, c4 = (tid & 31) * 4;
        const unsigned c4c = c4 < bs ? (unsigned)c4 : 0u;
        f32x4 ureg[QB / 8];
#pragma unroll
        for (int i = 0; i < QB / 8; ++i) {
            const int row = trow + 8 * i;
            const unsigned rowc = row < bs ? (unsigned)row : 0u;
            ureg[i] = *reinterpret_cast<const f32x4 *>(ubase + (rowc * lu + c4c));
        }
        const unsigned tc = tid < bs ? (unsigned)tid : 0u;
        float dg = ubase[tc * lu + tc];
        float wreg[QB / 4];
#pragma unroll
        for (int i = 0; i < QB / 4; ++i) {
            const int c = wv + 4 * i;
            wreg[i] = c < bs ? wbase[(unsigned)c * rt + (unsigned)rr] : 0.0f;
        }
        __builtin_amdgcn_sched_barrier(0);   // every load is issued before the first result is touched
#pragma unroll
        for (int i = 0; i < QB / 8; ++i)
            if (!(trow + 8 * i < bs && c4 < bs)) ureg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (tid >= bs) dg = 1.0f;
        QB_TR(34);
#pragma unroll
        for (int i = 0; i < QB / 4; ++i) wl[wv + 4 * i][lane] = wreg[i];
#pragma unroll
        for (int i = 0; i < QB / 8; ++i) *reinterpret_cast<f32x4 *>(ul + (trow + 8 * i) * ULD + c4) = ureg[i];
        QB_TR(35);
        if (tid < bs) rdl[tid] = recip_rn(dg);
        QB_TR(36);
    }
    __syncthreads();
    QB_TR(37);

    if (dynamic) {
        float lo = 3.4e38f, hi = -3.4e38f, am = 0.0f;
        for (int c = 0; c < bs; ++c) {
            const float v = wl[c][lane];
            lo = fminf(lo, v);
            hi = fmaxf(hi, v);
            am = fmaxf(am, fabsf(v));
        }
        grid_from_minmax(lo, hi, am, maxq, sym, s, z);
        if (live && wv == 0) {
            scale[r] = s;
            zero[r] = z;
        }
    }
    const float rs = recip_rn(s);
    QB_TR(1);

    const int n_sub = bs / SUB;
    for (int jb = 0; jb < n_sub; ++jb) {
        const int c0 = jb * SUB;   // first column of the sub-block
        float w[SUB], nerr[SUB], deq[SUB], qv[SUB];
        // FAST: divisions by fma from precomputed reciprocals, straight-line; else the plain ones
        auto walk = [&](auto fast_c) -> bool {
            constexpr bool FAST = decltype(fast_c)::value;
            DivRange bad;
#pragma unroll
            for (int i = 0; i < SUB; ++i) w[i] = wl[c0 + i][lane];
#pragma unroll
            for (int i = 0; i < SUB; ++i) {
                const float x = w[i];
                float quo;
                if constexpr (FAST) quo = div_rn_fast(x, s, rs, bad);
                else quo = x / s;
                const float qf = clampf(rintf(quo) + z, 0.0f, maxq);
                float dq;
                if constexpr (SEM == GQ_SEM_KERNEL) {
                    const _Float16 hx = (_Float16)qf, hz = (_Float16)z, hs = (_Float16)s;
                    _Float16 hd = hx - hz;
                    hd = hd * hs;
                    dq = (float)hd;
                } else {
                    dq = __fmul_rn(__fsub_rn(qf, z), s);
                }
                const float d = ul[(c0 + i) * ULD + c0 + i];
                float err;
                if constexpr (FAST) err = div_rn_fast(x - dq, d, rdl[c0 + i], bad);
                else err = (x - dq) / d;
                qv[i] = qf;
                deq[i] = dq;
                nerr[i] = SEM == GQ_SEM_KERNEL ? -err : err;
#pragma unroll
                for (int i2 = i + 1; i2 < SUB; ++i2) {
                    const float uv = ul[(c0 + i) * ULD + c0 + i2];
                    if constexpr (SEM == GQ_SEM_KERNEL) w[i2] = fmaf(nerr[i], uv, w[i2]);
                    else w[i2] = __fsub_rn(w[i2], __fmul_rn(nerr[i], uv));
                }
            }
            return bad.bad();
        };
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(walk(std::true_type{})) != 0, 0)) walk(std::false_type{});
#ifdef GQ_QB_TRACE
        if (jb < 8) QB_TR(2 + jb * 4);
#endif
        // outputs of the sub-block: every wave holds all of them, wave w stores columns i = w mod 4
        if (live) {
            const int64_t col0 = (int64_t)(a + c0) * r_total;   // wave-uniform
            float *__restrict__ eb = error_t + col0;
            float *__restrict__ qb = quant_t ? quant_t + col0 : nullptr;
            float *__restrict__ wb = wt + col0;
            uint8_t *__restrict__ q8 = (uint8_t *)q_out + col0;
            int16_t *__restrict__ q16 = (int16_t *)q_out + col0;
            // one copy of the store code per wave with its columns as constants: with the wave as a run-time
            // test inside the loop over the 16 columns every column cost a scalar branch and the flags were
            // tested again for each (1.5k of a sub-block's 8k cycles)
            auto store_columns = [&](auto wv_c) __attribute__((always_inline)) {
                constexpr int W = decltype(wv_c)::value;
#pragma unroll
                for (int i = W; i < SUB; i += 4) {
                    const unsigned o = (unsigned)i * rt + (unsigned)r;
                    eb[o] = SEM == GQ_SEM_KERNEL ? -nerr[i] : nerr[i];
                }
                if (quant_t) {
#pragma unroll
                    for (int i = W; i < SUB; i += 4) qb[(unsigned)i * rt + (unsigned)r] = deq[i];
                }
                if (write_back) {
#pragma unroll
                    for (int i = W; i < SUB; i += 4) wb[(unsigned)i * rt + (unsigned)r] = w[i];
                }
                if (q_out) {
                    if (q_is_u8) {
#pragma unroll
                        for (int i = W; i < SUB; i += 4) q8[(unsigned)i * rt + (unsigned)r] = (uint8_t)qv[i];
                    } else {
#pragma unroll
                        for (int i = W; i < SUB; i += 4) q16[(unsigned)i * rt + (unsigned)r] = (int16_t)(uint16_t)qv[i];
                    }
                }
            };
            if (wv == 0) store_columns(std::integral_constant<int, 0>{});
            else if (wv == 1) store_columns(std::integral_constant<int, 1>{});
            else if (wv == 2) store_columns(std::integral_constant<int, 2>{});
            else store_columns(std::integral_constant<int, 3>{});
        }
#ifdef GQ_QB_TRACE
        if (jb < 8) QB_TR(3 + jb * 4);
#endif
        if constexpr (SEM == GQ_SEM_KERNEL) {
            // The rest of the block on the matrix core: W^T[c'][r] += sum_k U[c0+k][c'] * nerr[k][r]
            // as 32 (columns) x 32 (rows) output blocks of v_mfma_f32_32x32x2_f32 -- per element the
            // same ascending-k chain of fused multiply-adds as the scalar form.  Column blocks are
            // aligned to 32, so the first one may cover columns that are already final: those are
            // computed and not stored (a slower wave may still be reading them).  A operand: U from LDS.  B operand: lane l
            // holds nerr[.] of row l; the swap below puts (k even, rows 0..31 | k odd, rows 0..31)
            // into the lanes of one register and the same for rows 32..63 into the other.
            const int li = lane & 31, lh = lane >> 5;
            float b0[SUB / 2], b1[SUB / 2];
#pragma unroll
            for (int sidx = 0; sidx < SUB / 2; ++sidx) {
                const auto pr = __builtin_amdgcn_permlane32_swap(__float_as_uint(nerr[2 * sidx]),
                                                                 __float_as_uint(nerr[2 * sidx + 1]), false, false);
                b0[sidx] = __uint_as_float(pr[0]);
                b1[sidx] = __uint_as_float(pr[1]);
            }
            const int cb_lo = (c0 + SUB) >> 5, n_cb = (bs + 31) >> 5;
            // (two of a wave's tasks as one stream -- reads of both first, MFMA chains interleaved -- was measured
            // slower: 2.4k against 1.75k cycles per sub-block)
            for (int t = wv; t < (n_cb - cb_lo) * 2; t += 4) {
                const int cblk = cb_lo + (t >> 1), rb = t & 1;
                f32x16 acc;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] = wl[cblk * 32 + 8 * (e >> 2) + (e & 3) + 4 * lh][rb * 32 + li];
#pragma unroll
                for (int sidx = 0; sidx < SUB / 2; ++sidx) {
                    const float av = ul[(c0 + 2 * sidx + lh) * ULD + cblk * 32 + li];
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, rb ? b1[sidx] : b0[sidx], acc, 0, 0, 0);
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int col = cblk * 32 + 8 * (e >> 2) + (e & 3) + 4 * lh;
                    if (col >= c0 + SUB) wl[col][rb * 32 + li] = acc[e];   // columns already final stay as they are
                }
            }
        } else {
            // the rest of the block, SUB columns at a time, one chunk per wave
            for (int cb = jb + 1 + wv; cb < n_sub; cb += 4) {
                const int d0 = cb * SUB;
                float v[SUB];
#pragma unroll
                for (int j = 0; j < SUB; ++j) v[j] = wl[d0 + j][lane];
#pragma unroll
                for (int k = 0; k < SUB; ++k) {
#pragma unroll
                    for (int j4 = 0; j4 < SUB; j4 += 4) {
                        const f32x4 uv = *reinterpret_cast<const f32x4 *>(ul + (c0 + k) * ULD + d0 + j4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[j4 + e] = __fsub_rn(v[j4 + e], __fmul_rn(nerr[k], uv[e]));
                    }
                }
#pragma unroll
                for (int j = 0; j < SUB; ++j) wl[d0 + j][lane] = v[j];
            }
        }
#ifdef GQ_QB_TRACE
        if (jb < 8) QB_TR(4 + jb * 4);
#endif
        __syncthreads();
#ifdef GQ_QB_TRACE
        if (jb < 8) QB_TR(5 + jb * 4);
#endif
    }
#ifdef GQ_QB_TRACE
    if (g_qb_trace && blockIdx.x == 0 && lane == 0) {
#pragma unroll
        for (int i = 0; i < 38; ++i) g_qb_trace[wv * 40 + i] = ts[i];
    }
#endif
}

#ifdef GQ_QB_TRACE
extern "C" int gq_debug_qb_trace(unsigned long long *buf)
{
    GQ_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_qb_trace), &buf, sizeof(buf)));
    return GQ_OK;
}
#endif

// ---- a super-block of up to 512 columns in ONE launch (native-loop semantics, grids given) ------
// The block kernel above is launched once per <= 128 columns and a GEMM launch between two of them
// carries the errors into the next block: per 512 columns that is a chain of seven dependent launches
// of 25-32 + 9 us each.  Here a workgroup owns 32 output features for a window of up to 512 columns,
// held in LDS, and never returns to the host inside it:
//   * wave 0 (the chain) walks 16 columns at a time, exactly as the block kernel walks a sub-block
//     (same instructions per column, so the same bits);
//   * the rank-16 update those 16 columns owe every later column of the window runs on the matrix
//     core as 32-column tiles: the tile that holds the next two sub-blocks by wave 0 itself right
//     before it walks on, everything behind it by waves 1-3 WHILE wave 0 walks the next sub-block;
//   * one barrier per 16 columns.  Update j-1 reaches sub-block j+1 one phase before wave 0 applies
//     update j to it, so every weight still receives its updates in ascending column order, each one
//     fused multiply-add, starting from the old weight: the cut into tiles and phases changes no bit.
// U is not staged: the chain's 16 x 16 piece goes through LDS (fetched one phase ahead by wave 3), the
// tiles' operands come from memory straight into registers (every workgroup reads the same rows of
// U at about the same time, so they are served by the XCD's L2).  Waves 1-3 also write the finished
// sub-block's outputs (errors, dequantised values) with 16-byte stores; the levels leave as 16 bytes
// per row and sub-block into a blocked layout [C/16][R][16] that unblock_q_kernel turns into [R][C].
constexpr int SW = 512;       // columns of a window
constexpr int SR = 32;        // output features per workgroup
constexpr int SB = 16;        // columns per phase
constexpr int SUS = 20;       // stride of a 16 x 16 piece of U in LDS (rows stay 16-byte aligned)
constexpr int SW_PAD = 32;    // columns behind the window that the last tiles may touch (never stored)
struct SuperTable {
    float maxq[SW / SB];        // per sub-block: 2^bits - 1 of its group
    uint16_t group[SW / SB];    // per sub-block: its group
};
constexpr int QS_LDS = ((SW + SW_PAD) * SR + 2 * SB * SR + 2 * SB * SUS + 2 * SB + 2 * 2 * SR) * 4 + 2 * SR * 16;

#ifdef GQ_QS_TRACE
// Debug build only (tools/trace_qs.py): cycle stamps of workgroup 0, [wave][phase < 12][5].
__device__ unsigned long long *g_qs_trace = nullptr;
#define QS_TR(k)                                                                                   \
    do {                                                                                           \
        if (g_qs_trace && blockIdx.x == 0 && lane == 0 && j < 12)                                  \
            g_qs_trace[(wv * 12 + j) * 5 + (k)] = __builtin_readcyclecounter();                    \
    } while (0)
#else
#define QS_TR(k)
#endif

// A workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every global load and
// store the wave has in flight; here loads are requested a phase ahead on purpose and the stores of outputs
// need no ordering inside the kernel.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__global__ __launch_bounds__(256, 2) void quant_super_kernel(
    const float *__restrict__ wt, float *__restrict__ quant_t, float *__restrict__ error_t, uint8_t *__restrict__ q_blk,
    const float *__restrict__ u, int64_t ldu, const float *__restrict__ scale_g, const float *__restrict__ zero_g,
    int64_t r_total, int a, int n, SuperTable tab)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float *win = reinterpret_cast<float *>(smem_raw);   // [SW + SW_PAD][SR]  win[c][row]
    float *ebuf = win + (SW + SW_PAD) * SR;             // [2][SB][SR]  -error of the last walked sub-block
    float *usm = ebuf + 2 * SB * SR;                    // [2][SB][SUS] U[sj + i][sj + i2]
    float *rdl = usm + 2 * SB * SUS;                    // [2][SB]      RN(1 / U[sj + i][sj + i])
    float *szb = rdl + 2 * SB;                          // [2][2][SR]   scale, zero of the last two groups
    uint32_t *qbuf = reinterpret_cast<uint32_t *>(szb + 4 * SR);   // [2][SR][4] levels, one byte each
    const int tid = threadIdx.x;
    const int lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t r0 = (int64_t)blockIdx.x * SR;
    const int64_t r = r0 + li;
    const bool live = r < r_total;
    const int64_t rr = live ? r : r_total - 1;          // dead lanes shadow the last row and store nothing
    const int ns = n / SB;
    const unsigned rt = (unsigned)r_total, lu = (unsigned)ldu;
    const float *__restrict__ wbase = wt + (int64_t)a * r_total;
    const float *__restrict__ ubase = u + (int64_t)a * ldu + a;   // U[a + i][a + c]
    const int cmax = (int)(ldu - a) - 1;                // last column of U, counted from a

    // U[sj + i][sj .. sj + 15] and the reciprocals of its diagonal for sub-block jn: one 16-byte load per lane
    auto fetch_usm = [&](int jn) {
        const int i = lane >> 2, c4 = (lane & 3) * 4, sjn = jn * SB;
        const f32x4 v = *reinterpret_cast<const f32x4 *>(ubase + ((unsigned)(sjn + i) * lu + (unsigned)(sjn + c4)));
        *reinterpret_cast<f32x4 *>(usm + (jn & 1) * SB * SUS + i * SUS + c4) = v;
        if ((i >> 2) == (lane & 3)) {
            const int e = i & 3;
            const float dg = e == 0 ? v[0] : e == 1 ? v[1] : e == 2 ? v[2] : v[3];
            rdl[(jn & 1) * SB + i] = recip_rn(dg);
        }
    };

    // ---- prologue: the window, the first piece of U, the first grid ------------------------------
    float s_nx = 0.0f, z_nx = 0.0f;
    {
        const int sub = lane >> 3, q4 = (lane & 7) * 4;
        int64_t r4 = r0 + q4;
        if (r4 > r_total - 4) r4 = r_total - 4;         // (r_total is a multiple of 4)
        f32x4 v[SW / 32];
#pragma unroll
        for (int i = 0; i < SW / 32; ++i) {
            int c = wv * 8 + sub + 32 * i;
            if (c > n - 1) c = n - 1;
            v[i] = *reinterpret_cast<const f32x4 *>(wbase + ((unsigned)c * rt + (unsigned)r4));
        }
        if (wv == 0) {
            const int g0 = tab.group[0];
            s_nx = scale_g[(int64_t)g0 * r_total + rr];
            z_nx = zero_g[(int64_t)g0 * r_total + rr];
        }
        if (wv == 3) fetch_usm(0);
#pragma unroll
        for (int i = 0; i < SW / 32; ++i) {
            const int c = wv * 8 + sub + 32 * i;
            if (c < n) *reinterpret_cast<f32x4 *>(win + c * SR + q4) = v[i];
        }
    }
    __syncthreads();

    if (wv == 0) {
        // ---- the chain ---------------------------------------------------------------------------
        // The tile of sub-blocks j and j+1 lives in the matrix core's accumulator layout: register e of lane
        // (li, lh) is column 8 (e / 4) + e % 4 + 4 lh of the tile, row li.  Its second half stays in
        // registers as the first half of the next phase's tile, so only the 16 columns that enter come from LDS.
        float nerr[SB];           // minus the scaled errors of the sub-block walked last (both half-waves hold row li)
        float a_la[SB / 2];       // U[rows of that sub-block][the 32 columns behind it], requested a phase ahead
        float keep[8];            // second half of the last tile
#pragma unroll
        for (int i = 0; i < SB; ++i) nerr[i] = 0.0f;
#pragma unroll
        for (int i = 0; i < SB / 2; ++i) a_la[i] = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; ++i) keep[i] = 0.0f;
        float s = 1.0f, z = 0.0f, rs = 1.0f;
        int g_cur = -1;
        for (int j = 0; j <= ns; ++j) {
            const int sj = j * SB;
            QS_TR(0);
            if (j < ns) {
                // update j-1 for the tile that holds sub-blocks j and j+1
                f32x16 acc;
#pragma unroll
                for (int e = 8; e < 16; ++e) acc[e] = win[(sj + 8 * (e >> 2) + (e & 3) + 4 * lh) * SR + li];
                if (j == 0) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] = win[(8 * (e >> 2) + (e & 3) + 4 * lh) * SR + li];
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] = keep[e];
#pragma unroll
                    for (int k2 = 0; k2 < SB / 2; ++k2) {
                        // (both values pinned to registers first: a select between two elements of an array becomes
                        // an indexed load, which puts the whole array into scratch memory)
                        float e0 = nerr[2 * k2], e1 = nerr[2 * k2 + 1];
                        asm volatile("" : "+v"(e0), "+v"(e1));
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_la[k2], lh ? e1 : e0, acc, 0, 0, 0);
                    }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) keep[e] = acc[8 + e];
                // the 16 columns to walk, all of them in every lane: the half-waves exchange their columns
                float w0[SB];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const auto pr = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[e]), __float_as_uint(acc[e]), false, false);
                    w0[8 * (e >> 2) + (e & 3)] = __uint_as_float(pr[0]);
                    w0[8 * (e >> 2) + (e & 3) + 4] = __uint_as_float(pr[1]);
                }
                QS_TR(1);
                // operands of the next such update, and the next group's grid, on their way during the walk
                if (j + 1 < ns) {
                    int col = sj + SB + li;
                    if (col > cmax) col = cmax;
#pragma unroll
                    for (int k2 = 0; k2 < SB / 2; ++k2) a_la[k2] = ubase[(unsigned)(sj + 2 * k2 + lh) * lu + (unsigned)col];
                }
                const int g = tab.group[j];
                const float maxq = tab.maxq[j];
                if (g != g_cur) {
                    s = s_nx;
                    z = z_nx;
                    rs = recip_rn(s);
                    g_cur = g;
                    if (lh == 0) {
                        szb[(g & 1) * 2 * SR + li] = s;
                        szb[(g & 1) * 2 * SR + SR + li] = z;
                    }
                }
                if (j + 1 < ns && tab.group[j + 1] != g) {
                    const int g1 = tab.group[j + 1];
                    s_nx = scale_g[(int64_t)g1 * r_total + rr];
                    z_nx = zero_g[(int64_t)g1 * r_total + rr];
                }
                const float rdv = rdl[(j & 1) * SB + (lane & 15)];
                const float *__restrict__ us = usm + (j & 1) * SB * SUS;
                // fp16 dequantisation as one fused operation: (q - z) is exact in binary16 and so is z * hs in
                // binary32, so RN16(q * hs - z * hs) is the reference's RN16((q - z) * hs); an hs that is not finite
                // sends the sub-block down the plain path
                const float hs32 = (float)(_Float16)s, nzhs = -((float)(_Float16)z * hs32);
                const bool odd_grid = !(fabsf(hs32) <= 65504.0f);
                float qv[SB];
                auto walk = [&](auto fast_c) -> bool {
                    constexpr bool FAST = decltype(fast_c)::value;
                    DivRange bad;
                    float w[SB];
#pragma unroll
                    for (int i = 0; i < SB; ++i) w[i] = w0[i];
                    // Row i+1 of the U piece comes out of LDS 16 bytes at a time (the same address in every lane) while
                    // column i's updates run: the walk never holds more than two rows in registers.
                    f32x4 ur[SB / 4], un[SB / 4];
#pragma unroll
                    for (int q = 0; q < SB / 4; ++q) ur[q] = *reinterpret_cast<const f32x4 *>(us + 4 * q);
#pragma unroll
                    for (int i = 0; i < SB; ++i) {
                        const float x = w[i];
                        const float d = ur[i / 4][i % 4];
                        const float rdi = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(rdv), i));
                        float qf, diff;
                        if constexpr (FAST) {
                            const float quo = div_rn_fast(x, s, rs, bad);
                            qf = __builtin_amdgcn_fmed3f(rintf(quo) + z, 0.0f, maxq);   // operands are finite here: a clamp
                            float dq16;
                            asm("v_fma_mixlo_f16 %0, %1, %2, %3" : "=v"(dq16) : "v"(qf), "v"(hs32), "v"(nzhs));
                            asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(diff) : "v"(dq16), "v"(x));
                        } else {
                            qf = clampf(rintf(x / s) + z, 0.0f, maxq);
                            const _Float16 hx = (_Float16)qf, hz = (_Float16)z, hs = (_Float16)s;
                            _Float16 hd = hx - hz;
                            hd = hd * hs;
                            diff = x - (float)hd;
                        }
                        float err;
                        if constexpr (FAST) err = div_rn_fast(diff, d, rdi, bad);
                        else err = diff / d;
                        qv[i] = qf;
                        nerr[i] = -err;
                        // (the reads go out here, in front of the updates, which write no new register: issued in front
                        // of the chain above, their destination registers were reused for its temporaries at once and
                        // every column waited for LDS)
                        __builtin_amdgcn_sched_barrier(0);
                        if (i + 1 < SB) {
#pragma unroll
                            for (int q = (i + 1) / 4; q < SB / 4; ++q) un[q] = *reinterpret_cast<const f32x4 *>(us + (i + 1) * SUS + 4 * q);
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int i2 = i + 1; i2 < SB; ++i2) w[i2] = fmaf(nerr[i], ur[i2 / 4][i2 % 4], w[i2]);
                        __builtin_amdgcn_sched_barrier(0);
                        // (every component of what was read counts as used until here: the ones in front of the diagonal
                        // are not needed, and a register that is dead while its load is in flight is handed to the next
                        // temporary, whose write then waits for the load)
#pragma unroll
                        for (int q = (i + 1) / 4; q < SB / 4; ++q) asm volatile("" ::"v"(un[q]));
#pragma unroll
                        for (int q = (i + 1) / 4; q < SB / 4; ++q) ur[q] = un[q];
                    }
                    return bad.bad() || odd_grid;
                };
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(walk(std::true_type{})) != 0, 0)) walk(std::false_type{});
                QS_TR(2);
                // what the other waves need of this sub-block: -errors (the half-waves write alternate columns),
                // levels (one byte each)
                u32x4 qp = {0u, 0u, 0u, 0u};
#pragma unroll
                for (int i = 0; i < SB; ++i) qp[i >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(qv[i], i & 3, qp[i >> 2]);
                float *__restrict__ eb = ebuf + (j & 1) * SB * SR;
#pragma unroll
                for (int k2 = 0; k2 < SB / 2; ++k2) {
                    float e0 = nerr[2 * k2], e1 = nerr[2 * k2 + 1];
                    asm volatile("" : "+v"(e0), "+v"(e1));
                    eb[(2 * k2 + lh) * SR + li] = lh ? e1 : e0;
                }
                if (lh == 0) {
                    *reinterpret_cast<u32x4 *>(qbuf + ((j & 1) * SR + li) * 4) = qp;
                    if (live) *reinterpret_cast<u32x4 *>(q_blk + ((int64_t)((a + sj) >> 4) * r_total + r) * 16) = qp;
                }
            } else {
                QS_TR(1);
                QS_TR(2);
            }
            QS_TR(3);
            lds_barrier();
            QS_TR(4);
        }
    } else {
        // ---- waves 1-3: update j-1 for the tiles behind the one wave 0 has just taken, tile t on wave 1 + t % 3 ----
        // A tile's piece of U comes from memory straight into registers, and memory is a microsecond away: the
        // pieces of update j (applied next phase; U does not change) are requested before update j-1 is applied
        // with the pieces requested a phase ago.
        constexpr int ST = (SW / 32 + 2) / 3;   // tiles of one update a wave may own
        float a_cur[ST][SB / 2], a_nxt[ST][SB / 2];
#pragma unroll
        for (int slot = 0; slot < ST; ++slot)
#pragma unroll
            for (int k2 = 0; k2 < SB / 2; ++k2) a_cur[slot][k2] = a_nxt[slot][k2] = 0.0f;
        for (int j = 0; j <= ns; ++j) {
            const int sj = j * SB;
            QS_TR(0);
            const int jp = j - 1, sp = jp * SB;
            const float *__restrict__ eb = ebuf + (jp & 1) * SB * SR;
            const int c_first = sj + 2 * SB;
            const int nt = (j > 0 && n > c_first) ? (n - c_first + 31) >> 5 : 0;
            const int c_next = sj + 3 * SB;
            const int nt_next = (j < ns && n > c_next) ? (n - c_next + 31) >> 5 : 0;
#pragma unroll
            for (int slot = 0; slot < ST; ++slot) {
                const int t = wv - 1 + 3 * slot;
                if (t < nt_next) {
                    int col = c_next + 32 * t + li;
                    if (col > cmax) col = cmax;
#pragma unroll
                    for (int k2 = 0; k2 < SB / 2; ++k2) a_nxt[slot][k2] = ubase[(unsigned)(sj + 2 * k2 + lh) * lu + (unsigned)col];
                }
            }
            QS_TR(1);
            if (wv - 1 < nt) {
                float b[SB / 2];
#pragma unroll
                for (int k2 = 0; k2 < SB / 2; ++k2) b[k2] = eb[(2 * k2 + lh) * SR + li];
#pragma unroll
                for (int slot = 0; slot < ST; ++slot) {
                    const int t = wv - 1 + 3 * slot;
                    if (t < nt) {
                        const int c0 = c_first + 32 * t;
                        f32x16 acc;
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[e] = win[(c0 + 8 * (e >> 2) + (e & 3) + 4 * lh) * SR + li];
#pragma unroll
                        for (int k2 = 0; k2 < SB / 2; ++k2)
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[slot][k2], b[k2], acc, 0, 0, 0);
#pragma unroll
                        for (int e = 0; e < 16; ++e) win[(c0 + 8 * (e >> 2) + (e & 3) + 4 * lh) * SR + li] = acc[e];
                    }
                }
            }
            QS_TR(2);
            // outputs of sub-block j-1, by the wave with the fewest tiles: 16 bytes (4 rows of a column) per lane
            if (j > 0 && wv == 1 + nt % 3) {
                const int g = tab.group[jp];
                const float *__restrict__ sz = szb + (g & 1) * 2 * SR;
                const uint32_t *__restrict__ qb = qbuf + (jp & 1) * SR * 4;
                const int i8 = lane >> 3, q4 = (lane & 7) * 4;
                const bool ok = r0 + q4 < r_total;
                const f32x4 sv = *reinterpret_cast<const f32x4 *>(sz + q4);
                const f32x4 zv = *reinterpret_cast<const f32x4 *>(sz + SR + q4);
#pragma unroll
                for (int pass = 0; pass < 2; ++pass) {
                    const int i = pass * 8 + i8;
                    const f32x4 ne = *reinterpret_cast<const f32x4 *>(eb + i * SR + q4);
                    const int64_t o = (int64_t)(a + sp + i) * r_total + r0 + q4;
                    if (ok) *reinterpret_cast<f32x4 *>(error_t + o) = f32x4{-ne[0], -ne[1], -ne[2], -ne[3]};
                    if (quant_t) {
                        f32x4 dq;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const uint32_t word = qb[(q4 + e) * 4 + (i >> 2)];
                            const float qf = (float)((word >> (8 * (i & 3))) & 0xffu);
                            const _Float16 hx = (_Float16)qf, hz = (_Float16)zv[e], hs = (_Float16)sv[e];
                            _Float16 hd = hx - hz;
                            hd = hd * hs;
                            dq[e] = (float)hd;
                        }
                        if (ok) *reinterpret_cast<f32x4 *>(quant_t + o) = dq;
                    }
                }
            }
            if (wv == 3 && j + 1 < ns) fetch_usm(j + 1);
#pragma unroll
            for (int slot = 0; slot < ST; ++slot)
#pragma unroll
                for (int k2 = 0; k2 < SB / 2; ++k2) a_cur[slot][k2] = a_nxt[slot][k2];
            QS_TR(3);
            lds_barrier();
            QS_TR(4);
        }
    }
}

#ifdef GQ_QS_TRACE
extern "C" int gq_debug_qs_trace(unsigned long long *buf)
{
    GQ_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_qs_trace), &buf, sizeof(buf)));
    return GQ_OK;
}
#endif

// q_blk [C/16][R][16] (what quant_super_kernel writes) -> qweight [R][C]
__global__ __launch_bounds__(256) void unblock_q_kernel(const uint8_t *__restrict__ src, int64_t r_total, int64_t c_total,
                                                        uint8_t *__restrict__ dst)
{
    __shared__ u32x4 tile[64][17];
    const int64_t r0 = (int64_t)blockIdx.x * 64, cb0 = (int64_t)blockIdx.y * 16, n_cb = c_total / 16;
    const int tr = threadIdx.x & 63, tq = threadIdx.x >> 6;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int cb = tq + 4 * pass;
        if (cb0 + cb < n_cb && r0 + tr < r_total)
            tile[tr][cb] = *reinterpret_cast<const u32x4 *>(src + ((cb0 + cb) * r_total + r0 + tr) * 16);
    }
    __syncthreads();
    const int ch = threadIdx.x & 15, rw = threadIdx.x >> 4;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int row = rw + 16 * pass;
        if (cb0 + ch < n_cb && r0 + row < r_total)
            *reinterpret_cast<u32x4 *>(dst + (r0 + row) * c_total + (cb0 + ch) * 16) = tile[row][ch];
    }
}

// ---- one column, one thread per output feature ------------------------------------------------
// The reference's own decomposition (gptq_cpp_kernel.cu:21-137: quantise column `col`, then the
// rank-1 update of everything behind it), for the columns of a range that the blocked kernel cannot
// take: a head or tail that is not aligned to 4 columns, or shapes whose rows of U are not 16-byte
// aligned.  Every later column receives the same fused multiply-add it receives from the blocked
// kernel and the trailing GEMM, so mixing the two forms changes no bit.
__global__ __launch_bounds__(256) void quant_col_kernel(float *__restrict__ wt, float *__restrict__ quant_t,
                                                        float *__restrict__ error_t, int16_t *__restrict__ q_out,
                                                        const float *__restrict__ u, const float *__restrict__ scale,
                                                        const float *__restrict__ zero, int64_t c_total, int64_t r_total,
                                                        int64_t col, float maxq)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= r_total) return;
    const float s = scale[r], z = zero[r];
    const float x = wt[col * r_total + r];
    const float qf = clampf(rintf(x / s) + z, 0.0f, maxq);
    const _Float16 hx = (_Float16)qf, hz = (_Float16)z, hs = (_Float16)s;
    _Float16 hd = hx - hz;
    hd = hd * hs;
    const float dq = (float)hd;
    const float *__restrict__ urow = u + col * c_total;
    const float err = (x - dq) / urow[col];
    quant_t[col * r_total + r] = dq;
    error_t[col * r_total + r] = err;
    if (q_out) q_out[col * r_total + r] = (int16_t)(uint16_t)qf;
    const float nerr = -err;
    for (int64_t c2 = col + 1; c2 < c_total; ++c2) wt[c2 * r_total + r] = fmaf(nerr, urow[c2], wt[c2 * r_total + r]);
}

// ---- grid for all groups from the unmodified weights (static groups) ------------------------
__global__ void find_params_kernel(const float *__restrict__ wt, int64_t r_total, int64_t col0, int64_t ncols,
                                   float maxq, int sym, float *__restrict__ scale, float *__restrict__ zero)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= r_total) return;
    float lo = 3.4e38f, hi = -3.4e38f, am = 0.0f;
    for (int64_t c = col0; c < col0 + ncols; ++c) {
        const float v = wt[c * r_total + r];
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
        am = fmaxf(am, fabsf(v));
    }
    float sc, zr;
    grid_from_minmax(lo, hi, am, maxq, sym, sc, zr);
    scale[r] = sc;
    zero[r] = zr;
}

// ---- W[R][C] (16-bit) -> W^T[C][R] fp32 with the act-order column gather ---------------------
template <int DTYPE>
__device__ __forceinline__ float to_f32(unsigned short raw)
{
    if constexpr (DTYPE == GQ_F16) return (float)__builtin_bit_cast(_Float16, raw);
    else return gq_bf16_to_f32(raw);
}

// tile[jj][rr] = W[r0+rr][perm[j0+jj]]  (64 x 64), 256 threads
template <int DTYPE>
__device__ __forceinline__ void load_w_tile(float (*tile)[65], const unsigned short *__restrict__ w,
                                            const int32_t *__restrict__ perm, int64_t r_total, int64_t c_total,
                                            int64_t r0, int64_t j0)
{
    const int jj = threadIdx.x & 63;
    const int64_t j = j0 + jj;
    const int64_t src_col = j < c_total ? (perm ? perm[j] : j) : -1;
#pragma unroll 4
    for (int pass = 0; pass < 16; ++pass) {
        const int rr = (threadIdx.x >> 6) + 4 * pass;
        const int64_t r = r0 + rr;
        float v = 0.0f;
        if (src_col >= 0 && r < r_total) v = to_f32<DTYPE>(w[r * c_total + src_col]);
        tile[jj][rr] = v;
    }
}

// wt[j][r] = W[r][perm[j]] as fp32.  A tile is 64 rows x 64 columns of W as they lie in memory; column s of W
// goes to row perm_inv[s] of wt, 256 contiguous bytes per tile and column.  (Picking the columns perm[j] of a
// row instead reads a 64-byte sector for every 2-byte weight unless the order is nearly monotone: 11008 x 4096
// took 0.75-1.0 ms so, against 0.11 ms.)
template <int DTYPE>
__global__ __launch_bounds__(256) void prep_kernel(const unsigned short *__restrict__ w, const int32_t *__restrict__ perm_inv,
                                                   int64_t r_total, int64_t c_total, float *__restrict__ wt)
{
    __shared__ float tile[64][65];
    const int64_t r0 = (int64_t)blockIdx.x * 64, j0 = (int64_t)blockIdx.y * 64;
    load_w_tile<DTYPE>(tile, w, nullptr, r_total, c_total, r0, j0);
    __syncthreads();
    const int rr = threadIdx.x & 63;
#pragma unroll 4
    for (int pass = 0; pass < 16; ++pass) {
        const int jj = (threadIdx.x >> 6) + 4 * pass;
        if (j0 + jj < c_total && r0 + rr < r_total) {
            const int64_t dst = perm_inv ? perm_inv[j0 + jj] : j0 + jj;
            // (an index array that is not a permutation leaves entries of its inverse at -1: those columns are dropped)
            if ((uint64_t)dst < (uint64_t)c_total) wt[dst * r_total + r0 + rr] = tile[jj][rr];
        }
    }
}

__global__ void invert_perm_kernel(const int32_t *__restrict__ perm, int64_t c, int32_t *__restrict__ perm_inv)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < c && (uint32_t)perm[j] < (uint32_t)c) perm_inv[perm[j]] = (int32_t)j;
}

// ---- metrics ----------------------------------------------------------------------------------
__device__ __forceinline__ void block_add(double v, double *dst, double *red)
{
    red[threadIdx.x] = v;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(dst, red[0]);
    __syncthreads();
}

// sums[2] += (quant_t[j][r] - W[r][perm[j]])^2 over 64x64 tiles: blockIdx.x = row tile, the column
// tiles are strided over blockIdx.y (a few hundred workgroups in all: each ends in one double
// atomic on the same address, and thousands of those serialise).  With perm_inv the tiles are columns of W
// as they lie in memory and column s is compared with row perm_inv[s] of quant_t (contiguous reads on both
// sides); without it the columns perm[j] are picked from W.
template <int DTYPE>
__global__ __launch_bounds__(256) void metrics_tile_kernel(const unsigned short *__restrict__ w,
                                                           const int32_t *__restrict__ perm,
                                                           const int32_t *__restrict__ perm_inv, int64_t r_total,
                                                           int64_t c_total, const float *__restrict__ quant_t,
                                                           double *__restrict__ sums)
{
    __shared__ float tile[64][65];
    __shared__ double red[256];
    const int64_t r0 = (int64_t)blockIdx.x * 64;
    const int rr = threadIdx.x & 63;
    double acc = 0.0;
    for (int64_t j0 = (int64_t)blockIdx.y * 64; j0 < c_total; j0 += (int64_t)gridDim.y * 64) {
        __syncthreads();
        load_w_tile<DTYPE>(tile, w, perm_inv ? nullptr : perm, r_total, c_total, r0, j0);
        __syncthreads();
        for (int pass = 0; pass < 16; ++pass) {
            const int jj = (threadIdx.x >> 6) + 4 * pass;
            if (j0 + jj < c_total && r0 + rr < r_total) {
                const int64_t row = perm_inv ? perm_inv[j0 + jj] : j0 + jj;
                if ((uint64_t)row < (uint64_t)c_total) {
                    const float d = quant_t[row * r_total + r0 + rr] - tile[jj][rr];
                    acc += (double)d * (double)d;
                }
            }
        }
    }
    block_add(acc, sums + 2, red);
}

// sums[0] += error^2 (flat), sums[1] += (W[r][j]/U[j][j])^2, sums[3] += W^2.
// Thread = one column j (its U[j][j] is loaded once) over the rows blockIdx.y, + gridDim.y, ...;
// the error array has no structure the sum cares about and is walked flat.
template <int DTYPE>
__global__ __launch_bounds__(256) void metrics_flat_kernel(const unsigned short *__restrict__ w,
                                                           const float *__restrict__ u, const float *__restrict__ error_t,
                                                           int64_t r_total, int64_t c_total, double *__restrict__ sums)
{
    __shared__ double red[256];
    double se = 0.0, sn = 0.0, sw = 0.0;
    if (error_t) {
        // 16 bytes per thread and turn, four turns in flight
        // (a pointer that is not 16-byte aligned is walked element by element in the tail loop below)
        const int64_t total4 = (((uintptr_t)error_t & 15) == 0) ? r_total * c_total / 4 : 0, n_threads = (int64_t)gridDim.x * gridDim.y * 256;
        const f32x4 *e4 = reinterpret_cast<const f32x4 *>(error_t);
        int64_t i = ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
        for (; i + 3 * n_threads < total4; i += 4 * n_threads) {
            f32x4 v[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t] = e4[i + t * n_threads];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int k = 0; k < 4; ++k) se += (double)v[t][k] * (double)v[t][k];
        }
        for (; i < total4; i += n_threads) {
            const f32x4 v = e4[i];
#pragma unroll
            for (int k = 0; k < 4; ++k) se += (double)v[k] * (double)v[k];
        }
        // (r * c is a multiple of 4 whenever the column loop ran: its row counts are; the odd tail otherwise)
        for (int64_t t = total4 * 4 + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x; t < r_total * c_total; t += n_threads)
            se += (double)error_t[t] * (double)error_t[t];
    }
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j < c_total) {
        const float ujj = u ? u[j * c_total + j] : 1.0f;
        auto take = [&](unsigned short raw) {
            const float wv = to_f32<DTYPE>(raw);
            // the reference squares in the weight's own 16-bit type (gptq_py.py:237): round each square
            float sq;
            if constexpr (DTYPE == GQ_F16) sq = (float)(_Float16)(wv * wv);
            else sq = (float)(__bf16)(wv * wv);
            sw += (double)sq;
            if (u) {
                const float q = wv / ujj;
                sn += (double)q * (double)q;
            }
        };
        // eight rows per turn with their loads in flight together (one load per turn was a memory round trip
        // per element and thread)
        const int64_t step = gridDim.y;
        int64_t r = blockIdx.y;
        for (; r + 7 * step < r_total; r += 8 * step) {
            unsigned short raw[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) raw[e] = w[(r + e * step) * c_total + j];
#pragma unroll
            for (int e = 0; e < 8; ++e) take(raw[e]);
        }
        for (; r < r_total; r += step) take(w[r * c_total + j]);
    }
    block_add(se, sums + 0, red);
    block_add(sn, sums + 1, red);
    block_add(sw, sums + 3, red);
}

// ---- result collation -----------------------------------------------------------------------
// q_t[C][R] uint8 -> qweight[R][C] uint8
__global__ __launch_bounds__(256) void transpose_u8_kernel(const uint8_t *__restrict__ src, int64_t rows, int64_t cols,
                                                           uint8_t *__restrict__ dst)
{
    __shared__ uint8_t tile[64][68];
    const int64_t i0 = (int64_t)blockIdx.y * 64, j0 = (int64_t)blockIdx.x * 64;   // src tile rows i0.., cols j0..
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int pass = 0; pass < 16; ++pass) {
        const int ii = ty + 4 * pass;
        if (i0 + ii < rows && j0 + tx < cols) tile[ii][tx] = src[(i0 + ii) * cols + j0 + tx];
    }
    __syncthreads();
    for (int pass = 0; pass < 16; ++pass) {
        const int jj = ty + 4 * pass;
        if (j0 + jj < cols && i0 + tx < rows) dst[(j0 + jj) * rows + i0 + tx] = tile[tx][jj];
    }
}

// scale_g / zero_g [G][R] fp32 -> scale [R][G] (weight dtype), qzero [R][G] uint8
template <int DTYPE>
__global__ void collate_grid_kernel(const float *__restrict__ scale_g, const float *__restrict__ zero_g, int64_t r_total,
                                    int n_groups, unsigned short *__restrict__ scale, uint8_t *__restrict__ qzero)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= r_total * n_groups) return;
    const int64_t g = i / r_total, r = i - g * r_total;
    const float s = scale_g[i];
    unsigned short raw;
    if constexpr (DTYPE == GQ_F16) raw = __builtin_bit_cast(unsigned short, (_Float16)s);
    else raw = __builtin_bit_cast(unsigned short, (__bf16)s);
    scale[r * n_groups + g] = raw;
    qzero[r * n_groups + g] = (uint8_t)zero_g[i];
}

// round-to-nearest: q and fp32 dequantised value, no error feedback (gptq_py.py:101-124)
__global__ void rtn_kernel(const float *__restrict__ wt, int64_t r_total, int64_t col0, int64_t ncols, float maxq,
                           const float *__restrict__ scale, const float *__restrict__ zero, uint8_t *__restrict__ q_t,
                           float *__restrict__ quant_t)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= r_total) return;
    const float s = scale[r], z = zero[r];
    for (int64_t c = col0; c < col0 + ncols; ++c) {
        const float x = wt[c * r_total + r];
        const float qf = clampf(rintf(x / s) + z, 0.0f, maxq);
        q_t[c * r_total + r] = (uint8_t)qf;
        if (quant_t) quant_t[c * r_total + r] = __fmul_rn(__fsub_rn(qf, z), s);
    }
}

struct LayerWorkspace {
    float *wt, *scale_g, *zero_g, *error_t;
    float *qscale_g, *sscale_g;   // EXL2: quantised scales [G][R], their step per group [G]
    uint8_t *q_t;
};

size_t carve_layer(LayerWorkspace &ws, char *base, int64_t r, int64_t c)
{
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char *p = base ? base + off : nullptr;
        off += (bytes + 255) & ~(size_t)255;
        return p;
    };
    ws.wt = (float *)take((size_t)r * c * 4);
    ws.error_t = (float *)take((size_t)r * c * 4);
    ws.q_t = (uint8_t *)take((size_t)r * c);
    ws.scale_g = (float *)take((size_t)r * c);   // up to C/4 groups of >= 4 columns, 4 bytes each
    ws.zero_g = (float *)take((size_t)r * c);
    ws.qscale_g = (float *)take((size_t)r * c);
    ws.sscale_g = (float *)take((size_t)c);
    return off;
}

int check_groups(const int32_t *sizes, const int32_t *bits, int n_groups, int64_t c, const char *who)
{
    GQ_REQUIRE(sizes && bits && n_groups > 0, "%s: group description missing", who);
    int64_t sum = 0;
    for (int g = 0; g < n_groups; ++g) {
        GQ_REQUIRE(sizes[g] > 0 && sizes[g] % 4 == 0, "%s: group %d has size %d (must be a positive multiple of 4)", who, g, sizes[g]);
        GQ_REQUIRE(bits[g] >= 1 && bits[g] <= 8, "%s: group %d has bit width %d (1..8 supported)", who, g, bits[g]);
        sum += sizes[g];
    }
    GQ_REQUIRE(sum == c, "%s: group sizes sum to %lld, in_features is %lld", who, (long long)sum, (long long)c);
    return GQ_OK;
}

// `inv_scratch`: C ints, free until the kernel has run
template <int DTYPE>
int launch_prep(const void *weight, const int32_t *perm, int32_t *inv_scratch, int64_t r, int64_t c, float *wt, hipStream_t stream)
{
    if (perm) {
        GQ_HIP(hipMemsetAsync(inv_scratch, 0xff, (size_t)c * sizeof(int32_t), stream));
        hipLaunchKernelGGL(invert_perm_kernel, dim3((unsigned)gq_cdiv(c, 256)), dim3(256), 0, stream, perm, c, inv_scratch);
        GQ_LAUNCH_CHECK();
    }
    dim3 grid((unsigned)gq_cdiv(r, 64), (unsigned)gq_cdiv(c, 64));
    hipLaunchKernelGGL(prep_kernel<DTYPE>, grid, dim3(256), 0, stream, (const unsigned short *)weight,
                       perm ? (const int32_t *)inv_scratch : nullptr, r, c, wt);
    GQ_LAUNCH_CHECK();
    return GQ_OK;
}

// largest launchable block that fits in `n` columns (n is a positive multiple of 4):
// up to 128 columns in multiples of 16, else 8 or 4
int block_size_for(int64_t n)
{
    if (n >= 16) return (int)std::min<int64_t>(QB, n - n % 16);
    return n >= 8 ? 8 : 4;
}

int launch_block(const QuantBlockArgs &p, int semantics, hipStream_t stream)
{
    const unsigned grid = (unsigned)gq_cdiv(p.r, 64);
#define GQ_QB_LAUNCH(SEM, SUB)                                                                                   \
    do {                                                                                                         \
        GQ_HIP(hipFuncSetAttribute((const void *)quant_block_kernel<SEM, SUB>,                                  \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, QB_LDS));                        \
        hipLaunchKernelGGL((quant_block_kernel<SEM, SUB>), dim3(grid), dim3(256), QB_LDS, stream, p.wt,          \
                           p.quant_t, p.error_t, p.q_out, p.u, p.scale, p.zero, p.c, p.r, p.a, p.bs, p.maxq,     \
                           p.q_is_u8, p.write_back, p.dynamic, p.sym);                                           \
    } while (0)
    if (p.bs % 16 == 0) {
        if (semantics == GQ_SEM_KERNEL) GQ_QB_LAUNCH(GQ_SEM_KERNEL, 16);
        else GQ_QB_LAUNCH(GQ_SEM_PYTHON, 16);
    } else {
        if (semantics == GQ_SEM_KERNEL) GQ_QB_LAUNCH(GQ_SEM_KERNEL, 4);
        else GQ_QB_LAUNCH(GQ_SEM_PYTHON, 4);
    }
#undef GQ_QB_LAUNCH
    GQ_LAUNCH_CHECK();
    return GQ_OK;
}

// 0: one launch per <= 128 columns (quant_block_kernel + a GEMM between two of them), 1: one launch per <= 512
// columns (quant_super_kernel) where the layer allows it.  Same bits either way; the switch is for comparisons
// and tests (gq_set_quant_super, or GQ_QUANT_SUPER=0 in the environment).
int g_quant_super = -1;

bool quant_super_enabled()
{
    if (g_quant_super < 0) {
        const char *e = getenv("GQ_QUANT_SUPER");
        g_quant_super = (e && e[0] == '0') ? 0 : 1;
    }
    return g_quant_super != 0;
}

// columns [s0, s1) of the layer (s1 - s0 <= 512, a multiple of 16), grids of all groups in scale_g / zero_g
int launch_super(const float *wt, float *quant_t, float *error_t, uint8_t *q_blk, const float *u, int64_t c, int64_t r,
                 const float *scale_g, const float *zero_g, const std::vector<int64_t> &gstart, const int32_t *gb,
                 int64_t s0, int64_t s1, hipStream_t stream)
{
    SuperTable tab{};
    size_t g = 0;
    for (int64_t col = s0; col < s1; col += SB) {
        while (gstart[g + 1] <= col) ++g;
        tab.group[(col - s0) / SB] = (uint16_t)g;
        tab.maxq[(col - s0) / SB] = (float)((1 << gb[g]) - 1);
    }
    static bool attr_set = false;
    if (!attr_set) {
        GQ_HIP(hipFuncSetAttribute((const void *)quant_super_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, QS_LDS));
        attr_set = true;
    }
    hipLaunchKernelGGL(quant_super_kernel, dim3((unsigned)gq_cdiv(r, SR)), dim3(256), QS_LDS, stream, wt, quant_t, error_t,
                       q_blk, u, c, scale_g, zero_g, r, (int)s0, (int)(s1 - s0), tab);
    GQ_LAUNCH_CHECK();
    return GQ_OK;
}

// W^T[m0:m1] -= U[k0:k1, m0:m1]^T * E^T[k0:k1]   (k ascending, fused, starting from the old weight)
int trailing_update(const float *u, float *wt, const float *error_t, int64_t c, int64_t r, int64_t k0, int64_t k1,
                    int64_t m0, int64_t m1, hipStream_t stream)
{
    if (m0 >= m1 || k0 >= k1) return GQ_OK;
    GqSgemmArgs g{};
    g.a = u + k0 * c + m0;        // A[k][m] = U[k0+k][m0+m]
    g.lda = c;
    g.b = error_t + k0 * r;       // B[k][n] = E^T[k0+k][n]
    g.ldb = r;
    g.c = wt + m0 * r;            // C[m][n] = W^T[m0+m][n]
    g.ldc = r;
    g.m = (int)(m1 - m0);
    g.n = (int)r;
    g.k = (int)(k1 - k0);
    g.layout = GQ_A_KM | GQ_B_KN;
    g.trim = GQ_TRIM_NONE;
    g.negate = 1;
    g.accumulate = 1;
    g.lower_only = 0;
    return gq_sgemm(g, stream);
}

// all groups' grids in one launch when group size and bit width are uniform
__global__ void find_params_all_kernel(const float *__restrict__ wt, int64_t r_total, int group_size, float maxq, int sym,
                                       float *__restrict__ scale_g, float *__restrict__ zero_g)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= r_total) return;
    const int64_t g = blockIdx.y, col0 = g * group_size;
    float lo = 3.4e38f, hi = -3.4e38f, am = 0.0f;
    for (int64_t c = col0; c < col0 + group_size; ++c) {
        const float v = wt[c * r_total + r];
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
        am = fmaxf(am, fabsf(v));
    }
    float sc, zr;
    grid_from_minmax(lo, hi, am, maxq, sym, sc, zr);
    scale_g[g * r_total + r] = sc;
    zero_g[g * r_total + r] = zr;
}


// ---- EXL2 scale quantisation and the shrink search ---------------------------------------------
// quantize2 / dequantize2 (quant.py:40-62): a group's scales on a quadratic grid q^2 * sscale
__device__ __forceinline__ float quantize2(float x, float s, float smaxq)
{
    return clampf(rintf(sqrtf(x / s)), 1.0f, smaxq + 1.0f);
}
__device__ __forceinline__ float dequantize2(float q, float s) { return __fmul_rn(__fmul_rn(q, q), s); }

// Up to GT groups handled by one launch (blockIdx.y = entry): their columns, bit widths, the columns
// the shrink search looks at, and where their grids live (group index into the [G][R] arrays).
constexpr int GT = 48;
struct GroupTable {
    int64_t col0[GT], m0[GT];
    int ncols[GT], mn[GT], bits[GT], group[GT];
};

__global__ void find_params_table_kernel(const float *__restrict__ wt, int64_t r_total, GroupTable tab, int sym,
                                         float *__restrict__ scale_g, float *__restrict__ zero_g)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= r_total) return;
    const int e = blockIdx.y;
    const int64_t col0 = tab.col0[e];
    float lo = 3.4e38f, hi = -3.4e38f, am = 0.0f;
    for (int64_t c = col0; c < col0 + tab.ncols[e]; ++c) {
        const float v = wt[c * r_total + r];
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
        am = fmaxf(am, fabsf(v));
    }
    float sc, zr;
    grid_from_minmax(lo, hi, am, (float)((1 << tab.bits[e]) - 1), sym, sc, zr);
    scale_g[(int64_t)tab.group[e] * r_total + r] = sc;
    zero_g[(int64_t)tab.group[e] * r_total + r] = zr;
}

// Quantizer.mse, PyTorch branch, rows (quant.py:151-155): every row tries the candidates
// scale * shrink[k] and keeps the one with the smallest sum over the search columns of
// |dequantize(quantize(x)) - x| ^ norm (first minimum).  The quantisation is the reference's fp32
// arithmetic; the error sums are accumulated in double, columns ascending.
// Block (64 rows, 16 candidate slots); slot t takes candidates t, t + 16, ...
__global__ __launch_bounds__(1024) void mse_rows_kernel(const float *__restrict__ wt, int64_t r_total, GroupTable tab,
                                                        float *__restrict__ scale_g, const float *__restrict__ zero_g,
                                                        const float *__restrict__ shrink, int n_shrink, double norm)
{
    __shared__ double best_err[16][64];
    __shared__ int best_k[16][64];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int e = blockIdx.y;
    const int64_t r = (int64_t)blockIdx.x * 64 + tx;
    const bool live = r < r_total;
    const int64_t rr = live ? r : r_total - 1;
    float *scale = scale_g + (int64_t)tab.group[e] * r_total;
    const float s0 = scale[rr], z = zero_g[(int64_t)tab.group[e] * r_total + rr];
    const float maxq = (float)((1 << tab.bits[e]) - 1);
    const int64_t m0 = tab.m0[e], m1 = m0 + tab.mn[e];
    double be = 1.0e300 * 1.0e300;   // +inf
    int bk = 0x7fffffff;
    for (int k = ty; k < n_shrink; k += 16) {
        const float sp = __fmul_rn(s0, shrink[k]);
        double err = 0.0;
        for (int64_t c = m0; c < m1; ++c) {
            const float x = wt[c * r_total + rr];
            const float q = clampf(rintf(x / sp) + z, 0.0f, maxq);
            const float dq = __fmul_rn(__fsub_rn(q, z), sp);
            err += pow((double)fabsf(__fsub_rn(dq, x)), norm);
        }
        if (err < be) {
            be = err;
            bk = k;
        }
    }
    best_err[ty][tx] = be;
    best_k[ty][tx] = bk;
    __syncthreads();
    if (ty == 0 && live) {
        for (int t = 1; t < 16; ++t) {
            const double e2 = best_err[t][tx];
            const int k = best_k[t][tx];
            if (e2 < be || (e2 == be && k < bk)) {
                be = e2;
                bk = k;
            }
        }
        if (bk != 0x7fffffff) scale[r] = __fmul_rn(s0, shrink[bk]);
    }
}

// One workgroup per table entry: the group's scales go onto the quadratic grid (quant.py:108-112),
// after an optional search for the best shrink of the grid's step (quant.py:156-165: sum over rows
// of |dequantize2(quantize2(s)) - s| ^ norm in double, first minimum).  In the search thread t walks
// candidate t % n_shrink over the rows t / n_shrink, + 256 / n_shrink, ...; the slices' sums are
// added in slice order.
__global__ __launch_bounds__(256) void exl2_scale_kernel(float *__restrict__ scale_g, float *__restrict__ qscale_g,
                                                         float *__restrict__ sscale_g, int64_t r_total, GroupTable tab,
                                                         float smaxq, int search, const float *__restrict__ shrink,
                                                         int n_shrink, double norm)
{
    __shared__ float red[256];
    __shared__ double cand_err[256];
    __shared__ float ss_shared;
    const int tid = threadIdx.x;
    const int g = tab.group[blockIdx.x];
    float *scale = scale_g + (int64_t)g * r_total, *qscale = qscale_g + (int64_t)g * r_total;
    float m = -3.4e38f;
    for (int64_t r = tid; r < r_total; r += 256) m = fmaxf(m, scale[r]);
    red[tid] = m;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (tid < w) red[tid] = fmaxf(red[tid], red[tid + w]);
        __syncthreads();
    }
    const float full = smaxq + 1.0f;
    float ss = red[0] / __fmul_rn(full, full) + 1e-12f;
    if (search) {
        const int slices = 256 / n_shrink;   // n_shrink <= 256
        const int k = tid % n_shrink, sl = tid / n_shrink;
        double err = 0.0;
        if (sl < slices) {
            const float sp = __fmul_rn(ss, shrink[k]);
            for (int64_t r = sl; r < r_total; r += slices) {
                const float x = scale[r];
                const float d = __fsub_rn(dequantize2(quantize2(x, sp, smaxq), sp), x);
                err += pow((double)fabsf(d), norm);
            }
        }
        cand_err[tid] = err;
        __syncthreads();
        if (tid == 0) {
            int bk = 0;
            double be = 1.0e300 * 1.0e300;
            for (int c = 0; c < n_shrink; ++c) {
                double e = 0.0;
                for (int q = 0; q < slices; ++q) e += cand_err[q * n_shrink + c];
                if (e < be) {
                    be = e;
                    bk = c;
                }
            }
            ss_shared = __fmul_rn(ss, shrink[bk]);
        }
        __syncthreads();
        ss = ss_shared;
    }
    for (int64_t r = tid; r < r_total; r += 256) {
        const float q = quantize2(scale[r], ss, smaxq);
        qscale[r] = q;
        scale[r] = dequantize2(q, ss);
    }
    if (tid == 0) sscale_g[g] = ss;
}

// qscale_g [G][R] fp32 -> qscale [R][G] uint8, sscale_g [G] fp32 -> sscale [G] (weight dtype)
template <int DTYPE>
__global__ void collate_exl2_kernel(const float *__restrict__ qscale_g, const float *__restrict__ sscale_g, int64_t r_total,
                                    int n_groups, uint8_t *__restrict__ qscale, unsigned short *__restrict__ sscale)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_groups) {
        const float s = sscale_g[i];
        if constexpr (DTYPE == GQ_F16) sscale[i] = __builtin_bit_cast(unsigned short, (_Float16)s);
        else sscale[i] = __builtin_bit_cast(unsigned short, (__bf16)s);
    }
    if (i >= r_total * n_groups) return;
    const int64_t g = i / r_total, r = i - g * r_total;
    qscale[r * n_groups + g] = (uint8_t)qscale_g[i];
}

}  // namespace

extern "C" int gq_find_params(const float *weights_t, int64_t c, int64_t r, int64_t col0, int64_t ncols, int bits,
                              int sym, float *scale, float *qzero, gq_stream_t stream)
{
    GQ_REQUIRE(weights_t && scale && qzero, "gq_find_params: NULL pointer");
    GQ_REQUIRE(c > 0 && r > 0 && col0 >= 0 && ncols > 0 && col0 + ncols <= c, "gq_find_params: bad column range");
    GQ_REQUIRE(bits >= 1 && bits <= 16, "gq_find_params: bad bit width %d", bits);
    const float maxq = (float)((1 << bits) - 1);
    hipLaunchKernelGGL(find_params_kernel, dim3((unsigned)gq_cdiv(r, 256)), dim3(256), 0, (hipStream_t)stream, weights_t,
                       r, col0, ncols, maxq, sym, scale, qzero);
    GQ_LAUNCH_CHECK();
    return GQ_OK;
}

extern "C" int gq_quantize_range(float *quant, const float *scale, int16_t *out_q, const float *qzero, float maxq,
                                 const float *hessian_inv, float *weights, float *error, int64_t c, int64_t r,
                                 int64_t a, int64_t b, gq_stream_t stream_)
{
    GQ_REQUIRE(quant && scale && qzero && hessian_inv && weights && error, "gq_quantize_range: NULL pointer");
    GQ_REQUIRE(c > 0 && r > 0 && 0 <= a && a <= b && b <= c, "gq_quantize_range: bad range [%lld, %lld) of %lld",
               (long long)a, (long long)b, (long long)c);
    GQ_REQUIRE(r <= (1 << 22) && c <= (1 << 22), "gq_quantize_range: at most 2^22 rows and columns");
    hipStream_t stream = (hipStream_t)stream_;
    auto single = [&](int64_t col) -> int {
        hipLaunchKernelGGL(quant_col_kernel, dim3((unsigned)gq_cdiv(r, 256)), dim3(256), 0, stream, weights, quant, error, out_q,
                           hessian_inv, scale, qzero, c, r, col, maxq);
        GQ_LAUNCH_CHECK();
        return GQ_OK;
    };
    // The blocked kernel and the trailing GEMM read U and the weights in 16-byte pieces: they take
    // the part of the range that starts and ends on a multiple of 4 columns when both dimensions
    // are multiples of 4; any other column goes through the one-column kernel (the reference takes
    // any range, gptq_cpp.cpp:51).
    const bool aligned = r % 4 == 0 && c % 4 == 0;
    const int64_t a4 = aligned ? std::min(b, gq_round_up(a, 4)) : b;
    const int64_t b4 = aligned ? std::max(a4, b - b % 4) : b;
    for (int64_t col = a; col < a4; ++col) {
        int rc = single(col);
        if (rc) return rc;
    }
    // any range is cut into register-sized blocks; the result does not depend on the cut
    for (int64_t s0 = a4, s1; s0 < b4; s0 = s1) {
        s1 = s0 + block_size_for(b4 - s0);
        QuantBlockArgs p{};
        p.wt = weights; p.quant_t = quant; p.error_t = error; p.q_out = out_q; p.u = hessian_inv;
        p.scale = const_cast<float *>(scale); p.zero = const_cast<float *>(qzero);
        p.c = c; p.r = r; p.a = (int)s0; p.bs = (int)(s1 - s0); p.maxq = maxq;
        p.q_is_u8 = 0; p.write_back = 1; p.dynamic = 0; p.sym = 0;
        int rc = launch_block(p, GQ_SEM_KERNEL, stream);
        if (rc) return rc;
        // the reference updates rows [c+1, b) in its loop and rows [b, C) afterwards; both are this chain
        if ((rc = trailing_update(hessian_inv, weights, error, c, r, s0, s1, s1, c, stream))) return rc;
    }
    for (int64_t col = b4; col < b; ++col) {
        int rc = single(col);
        if (rc) return rc;
    }
    return GQ_OK;
}

extern "C" int gq_set_quant_super(int on)
{
    const int before = quant_super_enabled() ? 1 : 0;
    g_quant_super = on ? 1 : 0;
    return before;
}

extern "C" size_t gq_quantize_layer_workspace(int64_t r, int64_t c)
{
    if (r <= 0 || c <= 0) return 0;
    LayerWorkspace ws;
    return carve_layer(ws, nullptr, r, c);
}

namespace {

// One launch of the block kernel: columns [s0, s1) of group g.
struct Launch { int64_t s0, s1; int g; };
// A run of launches between two points where a grid may be computed from the working weights:
// one launch (static grids) or one of the reference's blocks (dynamic grids).  `need_end`: the grids
// computed at the start of this unit read the working weights up to this column.
struct Unit { size_t first, last; int64_t col0, col1, need_end; int g_first, g_last; };

int quantize_layer_impl(const void *weight, int dtype, const int32_t *perm, const float *u, int64_t r, int64_t c,
                        const int32_t *gs, const int32_t *gb, int n_groups, const int32_t *bs, int n_blocks,
                        int semantics, const gq_grid_options &o, uint8_t *qweight, void *scale, uint8_t *qzero,
                        uint8_t *qscale, void *sscale, float *quant_t, float *error_t, void *workspace,
                        size_t workspace_bytes, hipStream_t stream, const char *who)
{
    GQ_REQUIRE(weight && qweight && scale && qzero && workspace, "%s: NULL pointer", who);
    GQ_REQUIRE(dtype == GQ_F16 || dtype == GQ_BF16, "%s: bad dtype %d", who, dtype);
    GQ_REQUIRE(semantics == GQ_SEM_KERNEL || semantics == GQ_SEM_PYTHON, "%s: bad semantics %d", who, semantics);
    GQ_REQUIRE(r > 0 && c > 0, "%s: bad shape", who);
    GQ_REQUIRE(!u || r % 4 == 0, "%s: out_features %lld must be a multiple of 4", who, (long long)r);
    GQ_REQUIRE(r <= (1 << 22) && c <= (1 << 22), "%s: at most 2^22 rows and columns", who);
    GQ_REQUIRE(u || !perm, "%s: a column permutation needs a Hessian factor", who);
    int rc = check_groups(gs, gb, n_groups, c, who);
    if (rc) return rc;
    const bool exl2 = o.scale_bits > 0;
    GQ_REQUIRE(o.scale_bits >= 0 && o.scale_bits <= 8, "%s: scale_bits %d (0..8 supported)", who, o.scale_bits);
    GQ_REQUIRE(!exl2 || (qscale && sscale), "%s: scale quantisation needs the qscale and sscale outputs", who);
    GQ_REQUIRE(!o.mse || (o.shrink && o.n_shrink >= 1 && o.n_shrink <= 256 && o.norm > 0.0f),
               "%s: the shrink search needs 1..256 candidates and a positive norm", who);
    const int sym = (o.sym || exl2) ? 1 : 0;
    const float smaxq = exl2 ? (float)((1 << o.scale_bits) - 1) : 0.0f;

    LayerWorkspace ws;
    const size_t need = carve_layer(ws, (char *)workspace, r, c);
    if (workspace_bytes < need) {
        gq_set_error("%s: workspace of %zu bytes, %zu needed", who, workspace_bytes, need);
        return GQ_EWORKSPACE;
    }
    float *err = error_t ? error_t : ws.error_t;
    std::vector<int64_t> gstart(n_groups + 1, 0);
    for (int g = 0; g < n_groups; ++g) gstart[g + 1] = gstart[g] + gs[g];

    // (the inverse permutation, 4 C bytes, borrows the start of the zero-point grids -- R C bytes with R >= 4 whenever
    // there is a permutation -- which are first written after this kernel, on the same stream)
    static_assert(sizeof(int32_t) == 4, "");
    GQ_REQUIRE(!perm || r >= 4, "%s: a column permutation needs at least 4 rows", who);
    rc = dtype == GQ_F16 ? launch_prep<GQ_F16>(weight, perm, (int32_t *)ws.zero_g, r, c, ws.wt, stream)
                         : launch_prep<GQ_BF16>(weight, perm, (int32_t *)ws.zero_g, r, c, ws.wt, stream);
    if (rc) return rc;

    // Grids of groups [g0, g1) from the working weights as they are now (quant.py:89-118).  With the
    // shrink search on (quant.py:120-165) each group looks at its own columns (mn < 0) or all of them
    // at columns [m0, m0 + mn).
    auto grids_for = [&](int g0, int g1, int64_t m0, int64_t mn) -> int {
        for (int base = g0; base < g1; base += GT) {
            const int n = std::min(GT, g1 - base);
            GroupTable tab{};
            for (int e = 0; e < n; ++e) {
                const int g = base + e;
                tab.col0[e] = gstart[g]; tab.ncols[e] = gs[g]; tab.bits[e] = gb[g]; tab.group[e] = g;
                tab.m0[e] = mn < 0 ? gstart[g] : m0;
                tab.mn[e] = mn < 0 ? gs[g] : (int)mn;
            }
            hipLaunchKernelGGL(find_params_table_kernel, dim3((unsigned)gq_cdiv(r, 256), (unsigned)n), dim3(256), 0, stream,
                               ws.wt, r, tab, sym, ws.scale_g, ws.zero_g);
            GQ_LAUNCH_CHECK();
            if (exl2) {
                hipLaunchKernelGGL(exl2_scale_kernel, dim3((unsigned)n), dim3(256), 0, stream, ws.scale_g, ws.qscale_g,
                                   ws.sscale_g, r, tab, smaxq, 0, (const float *)nullptr, 1, 0.0);
                GQ_LAUNCH_CHECK();
            }
            if (o.mse) {
                hipLaunchKernelGGL(mse_rows_kernel, dim3((unsigned)gq_cdiv(r, 64), (unsigned)n), dim3(64, 16), 0, stream, ws.wt,
                                   r, tab, ws.scale_g, ws.zero_g, o.shrink, o.n_shrink, (double)o.norm);
                GQ_LAUNCH_CHECK();
                if (exl2) {
                    hipLaunchKernelGGL(exl2_scale_kernel, dim3((unsigned)n), dim3(256), 0, stream, ws.scale_g, ws.qscale_g,
                                       ws.sscale_g, r, tab, smaxq, 1, o.shrink, o.n_shrink, (double)o.norm);
                    GQ_LAUNCH_CHECK();
                }
            }
        }
        return GQ_OK;
    };
    const bool plain_grid = !exl2 && !o.mse;
    bool super = false;   // the column loop ran on quant_super_kernel

    if (!u) {   // round to nearest (gptq_py.py:101-124)
        const unsigned rg = (unsigned)gq_cdiv(r, 256);
        if ((rc = grids_for(0, n_groups, 0, -1))) return rc;
        for (int g = 0; g < n_groups; ++g) {
            hipLaunchKernelGGL(rtn_kernel, dim3(rg), dim3(256), 0, stream, ws.wt, r, gstart[g], (int64_t)gs[g],
                               (float)((1 << gb[g]) - 1), ws.scale_g + (int64_t)g * r, ws.zero_g + (int64_t)g * r, ws.q_t, quant_t);
            GQ_LAUNCH_CHECK();
        }
    } else {
        // The reference's blocks: its native loop works group by group (gptq_py.py:132-144), its
        // PyTorch loop block by block (gptq_py.py:162-193).
        std::vector<int64_t> bstart;
        if (semantics == GQ_SEM_KERNEL || !bs) {
            bstart = gstart;
        } else {
            GQ_REQUIRE(n_blocks > 0, "%s: block description missing", who);
            bstart.assign(1, 0);
            for (int b = 0; b < n_blocks; ++b) {
                GQ_REQUIRE(bs[b] > 0 && bs[b] % 4 == 0, "%s: block %d has size %d (must be a positive multiple of 4)", who, b, bs[b]);
                bstart.push_back(bstart.back() + bs[b]);
            }
            GQ_REQUIRE(bstart.back() == c, "%s: block sizes sum to %lld, in_features is %lld", who,
                       (long long)bstart.back(), (long long)c);
        }
        // dynamic grids computed inside the block kernel: one launch per block == group
        bool in_kernel_grid = semantics == GQ_SEM_PYTHON && plain_grid && bstart == gstart;
        for (int g = 0; in_kernel_grid && g < n_groups; ++g) in_kernel_grid = gs[g] == block_size_for(gs[g]);

        // launches: a block cut at group boundaries, then into register-sized pieces
        std::vector<Launch> launches;
        std::vector<Unit> units;
        {
            int g = 0;
            for (size_t b = 0; b + 1 < bstart.size(); ++b) {
                const int64_t i1 = bstart[b], i2 = bstart[b + 1];
                Unit un{launches.size(), 0, i1, i2, i2, n_groups, -1};
                for (int64_t s0 = i1, s1; s0 < i2; s0 = s1) {
                    while (gstart[g + 1] <= s0) ++g;
                    if (gstart[g] == s0) {   // this group's grid is due at the start of the block (gptq_py.py:169-186)
                        un.g_first = std::min(un.g_first, g);
                        un.g_last = g;
                        un.need_end = std::max(un.need_end, gstart[g + 1]);
                    }
                    s1 = s0 + block_size_for(std::min(i2, gstart[g + 1]) - s0);
                    launches.push_back({s0, s1, g});
                }
                un.last = launches.size() - 1;
                if (semantics == GQ_SEM_KERNEL) {   // static grids: every launch may end a super-block
                    for (size_t l = un.first; l <= un.last; ++l)
                        units.push_back({l, l, launches[l].s0, launches[l].s1, launches[l].s1, n_groups, -1});
                } else {
                    units.push_back(un);
                }
            }
        }

        if (semantics == GQ_SEM_KERNEL) {   // grid of every group from the unmodified weights
            bool uniform = plain_grid;
            for (int g = 1; g < n_groups; ++g) uniform = uniform && gs[g] == gs[0] && gb[g] == gb[0];
            if (uniform) {
                dim3 grid((unsigned)gq_cdiv(r, 256), (unsigned)n_groups);
                hipLaunchKernelGGL(find_params_all_kernel, grid, dim3(256), 0, stream, ws.wt, r, (int)gs[0],
                                   (float)((1 << gb[0]) - 1), sym, ws.scale_g, ws.zero_g);
                GQ_LAUNCH_CHECK();
            } else if ((rc = grids_for(0, n_groups, 0, -1))) {
                return rc;
            }
        }

        // Native-loop semantics with every group a multiple of 16 columns: one launch per 512 columns and one
        // GEMM behind it (launch_super); anything else takes the block kernel below.
        super = semantics == GQ_SEM_KERNEL && quant_super_enabled() && c % SB == 0 && n_groups <= 65535;
        for (int g = 0; super && g < n_groups; ++g) super = gs[g] % SB == 0;
        if (super) {
            for (int64_t s0 = 0; s0 < c; s0 += SW) {
                const int64_t s1 = std::min<int64_t>(c, s0 + SW);
                if ((rc = launch_super(ws.wt, quant_t, err, ws.q_t, u, c, r, ws.scale_g, ws.zero_g, gstart, gb, s0, s1, stream))) return rc;
                if ((rc = trailing_update(u, ws.wt, err, c, r, s0, s1, s1, c, stream))) return rc;
            }
            units.clear();
        }

        // Units are gathered into super-blocks of about SUPER columns: after each launch only the
        // columns up to the super-block's eager end are updated (short GEMM, K = launch), the columns
        // behind it once per super-block (K = SUPER).  Each weight still receives its updates in
        // ascending column order.  The eager end covers every column a grid computed inside the
        // super-block reads, so those grids see all earlier updates.
        constexpr int64_t SUPER = 512;
        for (size_t i = 0; i < units.size();) {
            size_t j = i;
            const int64_t sup0 = units[i].col0;
            while (j < units.size() && units[j].col1 - sup0 < SUPER) ++j;
            const size_t last = std::min(j, units.size() - 1);
            const int64_t kend = units[last].col1;
            int64_t eager = kend;
            for (size_t k = i; k <= last; ++k) eager = std::max(eager, units[k].need_end);
            for (size_t k = i; k <= last; ++k) {
                const Unit &un = units[k];
                if (semantics == GQ_SEM_PYTHON && !in_kernel_grid && un.g_first <= un.g_last)   // the search looks at the block (gptq_py.py:178-179)
                    if ((rc = grids_for(un.g_first, un.g_last + 1, un.col0, un.col1 - un.col0))) return rc;
                for (size_t l = un.first; l <= un.last; ++l) {
                    const Launch &la = launches[l];
                    QuantBlockArgs p{};
                    p.wt = ws.wt; p.quant_t = quant_t; p.error_t = err; p.q_out = ws.q_t; p.u = u;
                    p.scale = ws.scale_g + (int64_t)la.g * r; p.zero = ws.zero_g + (int64_t)la.g * r;
                    p.c = c; p.r = r; p.a = (int)la.s0; p.bs = (int)(la.s1 - la.s0);
                    p.maxq = (float)((1 << gb[la.g]) - 1);
                    p.q_is_u8 = 1; p.write_back = 0; p.dynamic = in_kernel_grid; p.sym = sym;
                    if ((rc = launch_block(p, semantics, stream))) return rc;
                    if ((rc = trailing_update(u, ws.wt, err, c, r, la.s0, la.s1, la.s1, eager, stream))) return rc;
                }
            }
            if ((rc = trailing_update(u, ws.wt, err, c, r, sup0, kend, eager, c, stream))) return rc;
            i = last + 1;
        }
    }

    if (super) {   // the levels lie in 16-column pieces per row
        dim3 ugrid((unsigned)gq_cdiv(r, 64), (unsigned)gq_cdiv(c, 256));
        hipLaunchKernelGGL(unblock_q_kernel, ugrid, dim3(256), 0, stream, ws.q_t, r, c, qweight);
    } else {
        dim3 tgrid((unsigned)gq_cdiv(r, 64), (unsigned)gq_cdiv(c, 64));
        hipLaunchKernelGGL(transpose_u8_kernel, tgrid, dim3(256), 0, stream, ws.q_t, c, r, qweight);
    }
    GQ_LAUNCH_CHECK();
    const unsigned cg = (unsigned)gq_cdiv(r * n_groups, 256);
    if (dtype == GQ_F16) {
        hipLaunchKernelGGL(collate_grid_kernel<GQ_F16>, dim3(cg), dim3(256), 0, stream, ws.scale_g, ws.zero_g, r, n_groups, (unsigned short *)scale, qzero);
        if (exl2) hipLaunchKernelGGL(collate_exl2_kernel<GQ_F16>, dim3(cg), dim3(256), 0, stream, ws.qscale_g, ws.sscale_g, r, n_groups, qscale, (unsigned short *)sscale);
    } else {
        hipLaunchKernelGGL(collate_grid_kernel<GQ_BF16>, dim3(cg), dim3(256), 0, stream, ws.scale_g, ws.zero_g, r, n_groups, (unsigned short *)scale, qzero);
        if (exl2) hipLaunchKernelGGL(collate_exl2_kernel<GQ_BF16>, dim3(cg), dim3(256), 0, stream, ws.qscale_g, ws.sscale_g, r, n_groups, qscale, (unsigned short *)sscale);
    }
    GQ_LAUNCH_CHECK();
    return GQ_OK;
}

}  // namespace

extern "C" int gq_quantize_layer(const void *weight, int dtype, const int32_t *perm, const float *hessian_inv,
                                 int64_t r, int64_t c, const int32_t *group_sizes_host,
                                 const int32_t *group_bits_host, int n_groups, int sym, int semantics,
                                 uint8_t *qweight, void *scale, uint8_t *qzero, float *quant_t, float *error_t,
                                 void *workspace, size_t workspace_bytes, gq_stream_t stream_)
{
    GQ_REQUIRE(hessian_inv, "gq_quantize_layer: NULL pointer");
    gq_grid_options o{};
    o.sym = sym;
    return quantize_layer_impl(weight, dtype, perm, hessian_inv, r, c, group_sizes_host, group_bits_host, n_groups, nullptr,
                               0, semantics, o, qweight, scale, qzero, nullptr, nullptr, quant_t, error_t, workspace,
                               workspace_bytes, (hipStream_t)stream_, "gq_quantize_layer");
}

extern "C" int gq_quantize_rtn(const void *weight, int dtype, int64_t r, int64_t c, const int32_t *group_sizes_host,
                               const int32_t *group_bits_host, int n_groups, int sym, uint8_t *qweight, void *scale,
                               uint8_t *qzero, float *quant_t, void *workspace, size_t workspace_bytes,
                               gq_stream_t stream_)
{
    gq_grid_options o{};
    o.sym = sym;
    return quantize_layer_impl(weight, dtype, nullptr, nullptr, r, c, group_sizes_host, group_bits_host, n_groups, nullptr, 0,
                               GQ_SEM_KERNEL, o, qweight, scale, qzero, nullptr, nullptr, quant_t, nullptr, workspace,
                               workspace_bytes, (hipStream_t)stream_, "gq_quantize_rtn");
}

extern "C" int gq_quantize_layer_ex(const void *weight, int dtype, const int32_t *perm, const float *hessian_inv,
                                    int64_t r, int64_t c, const int32_t *group_sizes_host,
                                    const int32_t *group_bits_host, int n_groups, const int32_t *block_sizes_host,
                                    int n_blocks, int semantics, const gq_grid_options *options, uint8_t *qweight,
                                    void *scale, uint8_t *qzero, uint8_t *qscale, void *sscale, float *quant_t,
                                    float *error_t, void *workspace, size_t workspace_bytes, gq_stream_t stream_)
{
    GQ_REQUIRE(options, "gq_quantize_layer_ex: NULL options");
    return quantize_layer_impl(weight, dtype, perm, hessian_inv, r, c, group_sizes_host, group_bits_host, n_groups,
                               block_sizes_host, n_blocks, semantics, *options, qweight, scale, qzero, qscale, sscale, quant_t,
                               error_t, workspace, workspace_bytes, (hipStream_t)stream_, "gq_quantize_layer_ex");
}

extern "C" int gq_quant_metrics(const void *weight, int dtype, const int32_t *perm, const int32_t *perm_inv,
                                const float *hessian_inv, const float *quant_t, const float *error_t, int64_t r,
                                int64_t c, double *sums, gq_stream_t stream_)
{
    GQ_REQUIRE(weight && quant_t && sums, "gq_quant_metrics: NULL pointer");
    GQ_REQUIRE(dtype == GQ_F16 || dtype == GQ_BF16, "gq_quant_metrics: bad dtype %d", dtype);
    GQ_REQUIRE(r > 0 && c > 0, "gq_quant_metrics: bad shape");
    hipStream_t stream = (hipStream_t)stream_;
    GQ_HIP(hipMemsetAsync(sums, 0, 4 * sizeof(double), stream));
    // about 512 workgroups each
    const int64_t rt = gq_cdiv(r, 64), cb = gq_cdiv(c, 256);
    dim3 tgrid((unsigned)rt, (unsigned)std::max<int64_t>(1, std::min<int64_t>(gq_cdiv(c, 64), 512 / rt)));
    dim3 fg((unsigned)cb, (unsigned)std::max<int64_t>(1, std::min<int64_t>(r, 512 / cb)));
    const unsigned short *w = (const unsigned short *)weight;
    if (dtype == GQ_F16) {
        hipLaunchKernelGGL(metrics_tile_kernel<GQ_F16>, tgrid, dim3(256), 0, stream, w, perm, perm ? perm_inv : nullptr, r, c, quant_t, sums);
        hipLaunchKernelGGL(metrics_flat_kernel<GQ_F16>, dim3(fg), dim3(256), 0, stream, w, hessian_inv, error_t, r, c, sums);
    } else {
        hipLaunchKernelGGL(metrics_tile_kernel<GQ_BF16>, tgrid, dim3(256), 0, stream, w, perm, perm ? perm_inv : nullptr, r, c, quant_t, sums);
        hipLaunchKernelGGL(metrics_flat_kernel<GQ_BF16>, dim3(fg), dim3(256), 0, stream, w, hessian_inv, error_t, r, c, sums);
    }
    GQ_LAUNCH_CHECK();
    return GQ_OK;
}
