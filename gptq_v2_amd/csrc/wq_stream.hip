// y[M][R] = x[M][C] * W^T with W held as packed 4-bit or dense 3-bit levels + per-group scale and
// zero point: the serving form of a GPTQ-quantised Linear (fills the slot of the dense fp16
// nn.Linear the reference runs on a reconstructed weight, llama.py:172-177; the reference has no
// packed GEMM).  W[r][k] = half(q[r][k] - z[g(k)][r]) * scale[g(k)][r], rounded exactly as the
// reference's reconstruction rounds it (quant.py:30 with the operand types of quant.py:263-272).
//
// Same construction as the dense kernel of gemm_nt.hip -- 256 x 256 tile, 8 waves of 128 x 64 on
// v_mfma_f32_16x16x32, two 64 KiB LDS stages, ONE stream per wave with the fragment reads of the next
// 16 MFMAs issued between the current 16, one workgroup barrier per 64-deep k-step -- except that the
// W image of a stage is not copied in by DMA but expanded into it: every thread owns the 32 levels
// of half a row per k-step (3 or 4 words), loaded from global memory two k-steps ahead (inline asm,
// counted by hand) and expanded one word per 16 MFMAs, the VALU work riding in the shadow of the
// partner wave's matrix instructions.
//
// Timeline of step t (barrier(t) sits behind the third quarter of step t's MFMAs; stage = t & 1):
//   reads of step t          from stage t & 1        between barrier(t-1) and barrier(t)
//   x DMA of step t+1        into stage (t+1) & 1    right behind barrier(t-1), waited at barrier(t)
//   W words of step t+2      into register set t & 1 right behind barrier(t-1), waited at barrier(t)
//   expansion of step t+1    into stage (t+1) & 1    word 0 behind barrier(t-1), words 1..3 in the three
//                                                    quarters before barrier(t)  (set (t+1) & 1)
#include <algorithm>
#include <type_traits>

#include <cstdlib>

#include "gq_common.h"

namespace {

constexpr int ST = 256;                        // tile edge
constexpr int SK = 64;                         // k-step
constexpr int S_OP_BYTES = ST * SK * 2;        // 32 KiB per operand image
constexpr int S_STAGE_BYTES = 2 * S_OP_BYTES;  // 64 KiB
constexpr int S_LDS = 2 * S_STAGE_BYTES;       // 128 KiB: [x0][x1][w0][w1]

typedef float f32x4s __attribute__((ext_vector_type(4)));
typedef float f32x2s __attribute__((ext_vector_type(2)));

template <int OFF>
__device__ __forceinline__ void sq_read_b128(u32x4 &dst, unsigned lds_addr)
{
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(lds_addr), "i"(OFF));
}

typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));

template <int BITS>
struct SqRegsT {
    typename std::conditional<BITS == 4, u32x4, u32x3>::type q;   // the thread's packed words of one k-step
    unsigned s_raw;   // scale (16-bit pattern) as loaded
    unsigned z_raw;   // zero point as loaded
};
#define SqRegs SqRegsT<BITS>

// (x & mask) | magic in one instruction; the mask rides in an SGPR
__device__ __forceinline__ unsigned sq_and_or(unsigned x, unsigned mask, unsigned magic)
{
    unsigned r;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "s"(mask), "v"(magic));
    return r;
}

// ---- fp16: exponent tricks.  A level q sitting at bit offset o of a 16-bit half becomes the fp16
// number B + q by OR-ing the exponent pattern of B = 2^(10 - o) (its mantissa unit at bit o is 1);
// subtracting B + z is exact, the product with the scale is rounded once.
struct SqConstH {
    f16x2 s2;
    f16x2 zb[3];   // B + z for the offsets in use (4-bit: 1024, 64; 3-bit: 1024, 128, 16)
};

template <int BITS>
__device__ __forceinline__ SqConstH sq_const_h(const SqRegsT<BITS> &w)
{
    SqConstH d;
    const _Float16 s = __builtin_bit_cast(_Float16, (unsigned short)w.s_raw);
    const float z = (float)w.z_raw;
    d.s2 = f16x2{s, s};
    const _Float16 b0 = (_Float16)(1024.0f + z), b1 = (_Float16)((BITS == 4 ? 64.0f : 128.0f) + z), b2 = (_Float16)(16.0f + z);
    d.zb[0] = f16x2{b0, b0};
    d.zb[1] = f16x2{b1, b1};
    d.zb[2] = f16x2{b2, b2};
    return d;
}

// pair t (0..3) of chunk wi (0..3): levels 8 wi + 2 t and 8 wi + 2 t + 1 of the thread's 32, as one fp16 pair
template <int BITS>
__device__ __forceinline__ unsigned sq_expand_h(const SqRegsT<BITS> &w, const SqConstH &d, int wi, int t, unsigned m0, unsigned m1, unsigned m2)
{
    f16x2 v;
    if constexpr (BITS == 4) {
        // word wi holds its 8 levels pair-interleaved: (word >> 4p) & 0x000F000F = levels (2p, 2p+1)
        const unsigned src = t < 2 ? w.q[wi] : (w.q[wi] >> 8);
        if ((t & 1) == 0) v = __builtin_bit_cast(f16x2, sq_and_or(src, 0x000F000Fu, m0)) - d.zb[0];
        else v = __builtin_bit_cast(f16x2, sq_and_or(src, 0x00F000F0u, m1)) - d.zb[1];
    } else {
        // 3 words hold 16 pairs: pair P < 15 in word P / 5 at bit offset 3 (P % 5) of both halves, pair 15
        // in bits 15 and 31 of the three words.  Offsets 0, 3, 6 are masked in place, 9 and 12 after one shift.
        const int P = 4 * wi + t;
        if (P < 15) {
            const int j = P / 5, p = P % 5;
            const unsigned src = p < 3 ? w.q[j] : (w.q[j] >> 9);
            const int slot = p < 3 ? p : p - 3;
            if (slot == 0) v = __builtin_bit_cast(f16x2, sq_and_or(src, 0x00070007u, m0)) - d.zb[0];
            else if (slot == 1) v = __builtin_bit_cast(f16x2, sq_and_or(src, 0x00380038u, m1)) - d.zb[1];
            else v = __builtin_bit_cast(f16x2, sq_and_or(src, 0x01C001C0u, m2)) - d.zb[2];
        } else {
            const unsigned bits = ((w.q[0] >> 15) & 0x00010001u) | ((w.q[1] >> 14) & 0x00020002u) | ((w.q[2] >> 13) & 0x00040004u);
            v = __builtin_bit_cast(f16x2, bits | m0) - d.zb[0];
        }
    }
    return __builtin_bit_cast(unsigned, v * d.s2);
}

// ---- bf16: the same trick with bf16's 7 mantissa bits -- a level in the low bits of a half OR-ed with the
// exponent of 128 is the bf16 number 128 + q.  Widened to fp32 (a shift), one fma per value:
// (128 + q) * s - (128 + z) * s, every term and the sum exact in fp32, one rounding to bf16.
struct SqConstB {
    f32x2s s2;
    f32x2s nzs;   // -(B + z) * s, B = 128
};

template <int BITS>
__device__ __forceinline__ SqConstB sq_const_b(const SqRegsT<BITS> &w)
{
    SqConstB d;
    const float s = __uint_as_float(w.s_raw << 16), z = (float)w.z_raw;
    const float n0 = -(128.0f + z) * s;   // exact
    d.s2 = f32x2s{s, s};
    d.nzs = f32x2s{n0, n0};
    return d;
}

// src holds a level pair in the low bits of both halves; mask selects it, magic is the pattern of 128 in both halves
__device__ __forceinline__ unsigned sq_pair_b(unsigned src, unsigned mask, unsigned magic, const SqConstB &d)
{
    const unsigned both = sq_and_or(src, mask, magic);
    const f32x2s v = f32x2s{__uint_as_float(both << 16), __uint_as_float(both & 0xFFFF0000u)};
    const f32x2s r = __builtin_elementwise_fma(v, d.s2, d.nzs);
    typedef __bf16 bf16x2s __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2s));   // one v_cvt_pk_bf16_f32
}

template <int BITS>
__device__ __forceinline__ unsigned sq_expand_b(const SqRegsT<BITS> &w, const SqConstB &d, int wi, int t, unsigned m0)
{
    if constexpr (BITS == 4) {
        return sq_pair_b(w.q[wi] >> (4 * t), 0x000F000Fu, m0, d);
    } else {
        const int P = 4 * wi + t;
        if (P < 15) return sq_pair_b(w.q[P / 5] >> (3 * (P % 5)), 0x00070007u, m0, d);
        const unsigned bits = ((w.q[0] >> 15) & 0x00010001u) | ((w.q[1] >> 14) & 0x00020002u) | ((w.q[2] >> 13) & 0x00040004u);
        return sq_pair_b(bits, 0x00070007u, m0, d);
    }
}

template <int DT, int BITS>
__global__ __launch_bounds__(512) void wq_stream_kernel(unsigned short *__restrict__ y, const unsigned short *__restrict__ x,
                                                        const uint32_t *__restrict__ qw, const unsigned short *__restrict__ scale,
                                                        const uint8_t *__restrict__ zero, int64_t m_total, int64_t r_total,
                                                        int64_t c_total, int group_size, int tiles_n, int n_blocks, int band)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int WORDS = BITS == 4 ? 4 : 3;   // per thread and k-step
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
    const int64_t m0 = (int64_t)tm * ST, n0 = (int64_t)tn * ST;
    const int n_ktiles = (int)(c_total / SK);

    // LDS: x images of the two stages, then the W images: [x0][x1][w0][w1], 32 KiB each.  Every read
    // address is one of four lane registers (x / w, chunk parity) plus an immediate offset.
    const int wr = wave >> 2, wc = wave & 3;
    const int l15 = lane & 15, lg = lane >> 4;
    const unsigned lds_base = (unsigned)(uintptr_t)GQ_LDS_PTR(char, smem);
    const int sw = (l15 >> 1) & 7;
    unsigned xaddr[2], waddr[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const unsigned choff = (unsigned)(((4 * c + lg) ^ sw) << 4);
        xaddr[c] = lds_base + (wr * 128 + l15) * 128 + choff;
        waddr[c] = lds_base + 2 * S_OP_BYTES + (wc * 64 + l15) * 128 + choff;
    }

    f32x4s acc[4][8];   // [w block][x block]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = f32x4s{0.f, 0.f, 0.f, 0.f};

    // x pieces by LDS-DMA: a wave-uniform base plus one byte offset per piece and lane (row clamped to the
    // last row of x, swizzled chunk)
    const unsigned short *xt = x + m0 * c_total;
    unsigned voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + (lane >> 3);
        const int64_t lim = m_total - m0 - 1;
        const int rc = (int)(row < lim ? row : lim);
        voff[i] = (unsigned)rc * (unsigned)(c_total * 2) + (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) << 4);
    }
    auto dma = [&](int kt, int stage, int i) __attribute__((always_inline)) {
        const char *base = reinterpret_cast<const char *>(xt + (int64_t)kt * SK);
        __builtin_amdgcn_global_load_lds(GQ_GLOBAL_PTR(base + voff[i]), GQ_LDS_PTR(void, smem + stage * S_OP_BYTES + (wave * 4 + i) * 1024), 16, 0,
                                         0);
    };

    // packed W: thread = (row tid >> 1 of the tile, clamped to the last row of W; half tid & 1 of the k-step).
    // The read position advances by one k-step per load; the group of the thread's 32 levels is tracked by
    // `rem` (their offset inside the group; groups are multiples of 32, so they never straddle two).
    const int wrow_t = tid >> 1, whalf = tid & 1;
    const int64_t wr_lim = r_total - n0 - 1;
    const unsigned wrc = (unsigned)(wrow_t < wr_lim ? wrow_t : wr_lim);
    const char *qbase = reinterpret_cast<const char *>(qw + n0 * (2 * WORDS));   // advances by one k-step per load
    const unsigned qoff = wrc * (8 * WORDS) + whalf * (4 * WORDS);
    const int64_t kstep_bytes = r_total * (8 * WORDS);
    unsigned goff = wrc;                       // element offset into scale / zero: group * r_total + row
    int rem = whalf * 32;
    while (rem >= group_size) {
        rem -= group_size;
        goff += (unsigned)r_total;
    }
    const unsigned short *sbase = scale + n0;
    const uint8_t *zbase = zero + n0;
    auto load_w = [&](SqRegs &w) __attribute__((always_inline)) {
        // asm: hipcc must neither count these loads nor touch their results before the hand-placed wait;
        // uniform base in SGPRs, 32-bit lane offset
        const unsigned goff2 = goff * 2;
        if constexpr (BITS == 4) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(w.q) : "v"(qoff), "s"(qbase) : "memory");
        else asm volatile("global_load_dwordx3 %0, %1, %2" : "=v"(w.q) : "v"(qoff), "s"(qbase) : "memory");
        asm volatile("global_load_ushort %0, %1, %2" : "=v"(w.s_raw) : "v"(goff2), "s"(sbase) : "memory");
        asm volatile("global_load_ubyte %0, %1, %2" : "=v"(w.z_raw) : "v"(goff), "s"(zbase) : "memory");
        qbase += kstep_bytes;
        rem += SK;
#pragma unroll
        for (int i = 0; i < 2; ++i) {   // SK = 64 <= 2 * group_size
            const bool over = rem >= group_size;
            rem = over ? rem - group_size : rem;
            goff = over ? goff + (unsigned)r_total : goff;
        }
    };
    auto loads_landed = [&](SqRegs &w) __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(w.q), "+v"(w.s_raw), "+v"(w.z_raw)::"memory");
    };
    // exponent patterns of the offsets in use: fp16 1024 / 64 (4-bit) or 1024 / 128 / 16 (3-bit); bf16 128
    unsigned mg0 = DT == GQ_F16 ? 0x64006400u : 0x43004300u, mg1 = BITS == 4 ? 0x54005400u : 0x58005800u, mg2 = 0x4C004C00u;
    asm volatile("" : "+v"(mg0), "+v"(mg1), "+v"(mg2));   // keep them in VGPRs (operands of v_and_or_b32)
    // word wi of set `w` into the W image of `stage`.  The scale / zero constants are derived with word 0 and
    // kept for words 1..3 (one set of them is live at a time: word 0 of the next step follows word 3 of this
    // one).  The store is asm so that hipcc does not order it against the LDS-DMA in flight.
    typename std::conditional<DT == GQ_F16, SqConstH, SqConstB>::type cst;
    const unsigned wdst = (lds_base + 2 * S_OP_BYTES + wrow_t * 128) | (unsigned)(((whalf * 4) ^ ((wrow_t >> 1) & 7)) << 4);
    u32x4 wout;   // the chunk under construction
    auto expand_const = [&](const SqRegs &w) __attribute__((always_inline)) {
        if constexpr (DT == GQ_F16) cst = sq_const_h<BITS>(w);
        else cst = sq_const_b<BITS>(w);
    };
    auto expand_pair = [&](const SqRegs &w, int wi, int t) __attribute__((always_inline)) {
        if constexpr (DT == GQ_F16) wout[t] = sq_expand_h<BITS>(w, cst, wi, t, mg0, mg1, mg2);
        else wout[t] = sq_expand_b<BITS>(w, cst, wi, t, mg0);
    };
    auto expand_store = [&](auto stage_c, int wi) __attribute__((always_inline)) {
        const unsigned addr = wdst ^ (unsigned)(wi << 4);
        const u32x4 o = wout;
        asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(o), "i"(decltype(stage_c)::value * S_OP_BYTES) : "memory");
    };
    auto expand = [&](const SqRegs &w, auto stage_c, int wi) __attribute__((always_inline)) {   // a whole chunk (prologue)
        if (wi == 0) expand_const(w);
#pragma unroll
        for (int t = 0; t < 4; ++t) expand_pair(w, wi, t);
        expand_store(stage_c, wi);
    };
    // the chunk's work spread over a quarter (16 matrix instructions): one pair behind instructions 1, 5, 9 and
    // 13, the store behind 15, so that no gap between two matrix instructions holds more than a few VALU
    // instructions (both waves of a SIMD run this schedule in step; a long gap idles the matrix pipe)
    auto expand_spread = [&](const SqRegs &w, auto stage_c, int wi, int i) __attribute__((always_inline)) {
        if ((i & 3) == 1) expand_pair(w, wi, i >> 2);
        if (i == 15) expand_store(stage_c, wi);
    };

    u32x4 fw[2][4], fx[2][4];   // fw[chunk parity][w block], fx[half][x block of the half]
    auto read_w = [&](int set, auto stage_c, int c, auto u_c) __attribute__((always_inline)) {
        sq_read_b128<decltype(stage_c)::value * S_OP_BYTES + decltype(u_c)::value * 2048>(fw[set][decltype(u_c)::value], waddr[c]);
    };
    auto read_x = [&](auto half_c, auto stage_c, int c, auto u_c) __attribute__((always_inline)) {
        constexpr int HALF = decltype(half_c)::value;
        sq_read_b128<decltype(stage_c)::value * S_OP_BYTES + HALF * 8192 + decltype(u_c)::value * 2048>(fx[HALF][decltype(u_c)::value], xaddr[c]);
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
    auto lds_done = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;
    // the hooks name their read by a run-time index that is constant after unrolling: dispatch to the immediate
    auto for_u = [&](int u, auto f) __attribute__((always_inline)) {
        switch (u) {
        case 0: f(std::integral_constant<int, 0>{}); break;
        case 1: f(std::integral_constant<int, 1>{}); break;
        case 2: f(std::integral_constant<int, 2>{}); break;
        default: f(std::integral_constant<int, 3>{}); break;
        }
    };

    // prologue: steps 0 and 1 whole, the words of step 2 in flight
    SqRegs wset0, wset1;
    load_w(wset0);
#pragma unroll
    for (int i = 0; i < 4; ++i) dma(0, 0, i);
    if (n_ktiles > 1) {
        load_w(wset1);
#pragma unroll
        for (int i = 0; i < 4; ++i) dma(1, 1, i);
    }
    loads_landed(wset0);
    if (n_ktiles > 1) loads_landed(wset1);
#pragma unroll
    for (int wi = 0; wi < 4; ++wi) expand(wset0, C0{}, wi);
    if (n_ktiles > 1) {
#pragma unroll
        for (int wi = 0; wi < 4; ++wi) expand(wset1, C1{}, wi);
    }
    if (n_ktiles > 2) load_w(wset0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 4; ++u) for_u(u, [&](auto uc) __attribute__((always_inline)) { read_w(0, C0{}, 0, uc); });
#pragma unroll
    for (int u = 0; u < 4; ++u) for_u(u, [&](auto uc) __attribute__((always_inline)) { read_x(C0{}, C0{}, 0, uc); });
    lds_done();

    // One k-step kt of parity PAR (stage PAR, the other stage NXT).  w_odd is register set (kt+1) & 1: the
    // words of step kt+1, expanded here (words 1..3) unless kt == 0, then free behind barrier(kt) for the
    // words of step kt+3.  w_even is set kt & 1: the words of step kt+2, landed at barrier(kt); their word 0
    // is expanded in the last quarter.  LOAD: 0 = no step kt+3, 1 = there is one, 2 = ask.
    auto body = [&](auto par_c, auto more_c, auto more2_c, auto first_c, auto load_c, int kt, SqRegs &w_odd, SqRegs &w_even)
                    __attribute__((always_inline)) {
        using PAR = std::integral_constant<int, decltype(par_c)::value>;
        using NXT = std::integral_constant<int, 1 - decltype(par_c)::value>;
        constexpr bool MORE = decltype(more_c)::value, MORE2 = decltype(more2_c)::value;
        constexpr bool FIRST = decltype(first_c)::value;
        constexpr int LOAD = decltype(load_c)::value;
        half_step(0, 0, [&](int i) __attribute__((always_inline)) {
            if ((i & 1) == 0 && i < 8) for_u(i >> 1, [&](auto uc) __attribute__((always_inline)) { read_x(C1{}, PAR{}, 0, uc); });
            if constexpr (MORE && !FIRST) expand_spread(w_odd, NXT{}, 1, i);
        });
        lds_done();
        half_step(0, 1, [&](int i) __attribute__((always_inline)) {
            if ((i & 1) == 0) {
                if (i < 8) for_u(i >> 1, [&](auto uc) __attribute__((always_inline)) { read_w(1, PAR{}, 1, uc); });
                else for_u((i >> 1) - 4, [&](auto uc) __attribute__((always_inline)) { read_x(C0{}, PAR{}, 1, uc); });
            }
            if constexpr (MORE && !FIRST) expand_spread(w_odd, NXT{}, 2, i);
        });
        lds_done();
        half_step(1, 0, [&](int i) __attribute__((always_inline)) {
            if ((i & 1) == 0 && i < 8) for_u(i >> 1, [&](auto uc) __attribute__((always_inline)) { read_x(C1{}, PAR{}, 1, uc); });
            if constexpr (MORE && !FIRST) expand_spread(w_odd, NXT{}, 3, i);
        });
        lds_done();
        if constexpr (MORE) {
            // x pieces of step kt+1 and the words of step kt+2 (set kt & 1)
            loads_landed(w_even);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        half_step(1, 1, [&](int i) __attribute__((always_inline)) {
            if constexpr (MORE2) {
                if constexpr (LOAD == 1) {
                    if (i == 0) load_w(w_odd);
                } else if constexpr (LOAD == 2) {
                    if (i == 0 && kt + 3 < n_ktiles) load_w(w_odd);
                }
                if ((i & 3) == 3) dma(kt + 2, PAR::value, i >> 2);
            }
            if constexpr (MORE) {
                if ((i & 1) == 0) {
                    if (i < 8) for_u(i >> 1, [&](auto uc) __attribute__((always_inline)) { read_w(0, NXT{}, 0, uc); });
                    else for_u((i >> 1) - 4, [&](auto uc) __attribute__((always_inline)) { read_x(C0{}, NXT{}, 0, uc); });
                }
            }
            if constexpr (MORE2) {
                if (i == 0) expand_const(w_even);
                expand_spread(w_even, PAR{}, 0, i);
            }
        });
        if constexpr (MORE) lds_done();
    };
    using T = std::true_type;
    using F = std::false_type;
    using L0 = std::integral_constant<int, 0>;
    using L1 = std::integral_constant<int, 1>;
    using L2 = std::integral_constant<int, 2>;
    // even kt: w_odd = wset1, w_even = wset0; odd kt: the other way round
    if (n_ktiles > 2) {
        body(C0{}, T{}, T{}, T{}, L2{}, 0, wset1, wset0);
        int kt = 1;
        for (; kt + 4 < n_ktiles; kt += 2) {   // both steps have a step kt+3 to load for
            body(C1{}, T{}, T{}, F{}, L1{}, kt, wset0, wset1);
            body(C0{}, T{}, T{}, F{}, L1{}, kt + 1, wset1, wset0);
        }
        // kt is odd and 2, 3 or 4 steps are left
        const int left = n_ktiles - kt;
        if (left == 4) {
            body(C1{}, T{}, T{}, F{}, L1{}, kt, wset0, wset1);
            body(C0{}, T{}, T{}, F{}, L0{}, kt + 1, wset1, wset0);
            body(C1{}, T{}, F{}, F{}, L0{}, kt + 2, wset0, wset1);
            body(C0{}, F{}, F{}, F{}, L0{}, kt + 3, wset1, wset0);
        } else if (left == 3) {
            body(C1{}, T{}, T{}, F{}, L0{}, kt, wset0, wset1);
            body(C0{}, T{}, F{}, F{}, L0{}, kt + 1, wset1, wset0);
            body(C1{}, F{}, F{}, F{}, L0{}, kt + 2, wset0, wset1);
        } else {
            body(C1{}, T{}, F{}, F{}, L0{}, kt, wset0, wset1);
            body(C0{}, F{}, F{}, F{}, L0{}, kt + 1, wset1, wset0);
        }
    } else if (n_ktiles == 2) {
        body(C0{}, T{}, F{}, T{}, L0{}, 0, wset1, wset0);
        body(C1{}, F{}, F{}, F{}, L0{}, 1, wset0, wset1);
    } else {
        body(C0{}, F{}, F{}, T{}, L0{}, 0, wset1, wset0);
    }

    // lane: token l15 of x block b, output features 4 lg .. 4 lg + 3 of w block a
    {
        unsigned short *yw = y + (m0 + wr * 128) * r_total + n0 + wc * 64;
        const int rows_left = (int)min((int64_t)128, m_total - (m0 + wr * 128));
        const int cols_left = (int)min((int64_t)64, r_total - (n0 + wc * 64));
        const bool wide = (r_total & 3) == 0;
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
                unsigned short *dst = yw + (unsigned)row * (unsigned)r_total + (unsigned)col;
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
}

#undef SqRegs

}  // namespace

// Launches the stream kernel; bits 4: GQ_PACK_W4_KBLOCK operand, bits 3: GQ_PACK_W3_KBLOCK operand.
int gq_wq_stream_launch(void *y, const void *x, const uint32_t *qweight_packed, const void *scale, const uint8_t *qzero,
                        int64_t m, int64_t r, int64_t c, int group_size, int bits, int dtype, hipStream_t stream)
{
    const int tiles_m = (int)gq_cdiv(m, ST), tiles_n = (int)gq_cdiv(r, ST);
    const int n_blocks = tiles_m * tiles_n;
    const int band = 2;
    void (*kern)(unsigned short *, const unsigned short *, const uint32_t *, const unsigned short *, const uint8_t *, int64_t,
                 int64_t, int64_t, int, int, int, int) = nullptr;
    if (bits == 4) kern = dtype == GQ_F16 ? wq_stream_kernel<GQ_F16, 4> : wq_stream_kernel<GQ_BF16, 4>;
    else kern = dtype == GQ_F16 ? wq_stream_kernel<GQ_F16, 3> : wq_stream_kernel<GQ_BF16, 3>;
    GQ_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, S_LDS));
    hipLaunchKernelGGL(kern, dim3(n_blocks), dim3(512), S_LDS, stream, (unsigned short *)y, (const unsigned short *)x,
                       qweight_packed, (const unsigned short *)scale, qzero, m, r, c, group_size, tiles_n, n_blocks, band);
    GQ_LAUNCH_CHECK();
    return GQ_OK;
}
