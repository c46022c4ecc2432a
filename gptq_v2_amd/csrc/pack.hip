// Bit packing of quantised rows, dequantisation back to a dense 16-bit weight (the reference's
// reconstruct_nn_linear, quant.py:222-237, 262-279) and the activation column gather for
// act-order.  All bandwidth-bound, one pass over the data, coalesced on the wide side.
#include "gq_common.h"

namespace {

// nibble position of value v (0..7) inside a word in the GQ_PACK_W4_PAIRS layout
__host__ __device__ __forceinline__ int w4_pairs_nibble(int v) { return (v & 1) * 4 + (v >> 1); }

// GQ_PACK_W3_KBLOCK: 32 consecutive levels of a row in 3 words.  Level v < 30 sits in word v / 10 at bit
// offset 3 * ((v % 10) / 2) of the low (v even) or high (v odd) 16-bit half, so (word >> 3p) & 0x00070007
// yields the pair (10 j + 2 p, 10 j + 2 p + 1); bit j of level 30 is bit 15 of word j, bit j of level 31
// is bit 31 of word j.
__host__ __device__ __forceinline__ uint32_t w3_word(const uint8_t *v32, int j, int64_t n_left)
{
    uint32_t word = 0;
    for (int p = 0; p < 5; ++p) {
        const int a = 10 * j + 2 * p, b = a + 1;
        const uint32_t va = a < n_left ? (v32[a] & 7u) : 0u, vb = b < n_left ? (v32[b] & 7u) : 0u;
        word |= va << (3 * p);
        word |= vb << (16 + 3 * p);
    }
    const uint32_t v30 = 30 < n_left ? (v32[30] & 7u) : 0u, v31 = 31 < n_left ? (v32[31] & 7u) : 0u;
    word |= ((v30 >> j) & 1u) << 15;
    word |= ((v31 >> j) & 1u) << 31;
    return word;
}

__host__ __device__ __forceinline__ uint32_t w3_level(const uint32_t *w3, int v)
{
    if (v < 30) {
        const int j = v / 10, p = (v % 10) / 2, h = v & 1;
        return (w3[j] >> (3 * p + 16 * h)) & 7u;
    }
    const int sh = v == 30 ? 15 : 31;
    return ((w3[0] >> sh) & 1u) | (((w3[1] >> sh) & 1u) << 1) | (((w3[2] >> sh) & 1u) << 2);
}

// one thread per output word of the 3-bit k-block layout: packed[n/64][rows][6]
__global__ void pack_w3_kernel(const uint8_t *__restrict__ values, int64_t rows, int64_t n, uint32_t *__restrict__ packed)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * (n / 64) * 6) return;
    const int64_t kb = i / (rows * 6), rem = i - kb * rows * 6;
    const int64_t row = rem / 6;
    const int w = (int)(rem - row * 6);           // word of the k-step: chunk w / 3, word w % 3
    const uint8_t *src = values + row * n + kb * 64 + (w / 3) * 32;
    packed[i] = w3_word(src, w % 3, 32);
}

__global__ void unpack_w3_kernel(const uint32_t *__restrict__ packed, int64_t rows, int64_t n, uint8_t *__restrict__ values)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * n) return;
    const int64_t row = i / n, idx = i - row * n;
    const uint32_t *w3 = packed + ((idx >> 6) * rows + row) * 6 + ((idx & 63) >> 5) * 3;
    values[i] = (uint8_t)w3_level(w3, (int)(idx & 31));
}

// one thread per output word
__global__ void pack_kernel(const uint8_t *__restrict__ values, int64_t rows, int64_t n, int bits, int layout,
                            int64_t words_per_row, uint32_t *__restrict__ packed)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * words_per_row) return;
    int64_t row = i / words_per_row, wi = i - row * words_per_row;
    if (layout == GQ_PACK_W4_KBLOCK) {   // output word i = ((kb * rows) + row) * 8 + w
        const int64_t kb = i / (rows * 8), rem = i - kb * rows * 8;
        row = rem >> 3;
        wi = kb * 8 + (rem & 7);
    }
    const uint8_t *src = values + row * n;
    uint32_t word = 0;
    if (layout != GQ_PACK_STREAM) {
#pragma unroll
        for (int v = 0; v < 8; ++v) {
            const int64_t idx = wi * 8 + v;
            const uint32_t val = idx < n ? (src[idx] & 15u) : 0u;
            word |= val << (4 * w4_pairs_nibble(v));
        }
    } else {
        const int64_t bit0 = wi * 32;
        int64_t first = bit0 / bits;                  // first value with bits in this word
        for (int64_t idx = first; idx < n && idx * bits < bit0 + 32; ++idx) {
            const uint64_t val = src[idx] & ((1u << bits) - 1u);
            const int64_t sh = idx * bits - bit0;     // may be negative: value started in the previous word
            word |= sh >= 0 ? (uint32_t)(val << sh) : (uint32_t)(val >> (-sh));
        }
    }
    packed[i] = word;
}

// one thread per value
__global__ void unpack_kernel(const uint32_t *__restrict__ packed, int64_t rows, int64_t n, int bits, int layout,
                              int64_t words_per_row, uint8_t *__restrict__ values)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * n) return;
    const int64_t row = i / n, idx = i - row * n;
    const uint32_t *src = packed + row * words_per_row;
    uint32_t val;
    if (layout == GQ_PACK_W4_PAIRS) {
        val = (src[idx >> 3] >> (4 * w4_pairs_nibble((int)(idx & 7)))) & 15u;
    } else if (layout == GQ_PACK_W4_KBLOCK) {
        const uint32_t word = packed[((idx >> 6) * rows + row) * 8 + ((idx & 63) >> 3)];
        val = (word >> (4 * w4_pairs_nibble((int)(idx & 7)))) & 15u;
    } else {
        const int64_t bit = idx * bits, wi = bit >> 5;
        const int sh = (int)(bit & 31);
        uint64_t two = src[wi];
        if (sh + bits > 32) two |= (uint64_t)src[wi + 1] << 32;
        val = (uint32_t)(two >> sh) & ((1u << bits) - 1u);
    }
    values[i] = (uint8_t)val;
}

template <int DTYPE>
__device__ __forceinline__ unsigned short deq16(int q, int z, unsigned short s_raw)
{
    // (int16 q - int16 z) * scale evaluated in the 16-bit type: the integer difference is exact in
    // either type, the product of two such values is exact in fp32, so one rounding = the
    // correctly rounded 16-bit product.
    const float d = (float)(q - z);
    if constexpr (DTYPE == GQ_F16) {
        const float s = (float)__builtin_bit_cast(_Float16, s_raw);
        return __builtin_bit_cast(unsigned short, (_Float16)(d * s));
    } else {
        const float s = gq_bf16_to_f32(s_raw);
        return __builtin_bit_cast(unsigned short, (__bf16)(d * s));
    }
}

// One workgroup per DQ_ROWS rows: their levels (C bytes each), scales and zero points are staged in LDS, then
// out[r][j] = deq(q[r][perm_inv[j]]), eight columns per thread and pass with one 16-byte store per row.  The
// permutation entry and the group of a column are looked up once for all rows of the workgroup (they were a
// 4-byte load and a 64-bit division per ELEMENT: more traffic than the weights themselves).
constexpr int DQ_ROWS = 4;

template <int DTYPE>
__global__ __launch_bounds__(256) void dequant_kernel(const uint8_t *__restrict__ q, const unsigned short *__restrict__ scale,
                                                      const uint8_t *__restrict__ zero, const int32_t *__restrict__ perm_inv,
                                                      int64_t r_total, int64_t c, int n_groups, int group_size,
                                                      const int32_t *__restrict__ col_group, unsigned short *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ci = (int)c, cpad = (ci + 15) & ~15;
    uint8_t *qrow = (uint8_t *)smem;                                           // [DQ_ROWS][cpad]
    unsigned short *srow = (unsigned short *)(smem + DQ_ROWS * cpad);          // [DQ_ROWS][n_groups]
    uint8_t *zrow = (uint8_t *)(srow + DQ_ROWS * n_groups);                    // [DQ_ROWS][n_groups]
    const int64_t r0 = (int64_t)blockIdx.x * DQ_ROWS;
    const int rows = (int)min((int64_t)DQ_ROWS, r_total - r0);
    for (int i = 0; i < rows; ++i) {
        for (int j = threadIdx.x; j < ci; j += 256) qrow[i * cpad + j] = q[(r0 + i) * c + j];
        for (int g = threadIdx.x; g < n_groups; g += 256) {
            srow[i * n_groups + g] = scale[(r0 + i) * n_groups + g];
            zrow[i * n_groups + g] = zero[(r0 + i) * n_groups + g];
        }
    }
    __syncthreads();
    const bool wide = (ci & 7) == 0 && (((uintptr_t)out) & 15) == 0;
    if (wide) {
        for (int j0 = threadIdx.x * 8; j0 < ci; j0 += 256 * 8) {
            int k[8], g[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                k[e] = perm_inv ? perm_inv[j0 + e] : j0 + e;
                g[e] = group_size > 0 ? k[e] / group_size : col_group[k[e]];
            }
            for (int i = 0; i < rows; ++i) {
                unsigned short h[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) h[e] = deq16<DTYPE>(qrow[i * cpad + k[e]], zrow[i * n_groups + g[e]], srow[i * n_groups + g[e]]);
                u32x4 o;
#pragma unroll
                for (int w = 0; w < 4; ++w) o[w] = (unsigned)h[2 * w] | ((unsigned)h[2 * w + 1] << 16);
                *reinterpret_cast<u32x4 *>(out + (r0 + i) * c + j0) = o;
            }
        }
    } else {
        for (int j = threadIdx.x; j < ci; j += 256) {
            const int k = perm_inv ? perm_inv[j] : j;
            const int g = group_size > 0 ? k / group_size : col_group[k];
            for (int i = 0; i < rows; ++i)
                out[(r0 + i) * c + j] = deq16<DTYPE>(qrow[i * cpad + k], zrow[i * n_groups + g], srow[i * n_groups + g]);
        }
    }
}

// x_out[m][j] = x_in[m][perm[j]]; one workgroup per row, the row staged in LDS
__global__ __launch_bounds__(256) void permute_cols_kernel(const unsigned short *__restrict__ x, const int32_t *__restrict__ perm,
                                                           int64_t c, unsigned short *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned short *row = (unsigned short *)smem;
    const int64_t m = blockIdx.x;
    for (int64_t j = threadIdx.x; j < c; j += 256) row[j] = x[m * c + j];
    __syncthreads();
    for (int64_t j = threadIdx.x; j < c; j += 256) out[m * c + j] = row[perm[j]];
}

// Wide variant (c a multiple of 8, c < 65536, 16-byte aligned rows): a workgroup keeps the
// permutation in LDS as 16-bit indices and walks PR consecutive rows; rows are staged with 16-byte
// loads, every thread gathers 8 neighbouring outputs from LDS and writes them with one 16-byte store.
// NP = 16-byte pieces of a row per thread (c <= NP * 2048): the next row is loaded into registers
// before the current one is gathered and goes to LDS after it, so the load latency hides behind the
// gather instead of stalling every row.
constexpr int PR = 16;
template <int NP>
__global__ __launch_bounds__(256) void permute_cols_wide_kernel(const unsigned short *__restrict__ x, const int32_t *__restrict__ perm,
                                                                int64_t m, int64_t c, unsigned short *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned short *pidx = (unsigned short *)smem;          // c entries
    unsigned short *row = pidx + c;                          // 2 x c entries (two rows in flight)
    const int tid = threadIdx.x;
    const int n16 = (int)(c / 8);
    for (int64_t j = tid; j < c; j += 256) pidx[j] = (unsigned short)perm[j];
    const int64_t m_lo = (int64_t)blockIdx.x * PR, m_hi = m_lo + PR < m ? m_lo + PR : m;
    u32x4 pre[NP];
    auto fetch = [&](int64_t mm) {
        const u32x4 *src = reinterpret_cast<const u32x4 *>(x + mm * c);
#pragma unroll
        for (int i = 0; i < NP; ++i)
            if (tid + 256 * i < n16) pre[i] = src[tid + 256 * i];
    };
    auto put = [&](unsigned short *dst) {
#pragma unroll
        for (int i = 0; i < NP; ++i)
            if (tid + 256 * i < n16) reinterpret_cast<u32x4 *>(dst)[tid + 256 * i] = pre[i];
    };
    if (m_lo < m_hi) {
        fetch(m_lo);
        put(row);
    }
    __syncthreads();
    for (int64_t mm = m_lo; mm < m_hi; ++mm) {
        unsigned short *cur = row + ((mm - m_lo) & 1) * c, *nxt = row + ((mm - m_lo + 1) & 1) * c;
        if (mm + 1 < m_hi) fetch(mm + 1);
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int j = tid + 256 * i;
            if (j < n16) {
                const u32x4 pi = reinterpret_cast<const u32x4 *>(pidx)[j];
                u32x4 o;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const unsigned lo = cur[pi[w] & 0xffffu], hi = cur[pi[w] >> 16];
                    o[w] = lo | (hi << 16);
                }
                reinterpret_cast<u32x4 *>(out + mm * c)[j] = o;
            }
        }
        if (mm + 1 < m_hi) put(nxt);
        __syncthreads();
    }
}


// Packed k-block operand -> dense 16-bit W[r][c] (columns in the operand's own order): what the packed GEMM
// expands tile by tile, written out once.  Eight lanes share one (row, 64-column k-block): a lane turns 8
// levels into 16 bytes, so the eight lanes of a row write one 128-byte line and a wave reads 8 rows x 32 (24)
// contiguous bytes of packed words per k-block.  The arithmetic is deq16: the dense reconstruction's.
template <int DTYPE, int BITS>
__global__ __launch_bounds__(256) void wq_expand_kernel(const uint32_t *__restrict__ packed, const unsigned short *__restrict__ scale_t,
                                                          const uint8_t *__restrict__ zero_t, int64_t r, int64_t c, int group_size,
                                                          unsigned short *__restrict__ out)
{
    constexpr int KB_PER_BLOCK = 4;
    const int lane8 = threadIdx.x & 7;
    const int64_t row = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3);
    if (row >= r) return;
    const int64_t n_kb = c / 64;
#pragma unroll
    for (int i = 0; i < KB_PER_BLOCK; ++i) {
        const int64_t kb = (int64_t)blockIdx.y * KB_PER_BLOCK + i;
        if (kb >= n_kb) break;
        const int64_t col = kb * 64 + lane8 * 8;
        const int64_t g = col / group_size;   // 8 consecutive columns lie in one group (group sizes are multiples of 32)
        const unsigned short s_raw = scale_t[g * r + row];
        const int z = zero_t[g * r + row];
        unsigned short h[8];
        if constexpr (BITS == 4) {
            const uint32_t word = packed[(kb * r + row) * 8 + lane8];
#pragma unroll
            for (int v = 0; v < 8; ++v) h[v] = deq16<DTYPE>((int)((word >> (4 * w4_pairs_nibble(v))) & 15u), z, s_raw);
        } else {
            const uint32_t *w3 = packed + (kb * r + row) * 6 + (lane8 >> 2) * 3;
            const uint32_t w[3] = {w3[0], w3[1], w3[2]};
#pragma unroll
            for (int v = 0; v < 8; ++v) h[v] = deq16<DTYPE>((int)w3_level(w, (lane8 & 3) * 8 + v), z, s_raw);
        }
        u32x4 o;
#pragma unroll
        for (int w = 0; w < 4; ++w) o[w] = (unsigned)h[2 * w] | ((unsigned)h[2 * w + 1] << 16);
        *reinterpret_cast<u32x4 *>(out + row * c + col) = o;
    }
}

}  // namespace

extern "C" int gq_wq_expand(void *w_out, const uint32_t *qweight_packed, const void *scale_t, const uint8_t *qzero_t,
                            int64_t r, int64_t c, int group_size, int bits, int dtype, gq_stream_t stream)
{
    GQ_REQUIRE(w_out && qweight_packed && scale_t && qzero_t, "gq_wq_expand: NULL pointer");
    GQ_REQUIRE(dtype == GQ_F16 || dtype == GQ_BF16, "gq_wq_expand: bad dtype %d", dtype);
    GQ_REQUIRE(bits == 3 || bits == 4, "gq_wq_expand: bits must be 3 or 4, got %d", bits);
    GQ_REQUIRE(r > 0 && c > 0 && c % 64 == 0, "gq_wq_expand: bad shape r=%lld c=%lld (c must be a multiple of 64)", (long long)r,
               (long long)c);
    GQ_REQUIRE(group_size > 0 && group_size % 32 == 0 && c % group_size == 0,
               "gq_wq_expand: group size %d must be a multiple of 32 dividing c=%lld", group_size, (long long)c);
    GQ_REQUIRE(((uintptr_t)w_out & 15) == 0, "gq_wq_expand: w_out must be 16-byte aligned");
    const dim3 grid((unsigned)gq_cdiv(r, 32), (unsigned)gq_cdiv(c / 64, 4));
    GQ_REQUIRE(grid.y <= 65535, "gq_wq_expand: c=%lld is too wide", (long long)c);
    hipStream_t s = (hipStream_t)stream;
    const unsigned short *sc = (const unsigned short *)scale_t;
    unsigned short *out = (unsigned short *)w_out;
    if (dtype == GQ_F16 && bits == 4) hipLaunchKernelGGL((wq_expand_kernel<GQ_F16, 4>), grid, dim3(256), 0, s, qweight_packed, sc, qzero_t, r, c, group_size, out);
    else if (dtype == GQ_F16) hipLaunchKernelGGL((wq_expand_kernel<GQ_F16, 3>), grid, dim3(256), 0, s, qweight_packed, sc, qzero_t, r, c, group_size, out);
    else if (bits == 4) hipLaunchKernelGGL((wq_expand_kernel<GQ_BF16, 4>), grid, dim3(256), 0, s, qweight_packed, sc, qzero_t, r, c, group_size, out);
    else hipLaunchKernelGGL((wq_expand_kernel<GQ_BF16, 3>), grid, dim3(256), 0, s, qweight_packed, sc, qzero_t, r, c, group_size, out);
    GQ_LAUNCH_CHECK();
    return GQ_OK;
}

extern "C" int gq_pack_rows(const uint8_t *values, int64_t rows, int64_t n, int bits, int layout, uint32_t *packed,
                            gq_stream_t stream)
{
    GQ_REQUIRE(values && packed, "gq_pack_rows: NULL pointer");
    GQ_REQUIRE(rows >= 0 && n >= 0, "gq_pack_rows: bad shape");
    GQ_REQUIRE(bits == 2 || bits == 3 || bits == 4 || bits == 8, "gq_pack_rows: bit width %d not in {2,3,4,8}", bits);
    GQ_REQUIRE(layout == GQ_PACK_STREAM || (layout == GQ_PACK_W4_PAIRS && bits == 4) ||
                   (layout == GQ_PACK_W4_KBLOCK && bits == 4 && n % 64 == 0) ||
                   (layout == GQ_PACK_W3_KBLOCK && bits == 3 && n % 64 == 0),
               "gq_pack_rows: bad layout %d for %d bits, n = %lld", layout, bits, (long long)n);
    const int64_t wpr = gq_cdiv(n * bits, 32);
    if (rows * wpr == 0) return GQ_OK;
    if (layout == GQ_PACK_W3_KBLOCK) {
        hipLaunchKernelGGL(pack_w3_kernel, dim3((unsigned)gq_cdiv(rows * wpr, 256)), dim3(256), 0, (hipStream_t)stream, values, rows, n, packed);
        GQ_LAUNCH_CHECK();
        return GQ_OK;
    }
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned)gq_cdiv(rows * wpr, 256)), dim3(256), 0, (hipStream_t)stream, values, rows, n, bits, layout, wpr, packed);
    GQ_LAUNCH_CHECK();
    return GQ_OK;
}

extern "C" int gq_unpack_rows(const uint32_t *packed, int64_t rows, int64_t n, int bits, int layout, uint8_t *values,
                              gq_stream_t stream)
{
    GQ_REQUIRE(values && packed, "gq_unpack_rows: NULL pointer");
    GQ_REQUIRE(rows >= 0 && n >= 0, "gq_unpack_rows: bad shape");
    GQ_REQUIRE(bits == 2 || bits == 3 || bits == 4 || bits == 8, "gq_unpack_rows: bit width %d not in {2,3,4,8}", bits);
    GQ_REQUIRE(layout == GQ_PACK_STREAM || (layout == GQ_PACK_W4_PAIRS && bits == 4) ||
                   (layout == GQ_PACK_W4_KBLOCK && bits == 4 && n % 64 == 0) ||
                   (layout == GQ_PACK_W3_KBLOCK && bits == 3 && n % 64 == 0),
               "gq_unpack_rows: bad layout %d for %d bits, n = %lld", layout, bits, (long long)n);
    const int64_t wpr = gq_cdiv(n * bits, 32);
    if (rows * n == 0) return GQ_OK;
    if (layout == GQ_PACK_W3_KBLOCK) {
        hipLaunchKernelGGL(unpack_w3_kernel, dim3((unsigned)gq_cdiv(rows * n, 256)), dim3(256), 0, (hipStream_t)stream, packed, rows, n, values);
        GQ_LAUNCH_CHECK();
        return GQ_OK;
    }
    hipLaunchKernelGGL(unpack_kernel, dim3((unsigned)gq_cdiv(rows * n, 256)), dim3(256), 0, (hipStream_t)stream, packed, rows, n, bits, layout, wpr, values);
    GQ_LAUNCH_CHECK();
    return GQ_OK;
}

extern "C" int gq_dequantize(const uint8_t *qweight, const void *scale, const uint8_t *qzero, const int32_t *perm_inv,
                             int64_t r, int64_t c, int n_groups, int group_size, const int32_t *col_group, int dtype,
                             void *weight_out, gq_stream_t stream)
{
    GQ_REQUIRE(qweight && scale && qzero && weight_out, "gq_dequantize: NULL pointer");
    GQ_REQUIRE(dtype == GQ_F16 || dtype == GQ_BF16, "gq_dequantize: bad dtype %d", dtype);
    GQ_REQUIRE(r > 0 && c > 0 && n_groups > 0, "gq_dequantize: bad shape");
    GQ_REQUIRE(DQ_ROWS * (c + 16 + 3 * (int64_t)n_groups) + 32 <= 160 * 1024, "gq_dequantize: %d rows of %lld columns in %d groups do not fit in LDS", DQ_ROWS, (long long)c, n_groups);
    GQ_REQUIRE((group_size > 0 && (int64_t)group_size * n_groups >= c) || (group_size == 0 && col_group),
               "gq_dequantize: need a uniform group_size covering all columns, or group_size 0 with col_group");
    const size_t lds = (size_t)DQ_ROWS * (((c + 15) & ~15) + (size_t)n_groups * 3) + 16;   // rows of levels, their scales and zero points
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GQ_F16) {
        GQ_HIP(hipFuncSetAttribute((const void *)dequant_kernel<GQ_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(dequant_kernel<GQ_F16>, dim3((unsigned)gq_cdiv(r, DQ_ROWS)), dim3(256), lds, s, qweight, (const unsigned short *)scale, qzero, perm_inv, r, c, n_groups, group_size, col_group, (unsigned short *)weight_out);
    } else {
        GQ_HIP(hipFuncSetAttribute((const void *)dequant_kernel<GQ_BF16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(dequant_kernel<GQ_BF16>, dim3((unsigned)gq_cdiv(r, DQ_ROWS)), dim3(256), lds, s, qweight, (const unsigned short *)scale, qzero, perm_inv, r, c, n_groups, group_size, col_group, (unsigned short *)weight_out);
    }
    GQ_LAUNCH_CHECK();
    return GQ_OK;
}

// ---- lower triangle of a symmetric fp32 matrix <-> one flat buffer --------------------------------
// Element (r, c <= r) lies at r (r + 1) / 2 + c.  What travels between GPUs when Hessians are summed over ranks:
// half the matrix instead of all of it.
namespace {

// one 64 x 64 tile (tr >= tc) per workgroup; thread = (row in groups of 4, column)
__global__ __launch_bounds__(256) void pack_lower_kernel(const float *__restrict__ h, int64_t c, float *__restrict__ flat)
{
    const int64_t tr = blockIdx.y, tc = blockIdx.x;
    if (tc > tr) return;
    const int j = threadIdx.x & 63;
    const int64_t col = tc * 64 + j;
#pragma unroll 4
    for (int pass = 0; pass < 16; ++pass) {
        const int64_t row = tr * 64 + (threadIdx.x >> 6) + 4 * pass;
        if (row < c && col <= row) flat[row * (row + 1) / 2 + col] = h[row * c + col];
    }
}

// the same tile back, and its mirror image above the diagonal: H comes out exactly symmetric
__global__ __launch_bounds__(256) void unpack_lower_kernel(const float *__restrict__ flat, int64_t c, float *__restrict__ h)
{
    __shared__ float tile[64][65];
    const int64_t tr = blockIdx.y, tc = blockIdx.x;
    if (tc > tr) return;
    const int j = threadIdx.x & 63;
#pragma unroll 4
    for (int pass = 0; pass < 16; ++pass) {
        const int i = (threadIdx.x >> 6) + 4 * pass;
        const int64_t row = tr * 64 + i, col = tc * 64 + j;
        tile[i][j] = (row < c && col <= row) ? flat[row * (row + 1) / 2 + col] : 0.0f;
    }
    __syncthreads();
#pragma unroll 4
    for (int pass = 0; pass < 16; ++pass) {
        const int i = (threadIdx.x >> 6) + 4 * pass;
        const int64_t row = tr * 64 + i, col = tc * 64 + j;
        if (row < c && col < c) {
            if (tr != tc) h[row * c + col] = tile[i][j];
            else h[row * c + col] = j <= i ? tile[i][j] : tile[j][i];
        }
        // the mirrored tile: element (tc * 64 + i, tr * 64 + j) = tile[j][i]
        const int64_t mrow = tc * 64 + i, mcol = tr * 64 + j;
        if (tr != tc && mrow < c && mcol < c) h[mrow * c + mcol] = tile[j][i];
    }
}

}  // namespace

extern "C" int gq_pack_lower(const float *hessian, int64_t c, float *flat, gq_stream_t stream)
{
    GQ_REQUIRE(hessian && flat && c > 0 && c <= (1 << 22), "gq_pack_lower: bad argument");
    const unsigned nt = (unsigned)gq_cdiv(c, 64);
    hipLaunchKernelGGL(pack_lower_kernel, dim3(nt, nt), dim3(256), 0, (hipStream_t)stream, hessian, c, flat);
    GQ_LAUNCH_CHECK();
    return GQ_OK;
}

extern "C" int gq_unpack_lower(const float *flat, int64_t c, float *hessian, gq_stream_t stream)
{
    GQ_REQUIRE(hessian && flat && c > 0 && c <= (1 << 22), "gq_unpack_lower: bad argument");
    const unsigned nt = (unsigned)gq_cdiv(c, 64);
    hipLaunchKernelGGL(unpack_lower_kernel, dim3(nt, nt), dim3(256), 0, (hipStream_t)stream, flat, c, hessian);
    GQ_LAUNCH_CHECK();
    return GQ_OK;
}

extern "C" int gq_permute_columns(const void *x_in, const int32_t *perm, int64_t m, int64_t c, void *x_out, gq_stream_t stream)
{
    GQ_REQUIRE(m >= 0 && c > 0 && c <= 80 * 1024, "gq_permute_columns: bad shape");
    GQ_REQUIRE(perm && ((x_in && x_out) || m == 0), "gq_permute_columns: NULL pointer");   // no rows: no data pointers
    if (m == 0) return GQ_OK;
    GQ_REQUIRE(x_in != x_out, "gq_permute_columns: in-place is not supported");
    if (c % 8 == 0 && c <= 8 * 2048 && (((uintptr_t)x_in | (uintptr_t)x_out) & 15) == 0) {
        const size_t lds_w = (size_t)(c * 6);
        const int np = (int)gq_cdiv(c / 8, 256);
        void (*kern)(const unsigned short *, const int32_t *, int64_t, int64_t, unsigned short *) =
            np <= 2 ? permute_cols_wide_kernel<2> : np <= 4 ? permute_cols_wide_kernel<4>
            : np <= 6 ? permute_cols_wide_kernel<6> : permute_cols_wide_kernel<8>;
        GQ_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_w));
        hipLaunchKernelGGL(kern, dim3((unsigned)gq_cdiv(m, PR)), dim3(256), lds_w, (hipStream_t)stream,
                           (const unsigned short *)x_in, perm, m, c, (unsigned short *)x_out);
        GQ_LAUNCH_CHECK();
        return GQ_OK;
    }
    const size_t lds = (size_t)((c * 2 + 15) & ~15);
    GQ_HIP(hipFuncSetAttribute((const void *)permute_cols_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(permute_cols_kernel, dim3((unsigned)m), dim3(256), lds, (hipStream_t)stream, (const unsigned short *)x_in, perm, c, (unsigned short *)x_out);
    GQ_LAUNCH_CHECK();
    return GQ_OK;
}
