// The bandwidth-bound operators of a Llama block that stand between the calibration path's GEMMs when the block is
// replayed stage by stage: RMS normalisation, rotary position embedding and act(gate) * up.  As separate framework
// operators they are seven, ten and two passes over the activations; here each is one (read once, write once).
//
// All keep the roundings of the operator chains they replace, so that a replay through them sees the activations
// the model's own modules would produce:
//   norm:  x32 = float(x); v = mean(x32 * x32); y = x32 * rsqrt(v + eps); out = T(float(w) * float(T(y)))
//   rope:  out = T(float(T(x * cos)) + float(T(rotate_half(x) * sin)))
//   silu:  s = x32 / (1 + exp(-x32)); out = T(float(T(s)) * float(u))
// (T = the 16-bit type of the model).  Only the order in which the norm's squares are summed is a kernel's own.
#include "gq_common.h"

namespace {

template <typename T>
struct Vec8 {
    T v[8];
};

template <typename T>
__device__ __forceinline__ void load8(const T *p, float (&f)[8])
{
    const u32x4 raw = *reinterpret_cast<const u32x4 *>(p);
    Vec8<T> u;
    __builtin_memcpy(&u, &raw, 16);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = static_cast<float>(u.v[i]);
}

template <typename T>
__device__ __forceinline__ void store8(T *p, const T (&t)[8])
{
    u32x4 raw;
    __builtin_memcpy(&raw, t, 16);
    *reinterpret_cast<u32x4 *>(p) = raw;
}

// One wave per row, four rows per workgroup.  A row of up to 8192 elements stays in registers between the sum and
// the scaling (16 chunks of 8 per lane at most); longer rows are read a second time (from the cache: a row is a few
// tens of KiB).  hidden is a multiple of 8.
template <typename T, int CHUNKS>
__global__ __launch_bounds__(256) void rmsnorm_kernel(T *__restrict__ out, const T *__restrict__ x, const T *__restrict__ w,
                                                      int64_t n_rows, int hidden, float eps)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;   // wave-uniform
    const T *xr = x + row * hidden;
    T *orow = out + row * hidden;
    const int n_chunks = hidden >> 3;          // 16-byte chunks of the row
    float sum = 0.f;
    if constexpr (CHUNKS > 0) {
        float f[CHUNKS][8];
#pragma unroll
        for (int q = 0; q < CHUNKS; ++q) {
            const int ch = lane + 64 * q;
            if (ch < n_chunks) load8(xr + 8 * ch, f[q]);
            else
#pragma unroll
                for (int i = 0; i < 8; ++i) f[q][i] = 0.f;
        }
#pragma unroll
        for (int q = 0; q < CHUNKS; ++q)
#pragma unroll
            for (int i = 0; i < 8; ++i) sum += __fmul_rn(f[q][i], f[q][i]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
        const float r = rsqrtf(__fmul_rn(sum, 1.0f / (float)hidden) + eps);
#pragma unroll
        for (int q = 0; q < CHUNKS; ++q) {
            const int ch = lane + 64 * q;
            if (ch < n_chunks) {
                float wf[8];
                load8(w + 8 * ch, wf);
                T o[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = static_cast<T>(__fmul_rn(wf[i], static_cast<float>(static_cast<T>(__fmul_rn(f[q][i], r)))));
                store8(orow + 8 * ch, o);
            }
        }
    } else {
        for (int ch = lane; ch < n_chunks; ch += 64) {
            float f[8];
            load8(xr + 8 * ch, f);
#pragma unroll
            for (int i = 0; i < 8; ++i) sum += __fmul_rn(f[i], f[i]);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
        const float r = rsqrtf(__fmul_rn(sum, 1.0f / (float)hidden) + eps);
        for (int ch = lane; ch < n_chunks; ch += 64) {
            float f[8], wf[8];
            load8(xr + 8 * ch, f);
            load8(w + 8 * ch, wf);
            T o[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = static_cast<T>(__fmul_rn(wf[i], static_cast<float>(static_cast<T>(__fmul_rn(f[i], r)))));
            store8(orow + 8 * ch, o);
        }
    }
}

// out[i] = T(float(T(silu(gate[i]))) * float(up[i])), 8 elements per thread and turn, grid-stride
template <typename T>
__global__ __launch_bounds__(256) void silu_mul_kernel(T *__restrict__ out, const T *__restrict__ gate, const T *__restrict__ up,
                                                       int64_t n_chunks)
{
    for (int64_t ch = (int64_t)blockIdx.x * 256 + threadIdx.x; ch < n_chunks; ch += (int64_t)gridDim.x * 256) {
        float g[8], u[8];
        load8(gate + 8 * ch, g);
        load8(up + 8 * ch, u);
        T o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float s = g[i] / (1.0f + expf(-g[i]));
            o[i] = static_cast<T>(__fmul_rn(static_cast<float>(static_cast<T>(s)), u[i]));
        }
        store8(out + 8 * ch, o);
    }
}

// Rotary position embedding of `n_heads` adjacent heads per token, in place: for the two halves (lo, hi) of a head
//   lo' = T(float(T(lo * cos_lo)) + float(T(-hi * sin_lo))),  hi' = T(float(T(hi * cos_hi)) + float(T(lo * sin_hi)))
// -- the roundings of  x * cos + rotate_half(x) * sin  taken operator by operator in the 16-bit type.  Eight lanes
// cover one head (8 elements of each half per lane and turn), a wave eight heads of one token.
template <typename T>
__global__ __launch_bounds__(256) void rope_kernel(T *__restrict__ x, int64_t row_stride, const T *__restrict__ cos_t,
                                                   const T *__restrict__ sin_t, int64_t cs_batch_stride, int64_t seq,
                                                   int64_t n_tokens, int n_heads, int head_dim)
{
    const int half = head_dim >> 1;
    const int chunks = half >> 3;                        // 8-element chunks per half head
    const int64_t per_token = (int64_t)n_heads * chunks;   // work items of one token
    const int64_t total = n_tokens * per_token;
    for (int64_t it = (int64_t)blockIdx.x * 256 + threadIdx.x; it < total; it += (int64_t)gridDim.x * 256) {
        const int64_t tok = it / per_token;
        const int rem = (int)(it - tok * per_token);
        const int head = rem / chunks, ch = rem - head * chunks;
        const int64_t b = tok / seq, pos = tok - b * seq;
        T *xp = x + tok * row_stride + (int64_t)head * head_dim + 8 * ch;
        const T *cp = cos_t + b * cs_batch_stride + pos * head_dim + 8 * ch;
        const T *sp = sin_t + b * cs_batch_stride + pos * head_dim + 8 * ch;
        float lo[8], hi[8], cl[8], chh[8], sl[8], sh[8];
        load8(xp, lo);
        load8(xp + half, hi);
        load8(cp, cl);
        load8(cp + half, chh);
        load8(sp, sl);
        load8(sp + half, sh);
        T olo[8], ohi[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float a0 = static_cast<float>(static_cast<T>(__fmul_rn(lo[i], cl[i])));
            const float b0 = static_cast<float>(static_cast<T>(__fmul_rn(-hi[i], sl[i])));
            const float a1 = static_cast<float>(static_cast<T>(__fmul_rn(hi[i], chh[i])));
            const float b1 = static_cast<float>(static_cast<T>(__fmul_rn(lo[i], sh[i])));
            olo[i] = static_cast<T>(__fadd_rn(a0, b0));
            ohi[i] = static_cast<T>(__fadd_rn(a1, b1));
        }
        store8(xp, olo);
        store8(xp + half, ohi);
    }
}

template <typename T>
int launch_rmsnorm(void *out, const void *x, const void *w, int64_t n_rows, int64_t hidden, float eps, hipStream_t s)
{
    const unsigned grid = (unsigned)gq_cdiv(n_rows, 4);
    const int64_t chunks_per_lane = gq_cdiv(hidden / 8, 64);
#define GQ_NORM(C) hipLaunchKernelGGL((rmsnorm_kernel<T, C>), dim3(grid), dim3(256), 0, s, (T *)out, (const T *)x, (const T *)w, n_rows, (int)hidden, eps)
    if (chunks_per_lane <= 4) GQ_NORM(4);
    else if (chunks_per_lane <= 8) GQ_NORM(8);
    else if (chunks_per_lane <= 16) GQ_NORM(16);
    else GQ_NORM(0);
#undef GQ_NORM
    GQ_LAUNCH_CHECK();
    return GQ_OK;
}

}  // namespace

extern "C" int gq_rmsnorm(void *out, const void *x, const void *weight, int dtype, int64_t n_rows, int64_t hidden, float eps,
                          gq_stream_t stream)
{
    GQ_REQUIRE(out && x && weight, "gq_rmsnorm: NULL pointer");
    GQ_REQUIRE(dtype == GQ_F16 || dtype == GQ_BF16, "gq_rmsnorm: dtype %d is neither fp16 (0) nor bf16 (1)", dtype);
    GQ_REQUIRE(n_rows >= 0 && hidden > 0 && hidden % 8 == 0 && hidden < (1 << 30), "gq_rmsnorm: hidden must be a positive multiple of 8");
    GQ_REQUIRE(n_rows < ((int64_t)1 << 33), "gq_rmsnorm: too many rows");
    GQ_REQUIRE((((uintptr_t)out | (uintptr_t)x | (uintptr_t)weight) & 15) == 0, "gq_rmsnorm: pointers must be 16-byte aligned");
    if (n_rows == 0) return GQ_OK;
    return dtype == GQ_F16 ? launch_rmsnorm<_Float16>(out, x, weight, n_rows, hidden, eps, (hipStream_t)stream)
                           : launch_rmsnorm<__bf16>(out, x, weight, n_rows, hidden, eps, (hipStream_t)stream);
}

extern "C" int gq_silu_mul(void *out, const void *gate, const void *up, int dtype, int64_t n, gq_stream_t stream)
{
    GQ_REQUIRE(out && gate && up, "gq_silu_mul: NULL pointer");
    GQ_REQUIRE(dtype == GQ_F16 || dtype == GQ_BF16, "gq_silu_mul: dtype %d is neither fp16 (0) nor bf16 (1)", dtype);
    GQ_REQUIRE(n >= 0 && n % 8 == 0, "gq_silu_mul: the element count must be a multiple of 8");
    GQ_REQUIRE((((uintptr_t)out | (uintptr_t)gate | (uintptr_t)up) & 15) == 0, "gq_silu_mul: pointers must be 16-byte aligned");
    if (n == 0) return GQ_OK;
    const int64_t n_chunks = n / 8;
    const unsigned grid = (unsigned)std::min<int64_t>(gq_cdiv(n_chunks, 256), 256 * 32);
    if (dtype == GQ_F16)
        hipLaunchKernelGGL(silu_mul_kernel<_Float16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (_Float16 *)out, (const _Float16 *)gate,
                           (const _Float16 *)up, n_chunks);
    else
        hipLaunchKernelGGL(silu_mul_kernel<__bf16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (__bf16 *)out, (const __bf16 *)gate,
                           (const __bf16 *)up, n_chunks);
    GQ_LAUNCH_CHECK();
    return GQ_OK;
}

extern "C" int gq_rope_inplace(void *x, int64_t row_stride, const void *cos_t, const void *sin_t, int64_t cs_batch_stride, int64_t n_batch,
                               int64_t seq, int n_heads, int head_dim, int dtype, gq_stream_t stream)
{
    GQ_REQUIRE(x && cos_t && sin_t, "gq_rope_inplace: NULL pointer");
    GQ_REQUIRE(dtype == GQ_F16 || dtype == GQ_BF16, "gq_rope_inplace: dtype %d is neither fp16 (0) nor bf16 (1)", dtype);
    GQ_REQUIRE(n_batch >= 0 && seq >= 0 && n_heads > 0 && head_dim > 0 && head_dim % 16 == 0,
               "gq_rope_inplace: head_dim must be a positive multiple of 16");
    GQ_REQUIRE(row_stride >= (int64_t)n_heads * head_dim && row_stride % 8 == 0 && cs_batch_stride % 8 == 0 && cs_batch_stride >= 0,
               "gq_rope_inplace: row stride %lld must cover the %d heads and be a multiple of 8", (long long)row_stride, n_heads);
    GQ_REQUIRE((((uintptr_t)x | (uintptr_t)cos_t | (uintptr_t)sin_t) & 15) == 0, "gq_rope_inplace: pointers must be 16-byte aligned");
    const int64_t n_tokens = n_batch * seq;
    if (n_tokens == 0) return GQ_OK;
    const int64_t total = n_tokens * n_heads * (head_dim / 16);
    const unsigned grid = (unsigned)std::min<int64_t>(gq_cdiv(total, 256), 256 * 32);
    if (dtype == GQ_F16)
        hipLaunchKernelGGL(rope_kernel<_Float16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (_Float16 *)x, row_stride, (const _Float16 *)cos_t,
                           (const _Float16 *)sin_t, cs_batch_stride, seq, n_tokens, n_heads, head_dim);
    else
        hipLaunchKernelGGL(rope_kernel<__bf16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (__bf16 *)x, row_stride, (const __bf16 *)cos_t,
                           (const __bf16 *)sin_t, cs_batch_stride, seq, n_tokens, n_heads, head_dim);
    GQ_LAUNCH_CHECK();
    return GQ_OK;
}
