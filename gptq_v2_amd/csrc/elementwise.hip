// The two bandwidth-bound operators of a Llama block that stand between the calibration path's GEMMs when the
// block is replayed stage by stage: RMS normalisation and act(gate) * up.  As separate framework operators they
// are seven and two passes over the activations; here each is one (read once, write once).
//
// Both keep the roundings of the operator chains they replace, so that a replay through them sees the activations
// the model's own modules would produce:
//   norm:  x32 = float(x); v = mean(x32 * x32); y = x32 * rsqrt(v + eps); out = T(float(w) * float(T(y)))
//   silu:  s = x32 / (1 + exp(-x32)); out = T(float(T(s)) * float(u))
// (T = the 16-bit type of the model).  Only the order in which the squares are summed is this kernel's own.
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
