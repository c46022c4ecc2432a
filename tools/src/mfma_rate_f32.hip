// Issue-rate probe for v_mfma_f32_32x32x2_f32 (the exact-fp32 matrix instruction of the fp32 GEMM): every wave
// runs NACC independent accumulators on operands that never change; one workgroup per CU of WAVES waves.
// Prints the sustained TFLOP/s of back-to-back launches: the ceiling an fp32 GEMM can reach on this chip.
// build: hipcc --offload-arch=gfx950 -O3 tools/src/mfma_rate_f32.hip -o tools/_build/mfma_rate_f32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int SHAPE>
__global__ __launch_bounds__(512) void probe(float *out, int iters, const float *seed)
{
    float a = seed[threadIdx.x & 63], b = seed[(threadIdx.x * 3) & 63];
    if constexpr (SHAPE == 32) {
        f32x16 acc[NACC];
#pragma unroll
        for (int i = 0; i < NACC; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][5];
        if (s == 12345.678f) out[threadIdx.x] = s;
    } else {
        f32x4 acc[NACC];
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
        if (s == 12345.678f) out[threadIdx.x] = s;
    }
}

template <typename K>
static void run(const char *name, K kern, int waves, int nacc, double flops_per_mfma, int launches, float *out, const float *seed)
{
    int dev = 0, cus = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int iters = 40000 / nacc;
    hipEvent_t s, e;
    (void)hipEventCreate(&s); (void)hipEventCreate(&e);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(cus), dim3(waves * 64), 0, 0, out, iters, seed);
    (void)hipEventRecord(s);
    for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(kern, dim3(cus), dim3(waves * 64), 0, 0, out, iters, seed);
    (void)hipEventRecord(e);
    (void)hipEventSynchronize(e);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, s, e);
    const double flops = flops_per_mfma * (double)nacc * iters * waves * cus * launches;
    printf("%-28s %d waves/CU  %8.3f ms/launch  %8.1f TFLOP/s\n", name, waves, ms / launches, flops / (ms * 1e-3) / 1e12);
}

int main(int argc, char **argv)
{
    const int launches = argc > 1 ? atoi(argv[1]) : 10;
    float *out, *seed;
    (void)hipMalloc(&out, 4096);
    (void)hipMalloc(&seed, 256);
    float h[64];
    for (int i = 0; i < 64; ++i) h[i] = 0.000731f * ((i * 7919 + 13) % 4093) - 1.5f;
    (void)hipMemcpy(seed, h, 256, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run("32x32x2 4 acc, 1 wave/SIMD", probe<4, 32>, 4, 4, 2.0 * 32 * 32 * 2, launches, out, seed);
        run("32x32x2 4 acc, 2 waves/SIMD", probe<4, 32>, 8, 4, 2.0 * 32 * 32 * 2, launches, out, seed);
        run("32x32x2 8 acc, 1 wave/SIMD", probe<8, 32>, 4, 8, 2.0 * 32 * 32 * 2, launches, out, seed);
        run("16x16x4 16 acc, 1 wave/SIMD", probe<16, 16>, 4, 16, 2.0 * 16 * 16 * 4, launches, out, seed);
        run("16x16x4 16 acc, 2 waves/SIMD", probe<16, 16>, 8, 16, 2.0 * 16 * 16 * 4, launches, out, seed);
    }
    return 0;
}
