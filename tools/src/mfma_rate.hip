// Issue-rate probe for v_mfma_f32_16x16x32_f16: every wave runs a loop of NACC independent MFMAs on operands
// that never change (so the figure is an upper bound: operands that toggle cost more power); one workgroup per CU of WAVES waves (4 = one per SIMD, 8 = two per SIMD).  Prints the
// sustained TFLOP/s of back-to-back launches.  Measured: 1950-2020 TFLOP/s with one or two waves per SIMD and
// up to 32 accumulators per wave; 64 accumulators through the builtin fall to 710 (hipcc moves them between
// AGPRs and VGPRs around every MFMA).  build: hipcc --offload-arch=gfx950 -O3 tools/src/mfma_rate.hip -o
// tools/_build/mfma_rate;  usage: mfma_rate [launches]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int NOPS>
__global__ __launch_bounds__(512) void probe(float *out, int iters, const _Float16 *seed)
{
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = seed[(threadIdx.x + i) & 63]; b[i] = seed[(threadIdx.x * 3 + i) & 63]; }
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
            if (NOPS) asm volatile("s_nop 0");
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <typename K>
static void run(const char *name, K kern, int waves, int nacc, int launches, float *out, const _Float16 *seed)
{
    int dev = 0, cus = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int iters = 200000 / nacc;
    hipEvent_t s, e;
    (void)hipEventCreate(&s); (void)hipEventCreate(&e);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(cus), dim3(waves * 64), 0, 0, out, iters, seed);
    (void)hipEventRecord(s);
    for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(kern, dim3(cus), dim3(waves * 64), 0, 0, out, iters, seed);
    (void)hipEventRecord(e);
    (void)hipEventSynchronize(e);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, s, e);
    const double flops = 2.0 * 16 * 16 * 32 * (double)nacc * iters * waves * cus * launches;
    printf("%-28s %d waves/CU  %8.3f ms/launch  %8.1f TFLOP/s\n", name, waves, ms / launches, flops / (ms * 1e-3) / 1e12);
}

int main(int argc, char **argv)
{
    const int launches = argc > 1 ? atoi(argv[1]) : 10;
    float *out; _Float16 *seed;
    (void)hipMalloc(&out, 4096);
    (void)hipMalloc(&seed, 128);
    _Float16 h[64];
    for (int i = 0; i < 64; ++i) h[i] = (_Float16)(0.37f * ((i * 7919) % 13) - 2.0f);
    (void)hipMemcpy(seed, h, 128, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run("64 acc, 1 wave/SIMD", probe<64, 0>, 4, 64, launches, out, seed);
        run("64 acc + s_nop, 1 wave/SIMD", probe<64, 1>, 4, 64, launches, out, seed);
        run("32 acc, 1 wave/SIMD", probe<32, 0>, 4, 32, launches, out, seed);
        run("32 acc, 2 waves/SIMD", probe<32, 0>, 8, 32, launches, out, seed);
        run("32 acc + s_nop, 2 waves/SIMD", probe<32, 1>, 8, 32, launches, out, seed);
        run("16 acc, 2 waves/SIMD", probe<16, 0>, 8, 16, launches, out, seed);
        run("8 acc, 1 wave/SIMD", probe<8, 0>, 4, 8, launches, out, seed);
        run("4 acc, 1 wave/SIMD", probe<4, 0>, 4, 4, launches, out, seed);
    }
    return 0;
}
