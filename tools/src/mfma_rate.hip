// Issue-rate probe for v_mfma_f32_16x16x32_f16: every wave runs a loop of NACC independent MFMAs on operands
// that never change (so the figure is an upper bound: operands that toggle cost more power); one workgroup per CU of WAVES waves (4 = one per SIMD, 8 = two per SIMD).  Prints the
// sustained TFLOP/s of back-to-back launches.  Measured: 1950-2020 TFLOP/s with one or two waves per SIMD and
// up to 32 accumulators per wave; 64 accumulators through the builtin fall to 710 (hipcc moves them between
// AGPRs and VGPRs around every MFMA).  build: hipcc --offload-arch=gfx950 -O3 tools/src/mfma_rate.hip -o
// tools/_build/mfma_rate;  usage: mfma_rate [launches]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

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

// every accumulation register, as a clobber list: hipcc then keeps nothing of its own in the AGPRs the asm names
#define GQ_ALL_AGPRS "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255"
template <int B, int E, typename F>
__device__ __forceinline__ void sfor(F &&f)
{
    if constexpr (B < E) { f(std::integral_constant<int, B>{}); sfor<B + 1, E>(f); }
}
// 64 accumulators in hand-assigned AGPRs; ORDER 0: a[0:3], a[4:7], ... ; 1: a[0:3], a[32:35], ... (stride 32)
template <int ORDER, int NOPS, int TWO_SETS>
__global__ __launch_bounds__(256) void probe_agpr(float *out, int iters, const _Float16 *seed)
{
    f16x8 a[2], b[2];
    for (int i = 0; i < 8; ++i) {
        a[0][i] = seed[(threadIdx.x + i) & 63]; b[0][i] = seed[(threadIdx.x * 3 + i) & 63];
        a[1][i] = seed[(threadIdx.x * 5 + i) & 63]; b[1][i] = seed[(threadIdx.x * 7 + i) & 63];
    }
    sfor<0, 256>([&](auto n) { asm volatile("v_accvgpr_write_b32 a[%0], 0" ::"n"(decltype(n)::value) : GQ_ALL_AGPRS); });
    for (int it = 0; it < iters; it += 2) {
        sfor<0, 128>([&](auto ic) {
            constexpr int i = decltype(ic)::value & 63;
            constexpr int N = ORDER ? 4 * (8 * (i & 7) + (i >> 3)) : 4 * i;
            constexpr int set = TWO_SETS ? (decltype(ic)::value >> 6) : 0;
            // TWO_SETS = 2: a different operand pair for every MFMA (eight fragments per side, as a GEMM has)
            constexpr int sa = TWO_SETS == 2 ? (i & 1) : set, sb = TWO_SETS == 2 ? ((i >> 3) & 1) : set;
            asm volatile("v_mfma_f32_16x16x32_f16 a[%2:%3], %0, %1, a[%2:%3]" ::"v"(a[sa]), "v"(b[sb]), "n"(N), "n"(N + 3) : GQ_ALL_AGPRS);
            if (NOPS) asm volatile("s_nop 0");
        });
    }
    asm volatile("s_nop 15\n\ts_nop 15");
    float v;
    asm volatile("v_accvgpr_read_b32 %0, a[17]" : "=v"(v) :: GQ_ALL_AGPRS);
    if (v == 12345.678f) out[threadIdx.x] = v;
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
    for (int i = 0; i < 64; ++i) h[i] = (_Float16)(0.000731f * ((i * 7919 + 13) % 4093) - 1.5f);
    (void)hipMemcpy(seed, h, 128, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run("64 acc, 1 wave/SIMD", probe<64, 0>, 4, 64, launches, out, seed);
        run("64 acc + s_nop, 1 wave/SIMD", probe<64, 1>, 4, 64, launches, out, seed);
        run("32 acc, 1 wave/SIMD", probe<32, 0>, 4, 32, launches, out, seed);
        run("32 acc, 2 waves/SIMD", probe<32, 0>, 8, 32, launches, out, seed);
        run("32 acc + s_nop, 2 waves/SIMD", probe<32, 1>, 8, 32, launches, out, seed);
        run("16 acc, 2 waves/SIMD", probe<16, 0>, 8, 16, launches, out, seed);
        run("agpr 64 seq, 1 wave/SIMD", probe_agpr<0, 0, 0>, 4, 64, launches, out, seed);
        run("agpr 64 stride32", probe_agpr<1, 0, 0>, 4, 64, launches, out, seed);
        run("agpr 64 seq + s_nop", probe_agpr<0, 1, 0>, 4, 64, launches, out, seed);
        run("agpr 64 stride32 + s_nop", probe_agpr<1, 1, 0>, 4, 64, launches, out, seed);
        run("agpr 64, operands alternate per 64", probe_agpr<0, 0, 1>, 4, 64, launches, out, seed);
        run("agpr 64, operands alternate per MFMA", probe_agpr<0, 0, 2>, 4, 64, launches, out, seed);
        run("8 acc, 1 wave/SIMD", probe<8, 0>, 4, 8, launches, out, seed);
        run("4 acc, 1 wave/SIMD", probe<4, 0>, 4, 4, launches, out, seed);
    }
    return 0;
}
