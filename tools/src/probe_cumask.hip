// Does a chain of short kernels start beside a chip-filling kernel of another stream when that stream is
// confined to a part of the CUs (hipExtStreamCreateWithCUMask)?  A "filler" of many workgroups that each
// spin for a fixed wall time stands for a large update GEMM, a chain of one-workgroup kernels that spin a few
// microseconds each for the column loop's or the factorisation's chain.  Every loop ends by the clock.
// Prints the chain's time alone, beside an unmasked filler, and beside fillers with different masks.
// build: hipcc --offload-arch=gfx950 -O3 tools/src/probe_cumask.hip -o tools/_build/probe_cumask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                            \
    do {                                                                                    \
        hipError_t e_ = (x);                                                                \
        if (e_ != hipSuccess) {                                                             \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                        \
        }                                                                                   \
    } while (0)

// wall clock at 100 MHz
__device__ __forceinline__ unsigned long long now() { return __builtin_amdgcn_s_memrealtime(); }

__global__ __launch_bounds__(256) void spin_kernel(unsigned ticks, unsigned *sink)
{
    const unsigned long long t0 = now();
    unsigned n = 0;
    while (now() - t0 < ticks) ++n;
    if (n == 0xffffffffu) *sink = n;
}

static float chain_ms(hipStream_t chain, int links, int wgs, unsigned ticks, unsigned *sink, hipStream_t other, int filler_wgs,
                      unsigned filler_ticks)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    CHECK(hipDeviceSynchronize());
    if (other) hipLaunchKernelGGL(spin_kernel, dim3(filler_wgs), dim3(256), 0, other, filler_ticks, sink);
    CHECK(hipEventRecord(a, chain));
    for (int i = 0; i < links; ++i) hipLaunchKernelGGL(spin_kernel, dim3(wgs), dim3(256), 0, chain, ticks, sink);
    CHECK(hipEventRecord(b, chain));
    CHECK(hipDeviceSynchronize());
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, a, b));
    CHECK(hipEventDestroy(a));
    CHECK(hipEventDestroy(b));
    return ms;
}

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    unsigned *sink;
    CHECK(hipMalloc(&sink, 4));
    int cus = 0;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    hipStream_t chain, plain;
    CHECK(hipStreamCreate(&chain));
    CHECK(hipStreamCreate(&plain));
    const int words = (cus + 31) / 32;
    struct Mask { const char *name; std::vector<uint32_t> bits; };
    std::vector<Mask> masks;
    {   // the first three quarters of the CU indices
        Mask m{"CUs 0 .. 3/4", std::vector<uint32_t>(words, 0u)};
        for (int i = 0; i < cus * 3 / 4; ++i) m.bits[i / 32] |= 1u << (i % 32);
        masks.push_back(m);
    }
    // (a mask of three of every four CU indices made hipExtStreamCreateWithCUMask crash on the host here)
    {   // all but the last 32 indices
        Mask m{"all but the last 32", std::vector<uint32_t>(words, 0u)};
        for (int i = 0; i < cus - 32; ++i) m.bits[i / 32] |= 1u << (i % 32);
        masks.push_back(m);
    }
    const unsigned link_ticks = 1000;      // 10 us per link
    const unsigned filler_ticks = 20000;   // 200 us per workgroup
    const int filler_wgs = cus * 8 * 6;    // eight workgroups of 256 threads fit a CU: six rounds, 1.2 ms on all CUs
    // (the streams are made once and kept: making a second one after destroying the first ended in a segmentation
    // fault inside the runtime)
    std::vector<hipStream_t> masked(masks.size(), nullptr);
    for (size_t i = 0; i < masks.size(); ++i) {
        printf("(creating a stream with mask '%s', %d words)\n", masks[i].name, words);
        CHECK(hipExtStreamCreateWithCUMask(&masked[i], (uint32_t)words, masks[i].bits.data()));
    }
    for (int wgs : {1, 64, 172}) {
        printf("chain of 100 links of %d workgroup(s), 10 us each; filler %d workgroups of 200 us\n", wgs, filler_wgs);
        chain_ms(chain, 100, wgs, link_ticks, sink, nullptr, 0, 0);
        printf("  alone                         %7.3f ms\n", chain_ms(chain, 100, wgs, link_ticks, sink, nullptr, 0, 0));
        printf("  beside an unmasked filler     %7.3f ms\n", chain_ms(chain, 100, wgs, link_ticks, sink, plain, filler_wgs, filler_ticks));
        for (size_t i = 0; i < masks.size(); ++i)
            printf("  beside a filler on %-22s %7.3f ms\n", masks[i].name,
                   chain_ms(chain, 100, wgs, link_ticks, sink, masked[i], filler_wgs, filler_ticks));
    }
    CHECK(hipFree(sink));
    return 0;
}
