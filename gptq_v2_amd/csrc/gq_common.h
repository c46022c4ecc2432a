// Shared host/device helpers for libgptq_amd.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "gptq_amd.h"

// ---- error reporting -----------------------------------------------------------------
void gq_set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));

#define GQ_REQUIRE(cond, ...)                                                                  \
    do {                                                                                       \
        if (!(cond)) {                                                                         \
            gq_set_error(__VA_ARGS__);                                                         \
            return GQ_EINVAL;                                                                  \
        }                                                                                      \
    } while (0)

#define GQ_HIP(expr)                                                                           \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            gq_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,      \
                         __LINE__);                                                            \
            return GQ_EHIP;                                                                    \
        }                                                                                      \
    } while (0)

#define GQ_LAUNCH_CHECK() GQ_HIP(hipGetLastError())

// ---- vector types ----------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define GQ_LDS_PTR(T, p) ((T __attribute__((address_space(3))) *)(p))
#define GQ_GLOBAL_PTR(p) ((const void __attribute__((address_space(1))) *)(p))

static inline int64_t gq_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t gq_round_up(int64_t a, int64_t b) { return gq_cdiv(a, b) * b; }

__device__ __forceinline__ float gq_bf16_to_f32(unsigned short b)
{
    return __uint_as_float(((unsigned int)b) << 16);
}
