/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement (oracle) of the reference's native
 * quantize path.  Nothing in the product package may link, load or call this file;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * What is restated (all paths relative to the reference checkout):
 *   package/gptq/gptq_cpp.cpp:33-84          quantize_range host loop
 *   package/gptq/gptq_cpp_kernel.cu:21-74    fused_quantize_adjust_kernel
 *   package/gptq/gptq_cpp_kernel.cu:113-137  vv_mul_sub_kernel
 *
 * Arithmetic conventions that the CUDA sources leave to the compiler / cuBLAS and
 * that this file pins explicitly (the reference has no golden vectors for them):
 *   - `vz -= vy * vx` (gptq_cpp_kernel.cu:131-134) is contracted by nvcc's default
 *     -fmad=true into one fused multiply-add, so it is written as fmaf(-vy, vx, vz).
 *   - the trailing `addmm_(x, y, beta=1, alpha=-1)` (gptq_cpp.cpp:81-83) is a cuBLAS
 *     SGEMM whose summation order is not specified.  Here it is DEFINED as the
 *     k-ascending chain  w = fmaf(-U[k][c], E[k][r], w), k = a..b-1, i.e. the same
 *     per-element operation order as the in-block rank-1 updates.  Parity of the
 *     trailing update against the real cuBLAS result is therefore unpinned beyond
 *     fp32 rounding.
 *   - division and rintf are IEEE (nvcc default -prec-div=true, round-half-even).
 *   - __float2half_rn/__hsub/__hmul are emulated in software (round-to-nearest-even,
 *     no intermediate double rounding).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

/* ---- IEEE binary16 emulation ------------------------------------------------ */

static float h2f(uint16_t h)
{
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1fu;
    uint32_t man = h & 0x3ffu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else { /* subnormal: value = man * 2^-24 */
            float v = (float)man * 5.9604644775390625e-08f;
            memcpy(&bits, &v, 4);
            bits |= sign;
        }
    } else if (exp == 31) {
        bits = sign | 0x7f800000u | (man << 13);
    } else {
        bits = sign | ((exp + 112u) << 23) | (man << 13);
    }
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

/* double -> binary16, round-to-nearest-even.  A double holds the exact sum or
 * product of two binary16 values, so one call is one correctly rounded fp16 op. */
static uint16_t d2h(double d)
{
    uint64_t bits;
    memcpy(&bits, &d, 8);
    uint16_t sign = (uint16_t)((bits >> 48) & 0x8000u);
    double a = fabs(d);
    if (isnan(d)) return (uint16_t)(sign | 0x7e00u);
    if (a >= 65520.0) return (uint16_t)(sign | 0x7c00u); /* rounds to inf */
    if (a < 6.103515625e-05) {
        /* subnormal or zero: quantum 2^-24; nearbyint is round-half-even */
        double q = nearbyint(a * 16777216.0);
        return (uint16_t)(sign | (uint16_t)q); /* q == 1024 lands on the min normal */
    }
    int e;
    double m = frexp(a, &e); /* a = m * 2^e, m in [0.5, 1) */
    double q = nearbyint(m * 2048.0); /* 11-bit significand, in [1024, 2048] */
    if (q >= 2048.0) { q = 1024.0; e += 1; }
    int be = e - 1 + 15; /* biased exponent of 1.xxx form */
    if (be >= 31) return (uint16_t)(sign | 0x7c00u);
    return (uint16_t)(sign | (uint16_t)(be << 10) | ((uint16_t)q & 0x3ffu));
}

static uint16_t f2h(float f) { return d2h((double)f); }
static uint16_t hsub(uint16_t a, uint16_t b) { return d2h((double)h2f(a) - (double)h2f(b)); }
static uint16_t hmul(uint16_t a, uint16_t b) { return d2h((double)h2f(a) * (double)h2f(b)); }

/* exported so the tests can pin the emulation against numpy.float16 */
uint16_t gqref_f2h(float f) { return f2h(f); }
float gqref_h2f(uint16_t h) { return h2f(h); }
uint16_t gqref_hsub(uint16_t a, uint16_t b) { return hsub(a, b); }
uint16_t gqref_hmul(uint16_t a, uint16_t b) { return hmul(a, b); }

/* ---- gptq_cpp_kernel.cu:21-74 ------------------------------------------------ */

static float clampf(float x, float lo, float hi) { return fmaxf(lo, fminf(hi, x)); }

static void fused_quantize_adjust(const float *weights, float *quant, const float *scale,
                                  uint16_t *out_q, const float *hessian_inv, float *error,
                                  int row, int rows, int columns, const float *qzero,
                                  float maxq)
{
    const float d = hessian_inv[(uint64_t)row * (uint64_t)rows + (uint64_t)row];
    for (int column = 0; column < columns; column++) {
        uint64_t idx = (uint64_t)row * (uint64_t)columns + (uint64_t)column;
        float x = weights[idx];
        float s = scale[column];
        float z = qzero[column];
        x = x / s;
        x = rintf(x);
        x = x + z;
        x = clampf(x, 0.0f, maxq);
        if (out_q) out_q[idx] = (uint16_t)x;
        uint16_t h_s = f2h(s);
        uint16_t h_x = f2h(x);
        uint16_t h_z = f2h(z);
        h_x = hsub(h_x, h_z);
        h_x = hmul(h_x, h_s);
        float q = h2f(h_x);
        quant[idx] = q;
        float w = weights[idx];
        error[idx] = (w - q) / d;
    }
}

/* ---- gptq_cpp_kernel.cu:113-137: z[i][j] -= y[j] * x[i] (contracted) --------- */

static void vv_mul_sub(const float *x, const float *y, float *z, int x_size, int y_size)
{
    for (int i = 0; i < x_size; i++) {
        const float vx = x[i];
        float *zr = z + (uint64_t)y_size * (uint64_t)i;
        for (int j = 0; j < y_size; j++) zr[j] = fmaf(-y[j], vx, zr[j]);
    }
}

/* ---- gptq_cpp.cpp:33-84 ------------------------------------------------------
 * quant, weights, error: [rows=C][columns=R] fp32; out_q: same shape uint16 or NULL;
 * scale, qzero: [R]; hessian_inv: [C][C] (upper factor U, H^-1 = U^T U).
 * Returns 0, or -1 for the reference's "H shape mismatch" TORCH_CHECK.            */
int gqref_quantize_range(float *quant, const float *scale, uint16_t *out_q, const float *qzero,
                         float maxq, const float *hessian_inv, int hcolumns, float *weights,
                         float *error, int rows, int columns, int a, int b)
{
    if (hcolumns != rows) return -1;
    for (int c = a; c < b; c++) {
        fused_quantize_adjust(weights, quant, scale, out_q, hessian_inv, error, c, hcolumns,
                              columns, qzero, maxq);
        if (c + 1 < b) {
            vv_mul_sub(hessian_inv + (uint64_t)c * (uint64_t)hcolumns + (uint64_t)(c + 1),
                       error + (uint64_t)c * (uint64_t)columns,
                       weights + (uint64_t)(c + 1) * (uint64_t)columns, b - (c + 1), columns);
        }
    }
    /* weights[b:] += -1 * hessian_inv[a:b, b:]^T @ error[a:b]; k-ascending fmaf chain */
    for (int i = b; i < rows; i++) {
        float *wr = weights + (uint64_t)i * (uint64_t)columns;
        for (int k = a; k < b; k++) {
            const float u = hessian_inv[(uint64_t)k * (uint64_t)hcolumns + (uint64_t)i];
            const float *er = error + (uint64_t)k * (uint64_t)columns;
            for (int j = 0; j < columns; j++) wr[j] = fmaf(-u, er[j], wr[j]);
        }
    }
    return 0;
}

/* ---- H += X^T X with fp32 accumulation, for the accumulate op's tolerance tests.
 * accumulate_hessian_kernel.cu:10-64 (CUTLASS, order unspecified) and
 * gptq_triton/accumulate_hessian.py:55-131 (tl.dot, order unspecified) only fix
 * "fp32 accumulate of exact fp16/bf16 products"; this version accumulates each
 * element in double and rounds once, which is the value every fp32 ordering
 * approximates.  x_bits: raw fp16 (dtype 0) or bf16 (dtype 1) words, [n][c].      */
static float bf2f(uint16_t b)
{
    uint32_t bits = (uint32_t)b << 16;
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

int gqref_xtx_accumulate(float *hessian, const uint16_t *x_bits, int dtype, int64_t n, int64_t c)
{
    if (dtype != 0 && dtype != 1) return -1;
    for (int64_t i = 0; i < c; i++) {
        for (int64_t j = 0; j <= i; j++) {
            double acc = 0.0;
            for (int64_t t = 0; t < n; t++) {
                float xi = dtype ? bf2f(x_bits[t * c + i]) : h2f(x_bits[t * c + i]);
                float xj = dtype ? bf2f(x_bits[t * c + j]) : h2f(x_bits[t * c + j]);
                acc += (double)xi * (double)xj;
            }
            hessian[i * c + j] = (float)((double)hessian[i * c + j] + acc);
            if (j != i) hessian[j * c + i] = (float)((double)hessian[j * c + i] + acc);
        }
    }
    return 0;
}
