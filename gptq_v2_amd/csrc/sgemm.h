// fp32 GEMM on the f32-input MFMA (v_mfma_f32_32x32x2_f32), used by the Cholesky-inverse
// recursion and by the trailing update of the quantisation loop.
#pragma once

#include "gq_common.h"

// Operand storage: which index is contiguous in memory.
//   A_MK: A[m * lda + k]     A_KM: A[k * lda + m]
//   B_NK: B[n * ldb + k]     B_KN: B[k * ldb + n]
enum GqSgemmLayout { GQ_A_MK = 0, GQ_A_KM = 1, GQ_B_NK = 0, GQ_B_KN = 2 };

// Which part of the K range a 128x128 output tile (tile row tm, tile column tn) needs; the rest
// is known to be zero because one operand is triangular.
enum GqSgemmTrim {
    GQ_TRIM_NONE = 0,
    GQ_TRIM_K_LE_N = 1,  // k <  (tn + 1) * 128   (B[k][n] == 0 for k > n)
    GQ_TRIM_K_GE_N = 2,  // k >= tn * 128         (B[k][n] == 0 for k < n)
    GQ_TRIM_K_LE_M = 3,  // k <  (tm + 1) * 128   (A[m][k] == 0 for k > m)
};

struct GqSgemmArgs {
    const float *a;
    const float *b;
    float *c;
    int64_t lda, ldb, ldc;
    int m, n, k;
    int layout;       // GQ_A_* | GQ_B_*
    int trim;         // GqSgemmTrim
    int negate;       // C = beta*C - A*B instead of + A*B
    int accumulate;   // beta = 1 (C is read and is the start of every element's fma chain) else 0
    int lower_only;   // only tiles with tm >= tn are computed (symmetric rank-k update)
    // batch of independent products of the same shape: product z works on a + z * stride_a, ... (floats);
    // batch == 0 means 1
    int batch;
    int64_t stride_a, stride_b, stride_c;
};

// Every element of C is produced by ONE k-ascending chain  c = fma(+-a[m][k], b[k][n], c)
// starting from the old C (accumulate) or 0, which makes the result independent of tiling.
// Requirements: m, n, k, lda, ldb, ldc multiples of 4; pointers 16-byte aligned.
int gq_sgemm(const GqSgemmArgs &args, hipStream_t stream);

// Experiments and tests: 1 = every launch takes the compiler-scheduled kernels (also GQ_SGEMM=plain in the
// environment); results are bit-identical either way.
void gq_sgemm_set_plain(int on);

// c[n][n] (fp32, leading dimension ldc, tiles on and below the diagonal) -= x w^T for bf16 x, w [n][k] with k-contiguous
// rows: the replay GEMM's kernel with an fp32 read-modify-write epilogue (csrc/gemm_nt.hip), used by the factorisation's
// far updates on operands split into bf16 pieces (csrc/hessian.hip).
int gq_gemm_nt_sub_lower_f32(float *c, int64_t ldc, const void *x, const void *w, int64_t n, int64_t k, hipStream_t stream);
