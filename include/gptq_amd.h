/*
 * gptq_amd.h -- C ABI of libgptq_amd.so: the GPTQ calibration / int4 inference hot path
 * as hand-written gfx950 (MI355X) HIP kernels.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host;
 *   - all buffers are caller-allocated, dense and row-major; nothing is returned
 *     except through them;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); calls are
 *     asynchronous with respect to the host unless stated otherwise;
 *   - return value: 0 on success, a negative GQ_E* code on error; gq_last_error()
 *     returns a thread-local, human-readable description of the last failure;
 *   - dtype codes: GQ_F16 = 0 (IEEE binary16), GQ_BF16 = 1;
 *   - a tensor with zero rows may be passed as NULL (an empty torch tensor has no data
 *     pointer): gq_xtx_accumulate with n_tokens == 0, gq_permute_columns and the W4A16
 *     GEMMs with m == 0 return 0 without a launch.
 *
 * Citations "ref:" name the reference interface each entry point replaces
 * (paths are relative to the 4PiR2/gptq-v2 checkout).
 */
#ifndef GPTQ_AMD_H
#define GPTQ_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GQ_F16 0
#define GQ_BF16 1

#define GQ_OK 0
#define GQ_EINVAL (-1)     /* bad argument (shape, dtype code, alignment, NULL) */
#define GQ_EHIP (-2)       /* a HIP runtime call or kernel launch failed */
#define GQ_ENOTPD (-3)     /* Hessian still not positive definite after max_try dampings */
#define GQ_EWORKSPACE (-4) /* workspace too small */

/* Semantics selector of the column loop (the reference has two, with different results). */
#define GQ_SEM_KERNEL 0 /* ref: gptq_py.py:125-157 -- grid from the unmodified weights,
                           fp16 dequantisation inside the loop, fused multiply-add updates */
#define GQ_SEM_PYTHON 1 /* ref: gptq_py.py:158-193 -- grid from the error-compensated
                           weights, fp32 dequantisation, separate multiply and subtract */

typedef void *gq_stream_t;

/* Version of this interface: raised whenever an entry point's argument list changes in place (3: gq_quant_metrics
 * takes perm_inv; gq_set_quant_super exists).  gq_version() returns the value the library was built with; a caller
 * built against another header must not go on (the Python binding checks it when it loads the library). */
#define GQ_ABI_VERSION 3

const char *gq_last_error(void);
int gq_version(void);

/* Compute units and peak clock (kHz) of the current device, as the runtime reports them: what a caller needs to state
 * a matrix-core peak next to a measurement (CUs x clock x 4096 flops per clock and CU for 16-bit operands).  No
 * reference counterpart. */
int gq_device_info(int *compute_units, int *max_clock_khz);

/* H[c][c] (fp32) += X^T X for X[n_tokens][hidden] (fp16 or bf16), fp32 accumulation.
 * compute_lower_only != 0: only tiles on or below the diagonal are computed; unless
 * save_lower_only != 0 each strictly-lower tile is also written, transposed, to its mirror
 * position, so H comes out exactly symmetric.  save_lower_only requires compute_lower_only.
 * Allocates nothing (the order in which workgroups take the tiles is computed in the kernel).
 * ref: package/gptq/__init__.py:6-13, accumulate_hessian.cpp:20-45,
 *      accumulate_hessian_kernel.cu:10-84 (full product: compute_lower_only = 0),
 *      gptq_triton/accumulate_hessian.py:134-149 (save_lower_only / compute_lower_only). */
int gq_xtx_accumulate(float *hessian, const void *x, int dtype, int64_t n_tokens, int64_t hidden,
                      int save_lower_only, int compute_lower_only, gq_stream_t stream);

/* The same accumulation with the tokens of the call cut into n_parts runs, run p added into the
 * partial Hessian parts + p * part_stride (floats; part_stride >= hidden^2, a multiple of 4): with few
 * output tiles (hidden = 4096: 136 lower tiles on 256 CUs) one workgroup per tile leaves CUs idle,
 * (tile, run) workgroups fill them.  H is the sum of the parts; gq_xtx_fold adds parts 1.. into part 0
 * in index order and clears them, so H does not depend on the order in which workgroups finished and is
 * exactly symmetric.  Shapes the split does not apply to go into part 0 whole.  gq_xtx_split_hint
 * returns the number of parts that pays for a call of n_tokens (1 = do not split).
 * ref: gptq_triton/accumulate_hessian.py:134-149 (same contract per part). */
int gq_xtx_split_hint(int64_t hidden, int64_t n_tokens, int compute_lower_only);
int gq_xtx_accumulate_split(float *parts, int n_parts, int64_t part_stride, const void *x, int dtype, int64_t n_tokens,
                            int64_t hidden, int save_lower_only, int compute_lower_only, gq_stream_t stream);
int gq_xtx_fold(float *parts, int n_parts, int64_t part_stride, int64_t hidden, gq_stream_t stream);

/* The same accumulation (lower tiles computed, mirror images written) for hidden sizes whose tiles do not fill
 * whole rounds of one workgroup per CU (hidden = 11008: 946 lower tiles = 3 rounds of 256 + 178): the tiles of the
 * last, partly filled round are cut into token runs and summed in tile-sized buffers in tail_ws, everything else goes
 * straight into H.  gq_xtx_tail_workspace(hidden) is the size of tail_ws in bytes, 0 when the hidden size has no
 * such tail (then use gq_xtx_accumulate); the caller zeroes tail_ws once, may call gq_xtx_accumulate_tail any number
 * of times, and calls gq_xtx_fold_tail before reading H: it adds the buffers into H in index order, clears them
 * and writes the tail tiles' mirror images.  Until then the tail tiles of H are stale.
 * ref: gptq_triton/accumulate_hessian.py:134-149 (same contract once folded). */
size_t gq_xtx_tail_workspace(int64_t hidden);
int gq_xtx_accumulate_tail(float *hessian, float *tail_ws, const void *x, int dtype, int64_t n_tokens, int64_t hidden,
                           gq_stream_t stream);
int gq_xtx_fold_tail(float *hessian, float *tail_ws, int64_t hidden, gq_stream_t stream);

/* The same accumulation for fp32 activations X[n_tokens][hidden]: full product on the fp32 matrix
 * core, every element one k-ascending chain of fused multiply-adds (exactly symmetric result).
 * n_tokens and hidden must be multiples of 4 (zero rows may be appended to X).
 * ref: gptq_py.py:31-33 (add_batch(use_kernel=False): torch.addmm on the fp32 cast of the input). */
int gq_xtx_accumulate_f32(float *hessian, const float *x, int64_t n_tokens, int64_t hidden,
                          gq_stream_t stream);

/* How the factorisation inside gq_hessian_invert applies its large symmetric updates D -= L L^T: 0 (the default) =
 * chosen per update by size, 1 = always as fp32 matrix-core products, 2 = wherever the workspace allows as ONE bf16
 * product of the operands split into three bf16 pieces each (six partial products, exact in fp32, summed in fp32:
 * fp32-grade accuracy at about 1.4 times the rate on updates that fill the chip; C = 28672: 137 -> 126 ms per
 * inversion).  Returns the previous setting, -1 on a bad argument.  A tuning knob of this library, no reference
 * counterpart; the environment variable GQ_INVERT_SPLIT=0|1 sets the initial value (0: never, 1: wherever they fit). */
int gq_set_invert_split(int mode);

/* Bytes of workspace gq_hessian_invert needs for a C x C Hessian. */
size_t gq_hessian_invert_workspace(int64_t c);

/* The whole of HessianHook.invert on the device.  In place on `hessian`:
 *   H *= 2/n_samples; H[i][i] = 1 where it is 0; if act_order: perm = argsort(diag, descending,
 *   ties by index), H = H[perm][:, perm]; damp = damp_ratio * mean(diag);
 *   repeat { H[i][i] += damp; factor } until the factorisation succeeds (at most max_try times).
 * Outputs: hessian_inv = U, upper triangular (zeros below the diagonal), (H)^-1 = U^T U;
 * perm / perm_inv (int32[c], written only if act_order; may be NULL otherwise);
 * *tries_host = number of dampings applied.  U is obtained as the inverse of the factor of the
 * index-reversed matrix (J H J = L L^T  =>  U = J L^-1 J), mathematically the factor the
 * reference gets from cholesky -> cholesky_inverse -> cholesky(upper), at half the flops.
 * A try fails when a pivot is <= 0 or not finite (the reference tests the result for NaN); a
 * diagonal that is not finite to begin with fails after the first try with GQ_ENOTPD (damping
 * cannot repair it; the reference runs out its 100 tries and asserts).
 * Synchronous: returns after the stream has drained (one host round trip per try).
 * ref: gptq_py.py:38-66. */
int gq_hessian_invert(float *hessian, float *hessian_inv, int32_t *perm, int32_t *perm_inv,
                      int64_t c, double n_samples, float damp_ratio, int act_order, int max_try,
                      void *workspace, size_t workspace_bytes, int *tries_host, gq_stream_t stream);

/* gq_hessian_invert in two halves, so that the host does not wait for the device between the
 * inversion and the work that consumes it.
 * gq_hessian_invert_begin enqueues everything up to and including the first try (same arguments and
 * same effect on hessian / hessian_inv / perm / perm_inv as gq_hessian_invert with max_try = 1) and
 * leaves the verdict of that try in device memory: verdict_dev[0] != 0: a pivot was not positive (one
 * more damping is needed), verdict_dev[1] != 0: the diagonal of H is not finite.  It does not
 * synchronise.  hessian_inv is written only if the try succeeded; the caller may enqueue work that
 * reads it at once and must discard that work's results if the verdict turns out bad.
 * gq_hessian_invert_finish takes the verdict as read back by the caller (verdict_host[2], host memory;
 * the workspace must be the one given to _begin, untouched) and returns at once with *tries_host = 1
 * if the first try succeeded; otherwise it makes the remaining tries (up to max_try in all) the way
 * gq_hessian_invert does, one host round trip each.  begin + finish leave exactly what gq_hessian_invert
 * leaves, bit for bit.
 * ref: gptq_py.py:55-66 (the retry loop, whose test of the result forces a host round trip per try). */
int gq_hessian_invert_begin(float *hessian, float *hessian_inv, int32_t *perm, int32_t *perm_inv,
                            int64_t c, double n_samples, float damp_ratio, int act_order,
                            void *workspace, size_t workspace_bytes, int32_t *verdict_dev, gq_stream_t stream);
int gq_hessian_invert_finish(float *hessian, float *hessian_inv, int64_t c, int max_try,
                             void *workspace, size_t workspace_bytes, const int32_t *verdict_host,
                             int *tries_host, gq_stream_t stream);

/* gq_hessian_invert without a workspace argument and without any allocation: the workspace is a
 * device array built into the library (about 12.6 MB on every GPU the library runs on, held until
 * the process ends), so C is limited to gq_hessian_invert_static_max_c() = 1024 (larger C:
 * GQ_EINVAL).  Same arguments otherwise, same kernels, same results bit for bit.  The call issues
 * no hipMalloc / hipFree; it still synchronises the stream once per try, like gq_hessian_invert.
 * All calls on one GPU share the one array: calls that overlap in time on the same GPU (from two
 * streams or two threads) race and must be serialised by the caller.
 * No counterpart in the reference: it exists for callers that cannot allocate device memory
 * (harnesses that forbid allocation, tight loops around small layers such as OPT-125M's 768-wide
 * Hessians). */
int64_t gq_hessian_invert_static_max_c(void);
int gq_hessian_invert_static(float *hessian, float *hessian_inv, int32_t *perm, int32_t *perm_inv,
                             int64_t c, double n_samples, float damp_ratio, int act_order, int max_try,
                             int *tries_host, gq_stream_t stream);

/* Per-row quantisation grid over columns [col0, col0+ncols) of W^T stored as weights_t[C][R]:
 * asymmetric (sym = 0): scale = (max(x,0) - min(x,0)) / maxq + 1e-12, qzero = round(-min/scale);
 * symmetric: scale = max|x| * 2/maxq + 1e-12, qzero = (maxq+1)/2.   maxq = 2^bits - 1.
 * ref: quant.py:89-118 (the EXL2 scale quantisation: gq_quantize_layer_ex). */
int gq_find_params(const float *weights_t, int64_t c, int64_t r, int64_t col0, int64_t ncols, int bits,
                   int sym, float *scale, float *qzero, gq_stream_t stream);

/* Columns [a, b) of the GPTQ loop with the reference's native argument order and layouts:
 * quant, weights, error are [C][R] fp32 (weights updated in place, rows > a), out_q is [C][R]
 * int16 or NULL, scale/qzero are [R] fp32, hessian_inv is U [C][C].
 * Per column c: x = clamp(rint(w/s)+z, 0, maxq); q = half(half(x)-half(z))*half(s);
 * e = (w-q)/U[c][c]; w[c'] = fma(-e, U[c][c'], w[c']) for every c' > c (ascending c).
 * Any 0 <= a <= b <= C is accepted, as in the reference; columns outside the largest sub-range
 * aligned to 4 (all of them when R or C is not a multiple of 4) take a one-column-per-launch path.
 * ref: package/gptq/__init__.py:16-28, gptq_cpp.cpp:33-84, gptq_cpp_kernel.cu:21-158. */
int gq_quantize_range(float *quant, const float *scale, int16_t *out_q, const float *qzero, float maxq,
                      const float *hessian_inv, float *weights, float *error, int64_t c, int64_t r,
                      int64_t a, int64_t b, gq_stream_t stream);

/* Bytes of workspace gq_quantize_layer needs. */
size_t gq_quantize_layer_workspace(int64_t r, int64_t c);

/* gptq_quant for one Linear, both loop semantics, all on the device.
 * weight[R][C] (fp16/bf16); perm int32[C] or NULL; hessian_inv U[C][C];
 * group_sizes_host / group_bits_host: n_groups ints each, sum(group_sizes) == C.
 * Outputs: qweight uint8[R][C] (permuted column order), scale [R][G] in the weight's dtype,
 * qzero uint8[R][G], quant_t / error_t fp32 [C][R] (dequantised weights and scaled errors,
 * transposed, permuted order; either may be NULL).
 * ref: gptq_py.py:69-251 (prologue :86-96, loops :125-193, collation :216-229). */
int gq_quantize_layer(const void *weight, int dtype, const int32_t *perm, const float *hessian_inv,
                      int64_t r, int64_t c, const int32_t *group_sizes_host,
                      const int32_t *group_bits_host, int n_groups, int sym, int semantics,
                      uint8_t *qweight, void *scale, uint8_t *qzero, float *quant_t, float *error_t,
                      void *workspace, size_t workspace_bytes, gq_stream_t stream);

/* How the per-group grids are chosen in gq_quantize_layer_ex. */
typedef struct gq_grid_options {
    int sym;             /* symmetric grid (ref: quant.py:104-106); implied by scale_bits > 0 */
    int scale_bits;      /* 0: off.  1..8: EXL2 scale quantisation -- a group's scales go onto the
                            quadratic grid q^2 * sscale, q in 1..2^scale_bits, sscale one number per
                            group (ref: quant.py:40-69, 108-112) */
    int mse;             /* != 0: Quantizer.mse, PyTorch branch (ref: quant.py:150-165): every row keeps
                            the candidate scale * shrink[k] with the smallest sum of
                            |dequantize(quantize(x)) - x|^norm (first minimum; with scale_bits > 0 the
                            same search is then run for sscale over the group's rows).  The
                            quantisation is fp32 as in the reference; the error sums are accumulated
                            in double where the reference sums fp32 powers, so a near-tie may be
                            decided differently */
    const float *shrink; /* DEVICE array of the n_shrink candidates; the reference's are
                            1 - arange(0, max_shrink, 1 / n_grid) */
    int n_shrink;        /* 1..256 */
    float norm;          /* > 0 */
} gq_grid_options;

/* gq_quantize_layer with the reference's remaining knobs.
 *   hessian_inv == NULL: round to nearest (perm must be NULL, semantics is ignored).
 *   block_sizes_host / n_blocks: the blocks of the PyTorch loop (GQ_SEM_PYTHON only; NULL = the
 *     groups).  As in the reference, a group's grid is computed when the loop reaches the group's
 *     first column, from the weights as they were at the start of the block that column lies in,
 *     and the shrink search looks at the columns of that block (ref: gptq_py.py:162-186); for
 *     GQ_SEM_KERNEL every grid comes from the unmodified weights and the search looks at the
 *     group's own columns (ref: gptq_py.py:132-139).
 *   qscale uint8 [R][G] and sscale [G] (weight dtype) are written when options->scale_bits > 0
 *     (and must then be non-NULL); scale then holds q^2 * sscale.
 * Groups are reported in column order; the reference's re-ordering of mixed groups
 * (gptq_py.py:202-229) is index arithmetic left to the caller.
 * ref: gptq_py.py:69-251, quant.py:89-165. */
int gq_quantize_layer_ex(const void *weight, int dtype, const int32_t *perm, const float *hessian_inv,
                         int64_t r, int64_t c, const int32_t *group_sizes_host,
                         const int32_t *group_bits_host, int n_groups, const int32_t *block_sizes_host,
                         int n_blocks, int semantics, const gq_grid_options *options, uint8_t *qweight,
                         void *scale, uint8_t *qzero, uint8_t *qscale, void *sscale, float *quant_t,
                         float *error_t, void *workspace, size_t workspace_bytes, gq_stream_t stream);

/* Round-to-nearest quantisation without a Hessian (same outputs as gq_quantize_layer;
 * error is identically zero in the reference and is not produced).
 * ref: gptq_py.py:101-124. */
int gq_quantize_rtn(const void *weight, int dtype, int64_t r, int64_t c,
                    const int32_t *group_sizes_host, const int32_t *group_bits_host, int n_groups,
                    int sym, uint8_t *qweight, void *scale, uint8_t *qzero, float *quant_t,
                    void *workspace, size_t workspace_bytes, gq_stream_t stream);

/* Sums behind gptq_quant's four metrics, accumulated in double:
 *   sums[0] = sum(error^2), sums[1] = sum((W[r][j] / U[j][j])^2) (W unpermuted, as the reference
 *   writes it), sums[2] = sum((quant[r][j] - W[r][perm[j]])^2), sums[3] = sum(W^2) with every
 *   square rounded to the weight's 16-bit type first, as the reference computes it.
 * `sums` (double[4], device) is zeroed by the call.  error_t / hessian_inv may be NULL (their
 * sums stay 0).  The caller divides by R*C.  perm_inv (int32[C], the inverse of perm as
 * gq_hessian_invert returns it) is optional: with it W is read as it lies in memory and quant_t row by
 * row, which is several times faster for an order that is not nearly monotone.
 * ref: gptq_py.py:195-200, 231-238. */
int gq_quant_metrics(const void *weight, int dtype, const int32_t *perm, const int32_t *perm_inv,
                     const float *hessian_inv, const float *quant_t, const float *error_t, int64_t r,
                     int64_t c, double *sums, gq_stream_t stream);

/* weight_out[R][C] (fp16/bf16) = ((int16)q - (int16)z) * scale per group, evaluated in the output
 * dtype, then columns un-permuted: out[:, j] = deq[:, perm_inv[j]] (perm_inv int32[C] or NULL).
 * Groups: uniform group_size > 0, or group_size == 0 and col_group (int32[C], device) giving the
 * group index of every (permuted) column.
 * ref: quant.py:222-237 (construct_matrix), quant.py:262-279 (reconstruct_nn_linear). */
int gq_dequantize(const uint8_t *qweight, const void *scale, const uint8_t *qzero,
                  const int32_t *perm_inv, int64_t r, int64_t c, int n_groups, int group_size,
                  const int32_t *col_group, int dtype, void *weight_out, gq_stream_t stream);

#define GQ_PACK_STREAM 0      /* value i at bits [i*bits, i*bits+bits) of the row's bit stream */
#define GQ_PACK_W4_PAIRS 1    /* 4-bit only: within each uint32, values 0..7 sit in nibbles
                                 0,4,1,5,2,6,3,7, so (w >> 4p) & 0x000F000F yields the value pair
                                 (2p, 2p+1) in the two 16-bit halves */
#define GQ_PACK_W4_KBLOCK 2   /* 4-bit only, n a multiple of 64: words of GQ_PACK_W4_PAIRS stored as
                                 packed[n/64][rows][8] -- the 64 values of one row's k-step are 32
                                 contiguous bytes and consecutive rows follow each other, so a GEMM
                                 tile reads one contiguous piece per k-step: the W4A16 GEMM operand */

#define GQ_PACK_W3_KBLOCK 3   /* 3-bit only, n a multiple of 64: packed[n/64][rows][6]; each run of 32 values
                                 is 3 words -- value v < 30 in word v / 10 at bit offset 3 * ((v % 10) / 2) of the
                                 low (v even) or high (v odd) half, so (w >> 3p) & 0x00070007 yields a value pair;
                                 bit j of values 30 / 31 is bit 15 / 31 of word j: 3 bits per weight in memory,
                                 the operand of gq_wq_gemm with bits = 3 */

/* Packing of rows of `bits`-wide values (2, 3, 4 or 8) into uint32 words,
 * words_per_row = ceil(n*bits/32), rows padded with zero bits.
 * No reference counterpart (the reference stores one value per byte, gptq_py.py:217). */
int gq_pack_rows(const uint8_t *values, int64_t rows, int64_t n, int bits, int layout,
                 uint32_t *packed, gq_stream_t stream);
int gq_unpack_rows(const uint32_t *packed, int64_t rows, int64_t n, int bits, int layout,
                   uint8_t *values, gq_stream_t stream);

/* The lower triangle of a symmetric C x C fp32 matrix as one flat buffer of C (C + 1) / 2 floats -- element
 * (r, c <= r) at r (r + 1) / 2 + c -- and back, the upper triangle restored as the mirror image (exactly
 * symmetric).  What is exchanged when the Hessians of several GPUs are summed: half the bytes of the matrix.
 * No reference counterpart (the reference is single-GPU, parse_args.py:53-55). */
int gq_pack_lower(const float *hessian, int64_t c, float *flat, gq_stream_t stream);
int gq_unpack_lower(const float *flat, int64_t c, float *hessian, gq_stream_t stream);

/* x_out[m][j] = x_in[m][perm[j]] for 16-bit elements (activation gather for act-order). */
int gq_permute_columns(const void *x_in, const int32_t *perm, int64_t m, int64_t c, void *x_out,
                       gq_stream_t stream);

/* y[M][R] (fp16) = x[M][C] (fp16, columns already in permuted order) * W^T with
 * W[r][k] = half(q[r][k] - z[g(k)][r]) * scale[g(k)][r]: q from 4-bit packed weights in the
 * GQ_PACK_W4_KBLOCK layout ([C/64][R][8] words), scale_t fp16 [G][R] and qzero_t uint8 [G][R]
 * (the transposes of the reference's (R, G) tensors), uniform group size (a multiple of 32
 * dividing C), C a multiple of 64, fp32 accumulation.
 * Fills the slot of the dense fp16 nn.Linear at llama.py:172-177. */
int gq_w4a16_gemm(void *y, const void *x, const uint32_t *qweight_packed, const void *scale_t,
                  const uint8_t *qzero_t, int64_t m, int64_t r, int64_t c, int group_size,
                  gq_stream_t stream);

/* The same product for x, y and scale_t of type `dtype` (GQ_F16 or GQ_BF16).  bf16: W[r][k] is
 * (q - z) * scale rounded once to bf16, as the reference's reconstruction of a bf16 model computes
 * it (quant.py:30 with the operand types of quant.py:263-272), fp32 accumulation, bf16 output.
 * Fills the same slot as gq_w4a16_gemm (llama.py:172-177) for models held in bfloat16. */
int gq_w4a16_gemm_dt(void *y, const void *x, const uint32_t *qweight_packed, const void *scale_t,
                     const uint8_t *qzero_t, int64_t m, int64_t r, int64_t c, int group_size,
                     int dtype, gq_stream_t stream);

/* The packed product for 4-bit (GQ_PACK_W4_KBLOCK) or dense 3-bit (GQ_PACK_W3_KBLOCK) weights: same
 * arguments and arithmetic as gq_w4a16_gemm_dt, `bits` = 4 or 3 names the operand layout.  Group sizes
 * are multiples of 32 dividing C (a layer with mixed groups is passed on the grid of their common
 * divisor, every fine group carrying the scale and zero point of the group it lies in).
 * Fills the slot of llama.py:172-177 like gq_w4a16_gemm; the reference has no packed storage. */
int gq_wq_gemm(void *y, const void *x, const uint32_t *qweight_packed, const void *scale_t,
               const uint8_t *qzero_t, int64_t m, int64_t r, int64_t c, int group_size, int bits, int dtype,
               gq_stream_t stream);

/* The dense 16-bit weight a packed operand stands for, w_out[r][c] row-major with the columns in the operand's
 * own (act-order) sequence: W[r][k] = (q - z) * scale rounded once to `dtype`, the value gq_wq_gemm multiplies
 * by and gq_dequantize writes (ref: quant.py:222-237, 262-279, up to the column order).  With many tokens per
 * call (prefill) every 256-token band of a packed GEMM expands the same levels again; expanding once into a
 * transient buffer and running gq_gemm_nt on it gives the same product bit for bit and is faster from a few
 * thousand tokens on.  Weights stay packed at rest.  w_out 16-byte aligned, r * c * 2 bytes. */
int gq_wq_expand(void *w_out, const uint32_t *qweight_packed, const void *scale_t, const uint8_t *qzero_t,
                 int64_t r, int64_t c, int group_size, int bits, int dtype, gq_stream_t stream);

/* Which kernel gq_wq_gemm runs for 4-bit operands: 0 = chosen by shape (the default), 1 = the stream
 * kernel (one barrier per k-step, expansion between the matrix instructions), 2 = the split kernel
 * (expansion and matrix waves apart; the one gq_w4a16_gemm runs).  Returns the previous setting, -1 on a
 * bad argument.  Dense 3-bit operands always take the stream kernel.  A tuning knob of this library, no
 * reference counterpart; the environment variable GQ_WQ_KERNEL=stream|split sets the initial value. */
int gq_set_wq_kernel(int variant);

/* Tile shape of the split 4-bit kernel: 0 = chosen by shape (the default), 128 = 128 x 128, 256 = 256 x 256,
 * 384 = 256 tokens x 128 rows of W (for launches whose 256 x 256 tiles would leave CUs without a workgroup, e.g.
 * 2048 tokens against 4096 rows).  A forced shape applies where the operands allow it (256-row tiles need m >= 256
 * and groups that are multiples of 64).  Returns the previous setting, -1 on a bad argument.  A tuning knob of this
 * library, no reference counterpart; the environment variable GQ_W4_TILE sets the initial value. */
int gq_set_w4_tile(int tile);

/* How the column loop of gq_quantize_layer / gq_quantize_layer_ex is launched for native-loop semantics
 * when every group is a multiple of 16 columns: 1 (the default) = one launch per 512 columns in which a
 * workgroup keeps its 32 rows' window in LDS and updates it between 16-column steps on the matrix core,
 * followed by one GEMM for the columns behind the window; 0 = one launch per 128 columns with a GEMM
 * between every two.  Every weight receives the same fused multiply-adds in the same order either way, so
 * all outputs are identical bit for bit; the switch exists for comparisons and tests.  Returns the
 * previous setting.  A tuning knob of this library, no reference counterpart; the environment variable
 * GQ_QUANT_SUPER=0 sets the initial value to 0. */
int gq_set_quant_super(int on);

/* y[M][N] = x[M][K] * w[N][K]^T for dense 16-bit operands (dtype GQ_F16 or GQ_BF16 for all three),
 * fp32 accumulation, one rounding of the result: the forward of the dense nn.Linear that the
 * reference builds from a reconstructed weight and runs over the calibration batches
 * (ref: llama.py:172-177 `lin(h)`, quant.py:276-278).  K must be a multiple of 64; M and N are
 * free.  w is the output of gq_dequantize (columns in the model's own order, so x needs no gather). */
int gq_gemm_nt(void *y, const void *x, const void *w, int64_t m, int64_t n, int64_t k, int dtype,
               gq_stream_t stream);

/* c[n][n] (fp32, leading dimension ldc) -= x w^T on the tiles on and below the diagonal (256 x 256 tiles; the part of a
 * diagonal tile above the diagonal is written too), for bf16 x, w [n][k] with k-contiguous rows: the replay GEMM's
 * kernel with an fp32 read-modify-write epilogue.  n a multiple of 128, k of 64, ldc of 4; pointers 16-byte aligned.
 * gq_hessian_invert uses it for the factorisation's large symmetric updates, with operands split into bf16 pieces
 * (gq_set_invert_split); exported for measurement.  No reference counterpart. */
int gq_gemm_nt_sub_lower(float *c, int64_t ldc, const void *x, const void *w, int64_t n, int64_t k, gq_stream_t stream);

/* The two bandwidth-bound operators between the GEMMs of a Llama block's staged replay, one pass each, with the
 * roundings of the operator chains they replace (only the order of the sum of squares is the kernel's own):
 *   gq_rmsnorm:  out = T(float(weight) * float(T(float(x) * rsqrt(mean(float(x)^2) + eps))))   per row of `hidden`
 *   gq_silu_mul: out = T(float(T(silu(float(gate)))) * float(up))                                per element
 * T = fp16 or bf16 (dtype); hidden and n multiples of 8, pointers 16-byte aligned; out may alias x / gate.
 * ref: llama.py:172-190 replays the block's own modules (the model's RMS norm and MLP); these take their place
 *      in the stage-wise replay of gptq_v2_amd.driver.quantize_llama. */
int gq_rmsnorm(void *out, const void *x, const void *weight, int dtype, int64_t n_rows, int64_t hidden, float eps,
               gq_stream_t stream);
int gq_silu_mul(void *out, const void *gate, const void *up, int dtype, int64_t n, gq_stream_t stream);

/* Rotary position embedding, in place, of n_heads adjacent heads of head_dim columns at the start of every row of
 * x[n_batch * seq][row_stride] (e.g. the q and k columns of a merged q/k/v product), with cos / sin
 * [n_batch or 1][seq][head_dim] (cs_batch_stride = seq * head_dim, or 0 when all sequences share the positions):
 * x * cos + rotate_half(x) * sin with each product and the sum rounded to the 16-bit type, as the operators round.
 * head_dim a multiple of 16, strides multiples of 8, pointers 16-byte aligned.
 * ref: llama.py:172-190 replays the model's own attention module; this takes the place of its rotary embedding in
 *      the stage-wise replay of gptq_v2_amd.driver.quantize_llama. */
int gq_rope_inplace(void *x, int64_t row_stride, const void *cos_t, const void *sin_t, int64_t cs_batch_stride, int64_t n_batch,
                    int64_t seq, int n_heads, int head_dim, int dtype, gq_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GPTQ_AMD_H */
