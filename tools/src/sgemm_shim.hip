// Experiment shim: exports the library-internal fp32 GEMM for tools/bench_sgemm.py.
#include "sgemm.h"
extern "C" void gq_debug_sgemm_plain(int on) { gq_sgemm_set_plain(on); }
// pad > 0: leading dimensions are pad floats longer than the rows (operands are views into wider matrices)
extern "C" int gq_debug_sgemm_ld(const float *a, const float *b, float *c, int m, int n, int k, int layout, int trim,
                                 int negate, int accumulate, int lower_only, void *stream, int pad, int batch)
{
    GqSgemmArgs g{};
    g.a = a; g.b = b; g.c = c; g.m = m; g.n = n; g.k = k;
    g.lda = ((layout & GQ_A_KM) ? m : k) + pad;
    g.ldb = ((layout & GQ_B_KN) ? n : k) + pad;
    g.ldc = n + pad;
    g.layout = layout; g.trim = trim; g.negate = negate; g.accumulate = accumulate; g.lower_only = lower_only;
    g.batch = batch;
    g.stride_a = g.lda * ((layout & GQ_A_KM) ? k : m);
    g.stride_b = g.ldb * ((layout & GQ_B_KN) ? k : n);
    g.stride_c = g.ldc * (int64_t)m;
    return gq_sgemm(g, (hipStream_t)stream);
}
extern "C" int gq_debug_sgemm(const float *a, const float *b, float *c, int m, int n, int k, int layout, int trim,
                              int negate, int accumulate, int lower_only, void *stream)
{
    GqSgemmArgs g{};
    g.a = a; g.b = b; g.c = c; g.m = m; g.n = n; g.k = k;
    g.lda = (layout & GQ_A_KM) ? m : k;
    g.ldb = (layout & GQ_B_KN) ? n : k;
    g.ldc = n;
    g.layout = layout; g.trim = trim; g.negate = negate; g.accumulate = accumulate; g.lower_only = lower_only;
    return gq_sgemm(g, (hipStream_t)stream);
}
