"""The fp32-accumulating lower-triangular bf16 product behind the inversion's far updates (gq_gemm_nt_sub_lower) against the
plain replay GEMM on operands of the same size: time per launch and bf16 TFLOP/s (experiment helper)."""
import ctypes, os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gptq_v2_amd as amd
lib = amd._lib.load()
dev = torch.device("cuda:0")


def timed(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


for n, k in [(int(a.split("x")[0]), int(a.split("x")[1])) for a in sys.argv[1:]] or ((20480, 3072), (10240, 3072), (20480, 6144)):
    x = torch.randn(n, k, device=dev).bfloat16()
    w = torch.randn(n, k, device=dev).bfloat16()
    c = torch.zeros(n, n, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    ms = timed(lambda: lib.gq_gemm_nt_sub_lower(p(c), n, p(x), p(w), n, k, st))
    t = n // 256
    tiles = t * (t + 1) // 2
    flops = tiles * 2.0 * 256 * 256 * k
    ms_full = timed(lambda: amd.ops.gemm_nt(x, w))
    print(json.dumps({"n": n, "k": k, "lower_tiles": tiles, "sub_lower_ms": round(ms, 3), "sub_lower_tflops": round(flops / ms / 1e9, 1),
                      "plain_full_ms": round(ms_full, 3), "plain_tflops": round(2.0 * n * n * k / ms_full / 1e9, 1)}), flush=True)
    del x, w, c
