"""Dense replay GEMM over back-to-back launches (10 per shape after 3 warm-up launches) at the replay
shapes of a Llama-2-7B block, M = 32768: the sustained, power-limited rate (experiment helper; set
GQ_LIB=1 to time the library's GEMM on the same operands in the same process)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gptq_v2_amd as amd
dev = torch.device("cuda:0")
tag = os.environ.get("GQ_TAG", "")
for (m, n, k) in ((32768, 12288, 4096), (32768, 4096, 4096), (32768, 11008, 4096), (32768, 4096, 11008)):
    x = (torch.randn(m, k, device=dev) * (1 + 3 * torch.arange(k, device=dev) / k)).half()
    w = (torch.randn(n, k, device=dev) * 0.02).half()
    fns = {"nt": lambda: amd.ops.gemm_nt(x, w)}
    if os.environ.get("GQ_LIB"):
        fns["lib"] = lambda: torch.matmul(x, w.t())
    out = []
    for name, f in fns.items():
        for _ in range(3): f()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): f()
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 10
        out.append("%s %.3f ms %.1f TFLOP/s" % (name, ms, 2.0 * m * n * k / ms / 1e9))
    print(tag, m, n, k, " | ".join(out), flush=True)
