"""Host enqueue time vs GPU time of invert and of one quantize_layer call (experiment helper)."""
import os, sys, json, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gptq_v2_amd as amd
from gptq_v2_amd import ops
dev = torch.device("cuda:0")
res = {}
for c, r in ((4096, 4096), (11008, 4096)):
    x = (torch.randn(8192, c, device=dev) * (1 + 3 * torch.arange(c, device=dev) / c)).half()
    w = (torch.randn(r, c, device=dev) * 0.02).half()
    gs = [128] * (c // 128); gb = [4] * (c // 128)
    hook = amd.HessianHook(); hook.add_batch(x.view(4, 2048, c)); hook.invert()
    perm32 = hook.perm.to(torch.int32).contiguous()
    for rows in (r, r // 4):
        for it in range(3):
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter(); s.record()
            out = ops.quantize_layer(w[:rows], perm32, hook.hessian_inv, gs, gb, sym=False, semantics=ops.SEM_KERNEL,
                                     want_quant=True, want_error=True)
            e.record(); t1 = time.perf_counter()
            torch.cuda.synchronize()
        res[f"C{c}_R{rows}"] = {"host_enqueue_ms": round((t1 - t0) * 1e3, 2), "gpu_ms": round(s.elapsed_time(e), 2)}
print(json.dumps(res))
