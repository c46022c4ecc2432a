"""Micro-benchmark of the XtX kernel (experiment helper, not part of the product)."""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gptq_v2_amd as amd
dev = torch.device("cuda:0")
res = {}
for c in (4096, 11008):
    n = 32768
    x = (torch.randn(n, c, device=dev) ).half()
    h = torch.zeros(c, c, device=dev)
    for _ in range(3): amd.ops.xtx_accumulate(h, x)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): amd.ops.xtx_accumulate(h, x)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    res[c] = (round(ms, 3), round(2.0 * n * c * c / ms / 1e9, 1))
print(os.environ.get("GQ_XTX_VARIANT", "default"), json.dumps(res))
