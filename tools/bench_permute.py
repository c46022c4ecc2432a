"""Activation column gather (act-order) at the two Llama-2-7B widths (experiment helper)."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gptq_v2_amd as amd
dev = torch.device("cuda:0")
res = {}
for c in (4096, 11008):
    x = torch.randn(32768, c, device=dev).half()
    perm = torch.randperm(c, device=dev).to(torch.int32)
    for _ in range(2): amd.ops.permute_columns(x, perm)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): y = amd.ops.permute_columns(x, perm)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    assert torch.equal(y, x[:, perm.long()])
    res[c] = {"ms": round(ms, 4), "TB/s": round(4.0 * x.numel() / ms / 1e9, 2)}
print(json.dumps(res))
