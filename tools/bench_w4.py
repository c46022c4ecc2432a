"""Micro-benchmark of the W4A16 GEMM (experiment helper, not part of the product)."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gptq_v2_amd as amd
dev = torch.device("cuda:0")
res = {}
for (m, r, c) in ((32768, 4096, 4096), (32768, 11008, 4096), (32768, 4096, 11008), (2048, 4096, 4096)):
    q = torch.randint(0, 16, (r, c), device=dev, dtype=torch.uint8)
    pk = amd.ops.pack_rows(q, 4, amd.ops.PACK_W4_KBLOCK)
    scale = (torch.rand(c // 128, r, device=dev) * 0.01 + 0.002).half()
    zero = torch.randint(0, 16, (c // 128, r), device=dev, dtype=torch.uint8)
    x = torch.randn(m, c, device=dev).half()
    for _ in range(2): amd.ops.w4a16_gemm(x, pk, scale, zero, 128)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): amd.ops.w4a16_gemm(x, pk, scale, zero, 128)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 5
    res[f"{m}x{r}x{c}"] = (round(ms, 3), round(2.0 * m * r * c / ms / 1e9, 1))
print(json.dumps(res))
