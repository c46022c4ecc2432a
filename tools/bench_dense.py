"""Dense fp16 library GEMM (torch.matmul -> hipBLASLt) at the replay shapes, for calibration of
what the machine delivers on the same problem sizes (experiment helper, not part of the product)."""
import json
import torch
dev = torch.device("cuda:0")
res = {}
for (m, r, c) in ((32768, 4096, 4096), (32768, 11008, 4096), (32768, 4096, 11008), (2048, 4096, 4096)):
    x = torch.randn(m, c, device=dev).half()
    w = (torch.randn(r, c, device=dev) * 0.02).half()
    for _ in range(3): y = x @ w.t()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): y = x @ w.t()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    res[f"{m}x{r}x{c}"] = (round(ms, 3), round(2.0 * m * r * c / ms / 1e9, 1))
# fp32 GEMM (invert / trailing-update class)
for n in (1024, 4096, 5504):
    a = torch.randn(n, n, device=dev); b = torch.randn(n, n, device=dev)
    for _ in range(3): y = a @ b
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): y = a @ b
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    res[f"f32_{n}"] = (round(ms, 3), round(2.0 * n ** 3 / ms / 1e9, 1))
print(json.dumps(res))
