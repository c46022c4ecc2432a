import torch
dev = torch.device("cuda:0")
torch.backends.cuda.matmul.allow_tf32 = False
for m, n, k in [(4096, 4096, 4096), (5504, 5504, 5504), (4096, 10496, 512)]:
    a = torch.randn(m, k, device=dev); b = torch.randn(n, k, device=dev)
    for _ in range(2): c = a @ b.t()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): c = a @ b.t()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 5
    print("torch fp32", m, n, k, round(ms, 4), "ms", round(2.0 * m * n * k / ms / 1e9, 1), "TFLOP/s")
