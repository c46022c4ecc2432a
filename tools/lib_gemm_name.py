"""Runs the library fp16 GEMM once per replay shape so that a kernel trace shows which kernel
configuration the library picks (experiment helper)."""
import torch
dev = torch.device("cuda:0")
for (m, r, c) in ((32768, 4096, 4096), (32768, 11008, 4096), (32768, 4096, 11008)):
    x = torch.randn(m, c, device=dev).half()
    w = (torch.randn(r, c, device=dev) * 0.02).half()
    for _ in range(3):
        y = x @ w.t()
    torch.cuda.synchronize()
