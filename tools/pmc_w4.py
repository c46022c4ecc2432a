"""Three W4A16 launches per Linear shape of a Llama-2-7B block at the bench's M (32768 tokens), for
rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE); nothing else of this library touches the GPU."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gptq_v2_amd as amd
dev = torch.device("cuda:0")
m = 32768
for r, c in ((4096, 4096), (11008, 4096), (4096, 11008)):
    q = torch.randint(0, 16, (r, c), device=dev, dtype=torch.uint8)
    pk = amd.ops.pack_rows(q, 4, amd.ops.PACK_W4_KBLOCK)
    scale = (torch.rand(c // 128, r, device=dev) * 0.01 + 0.002).half()
    zero = torch.randint(0, 16, (c // 128, r), device=dev, dtype=torch.uint8)
    x = torch.randn(m, c, device=dev).half()
    for _ in range(3):
        amd.ops.w4a16_gemm(x, pk, scale, zero, 128)
    torch.cuda.synchronize()
