"""Five XtX launches at the bench shape (C = 4096, 32768 tokens) and five at C = 11008, for
rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE); nothing else touches the GPU."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gptq_v2_amd as amd
dev = torch.device("cuda:0")
for c in (4096, 11008):
    x = torch.randn(32768, c, device=dev).half()
    h = torch.zeros(c, c, device=dev)
    for _ in range(5):
        amd.ops.xtx_accumulate(h, x)
    torch.cuda.synchronize()
