"""gptq_quant on row counts that are not powers of two (experiment helper)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gptq_v2_amd as amd
dev = torch.device("cuda:0")
c = 4096
x = (torch.randn(4096, c, device=dev) * (1 + 3 * torch.arange(c, device=dev) / c)).half()
hook = amd.HessianHook(); hook.add_batch(x.view(2, 2048, c)); hook.invert()
gs = torch.full([c // 128], 128, dtype=torch.int32); gb = torch.full([c // 128], 4, dtype=torch.int32)
for r in (2048, 5504, 1376, 11008):
    w = (torch.randn(r, c, device=dev) * 0.02).half()
    t0 = time.time()
    res = amd.gptq_quant_set([w, w.clone()], hook, gs, gb, gptq_use_kernel=True, gptq_block_sizes=gs, quant_mse=False, save_device=dev)
    torch.cuda.synchronize()
    print(r, round(time.time() - t0, 3), res[0]["metrics"]["gptq_error"], flush=True)
