"""Invert + quantize of one layer at C = 4096 / 11008 (experiment helper for rocprofv3 traces)."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gptq_v2_amd as amd
dev = torch.device("cuda:0")
res = {}
for c, r in ((4096, 4096), (11008, 4096)):
    x = (torch.randn(8192, c, device=dev) * (1 + 3 * torch.arange(c, device=dev) / c)).half()
    w = (torch.randn(r, c, device=dev) * 0.02).half()
    gs = torch.full([c // 128], 128, dtype=torch.int32); gb = torch.full([c // 128], 4, dtype=torch.int32)
    for it in range(2):
        hook = amd.HessianHook(); hook.add_batch(x.view(4, 2048, c))
        torch.cuda.synchronize()
        s, e, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        s.record(); hook.invert(damp_ratio=1e-2, act_order=True); e.record()
        amd.gptq_quant(w, hook, gs, gb, scale_bit_width=None, gptq_use_kernel=True, gptq_block_sizes=gs,
                       quant_symmetric=False, quant_mse=False, save_device=dev)
        e2.record(); torch.cuda.synchronize()
    res[c] = {"invert_ms": round(s.elapsed_time(e), 2), "quant_ms": round(e.elapsed_time(e2), 2)}
print(json.dumps(res))
