"""Column loops of the four input-sharing sets of a Llama-2-7B block, each timed alone (experiment helper)."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gptq_v2_amd as amd
dev = torch.device("cuda:0")
res = {}
hooks = {}
for c in (4096, 11008):
    x = (torch.randn(8192, c, device=dev) * (1 + 3 * torch.arange(c, device=dev) / c)).half()
    hook = amd.HessianHook(); hook.add_batch(x.view(4, 2048, c)); hook.invert(damp_ratio=1e-2, act_order=True)
    hooks[c] = hook
from gptq_v2_amd import _lib
for mode, name, c, outs in [(m,) + t for m in (0, 1, 0, 1) for t in (("qkv", 4096, [4096] * 3), ("o", 4096, [4096]), ("gate_up", 4096, [11008] * 2), ("down", 11008, [4096]))]:
    _lib.load().gq_set_quant_super(mode)
    ws = [(torch.randn(r, c, device=dev) * 0.02).half() for r in outs]
    gs = torch.full([c // 128], 128, dtype=torch.int32); gb = torch.full([c // 128], 4, dtype=torch.int32)
    best = []
    for it in range(4):
        torch.cuda.synchronize()
        s, e, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        s.record()
        part = amd.gptq_quant_set(ws, hooks[c], gs, gb, scale_bit_width=None, gptq_use_kernel=True, gptq_block_sizes=gs,
                                  quant_symmetric=False, quant_mse=False, save_device=dev, wait=False)
        e.record()
        part.settle()
        e2.record(); torch.cuda.synchronize()
        best.append((round(s.elapsed_time(e), 3), round(s.elapsed_time(e2), 3)))
    res.setdefault(("super" if mode else "blocks"), {}).setdefault(name, []).append(min(b[0] for b in best[1:]))
    res[name] = {"levels_ms": min(b[0] for b in best[1:]), "with_metrics_ms": min(b[1] for b in best[1:])}
print(json.dumps(res))
