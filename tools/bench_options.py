"""Cost of gptq_quant's optional grid refinements at a Llama-2-7B layer (experiment helper)."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gptq_v2_amd as amd
dev = torch.device("cuda:0")
c = r = 4096
x = (torch.randn(8192, c, device=dev) * (1 + 3 * torch.arange(c, device=dev) / c)).half()
w = (torch.randn(r, c, device=dev) * 0.02).half()
gs = torch.full([c // 128], 128, dtype=torch.int32); gb = torch.full([c // 128], 4, dtype=torch.int32)
hook = amd.HessianHook(); hook.add_batch(x.view(4, 2048, c)); hook.invert()
res = {}
for name, kw in (("plain", dict(quant_mse=False)), ("mse", dict(quant_mse=True)),
                 ("exl2", dict(quant_mse=False, scale_bit_width=torch.tensor(4))),
                 ("mse+exl2", dict(quant_mse=True, scale_bit_width=torch.tensor(4)))):
    for it in range(2):
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        out = amd.gptq_quant(w, hook, gs, gb, gptq_use_kernel=True, gptq_block_sizes=gs, save_device=dev, **kw)
        e.record(); torch.cuda.synchronize()
    res[name] = {"ms": round(s.elapsed_time(e), 2), "l2_error": out["metrics"]["l2.0_error"],
                 "gptq_error": out["metrics"]["gptq_error"]}
print(json.dumps(res))
