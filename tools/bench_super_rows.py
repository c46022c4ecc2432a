"""Column loop of one Linear with 4096 columns at several row counts (128 / 256 / 344 / 512 workgroups of the
super-block kernel on 256 CUs): how much a second workgroup on a CU costs (experiment helper; run under
rocprofv3 --kernel-trace and summarise with tools/rocpd_kernels.py, or read the wall times printed here)."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gptq_v2_amd as amd
dev = torch.device("cuda:0")
c = 4096
x = (torch.randn(8192, c, device=dev) * (1 + 3 * torch.rand(c, device=dev))).half()
hook = amd.HessianHook(); hook.add_batch(x.view(4, 2048, c)); hook.invert(damp_ratio=1e-2, act_order=True)
gs = torch.full([c // 128], 128, dtype=torch.int32); gb = torch.full([c // 128], 4, dtype=torch.int32)
out = {}
for r in (4096, 8192, 11008, 16384):
    w = (torch.randn(r, c, device=dev) * 0.02).half()
    best = 1e9
    for it in range(4):
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        amd.gptq_quant(w, hook, gs, gb, scale_bit_width=None, gptq_use_kernel=True, gptq_block_sizes=gs, quant_symmetric=False,
                       quant_mse=False, save_device=dev)
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e))
    out[r] = round(best, 3)
print(json.dumps(out))
