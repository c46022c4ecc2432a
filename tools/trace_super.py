"""One 4096 x 4096 and one 4096 x 11008 layer through the column loop, both launch forms (run under rocprofv3 --kernel-trace)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gptq_v2_amd as amd
from gptq_v2_amd import _lib, ops
dev = torch.device("cuda:0")
for c in (4096, 11008):
    x = (torch.randn(8192, c, device=dev) * (1 + 3 * torch.rand(c, device=dev))).half()
    hook = amd.HessianHook(); hook.add_batch(x.view(4, 2048, c)); hook.invert()
    w = (torch.randn(4096, c, device=dev) * 0.02).half()
    gs, gb = [128] * (c // 128), [4] * (c // 128)
    for mode in (1, 0, 1, 0):
        _lib.load().gq_set_quant_super(mode)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); ops.quantize_layer(w, hook.perm.int(), hook.hessian_inv, gs, gb); e.record(); torch.cuda.synchronize()
        print(c, "super" if mode else "blocks", round(s.elapsed_time(e), 3), "ms", flush=True)
