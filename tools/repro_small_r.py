"""Row 0 of a layer quantised with R = 4, 8, 64, 128, 200 rows (row 0 identical, others random)
against the CPU oracle: rows are independent, so every run must give the same row 0."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gptq_v2_amd as amd
from oracle import gptq_oracle as oracle
gpu = torch.device("cuda:0")
c = 256
tokens = int(sys.argv[1]) if len(sys.argv) > 1 else 192
x = (torch.randn(2, tokens // 2, c, generator=torch.Generator().manual_seed(9)) * (1 + torch.arange(c) / c)).half().to(gpu)
hook = amd.HessianHook(); hook.add_batch(x); hook.invert()
gen = torch.Generator().manual_seed(10)
wall = (torch.randn(256, c, generator=gen) * 0.02).half()
gs, gb = torch.tensor([128, 128], dtype=torch.int32), torch.tensor([4, 4], dtype=torch.int32)
oh = oracle.OracleHessian()
oh.hessian, oh.hessian_inv = hook.hessian.cpu(), hook.hessian_inv.cpu()
oh.perm, oh.perm_inv = hook.perm.cpu(), hook.perm_inv.cpu()
want = oracle.gptq_quant(wall[:8].clone(), oh, gs, gb, use_kernel_semantics=True)["quant_meta"]["qweight"]
print("U diag min/max", float(hook.hessian_inv.diag().min()), float(hook.hessian_inv.diag().max()), "U absmax", float(hook.hessian_inv.abs().max()))
for r in (4, 8, 64, 128, 200, 256):
    res = amd.gptq_quant(wall[:r].to(gpu), hook, gs, gb, gptq_use_kernel=True, gptq_block_sizes=gs, quant_mse=False)["quant_meta"]["qweight"]
    n = min(r, 8)
    diff = (res[:n] != want[:n])
    print("R", r, "rows compared", n, "mismatching entries", int(diff.sum()), "first col", (int(diff.any(0).nonzero()[0]) if diff.any() else None))
