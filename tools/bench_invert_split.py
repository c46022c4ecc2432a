"""Hessian inversion with the far updates as fp32 products (GQ_INVERT_SPLIT=0) or as split bf16 products where the cost
model picks them (default) or wherever the buffers fit (GQ_INVERT_SPLIT=1): time per inversion and U against an fp64
inverse factor (experiment helper; the variable is read once per process, so run it once per setting)."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gptq_v2_amd as amd
dev = torch.device("cuda:0")
out = {"mode": os.environ.get("GQ_INVERT_SPLIT", "auto")}
for c in [int(a) for a in sys.argv[1:]] or (4096, 11008):
    x = (torch.randn(4 * 2048 if c <= 16384 else 16 * 2048, c, device=dev) * (1 + 3 * torch.rand(c, device=dev))).half()
    best = 1e9
    for it in range(3):
        hook = amd.HessianHook(); hook.add_batch(x.view(-1, 2048, c))
        h0 = hook.hessian.clone() if it == 0 and c <= 12288 else None
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); u = hook.invert(damp_ratio=1e-2, act_order=True); e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e))
        if h0 is not None:
            # fp64 check of the first run: U^T U (H damped, permuted) = I
            hd = hook.hessian.double()
            ud = u.double()
            resid = (ud @ hd @ ud.t() - torch.eye(c, device=dev, dtype=torch.float64)).abs().max().item()
            out["c%d_resid" % c] = resid
    out["c%d_ms" % c] = round(best, 3)
    del x, hook, u
print(json.dumps(out))
