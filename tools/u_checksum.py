"""SHA-256 of U (hessian_inv) and perm for fixed seeded inputs at several sizes: a change of schedule inside the
inversion's kernels must leave these digests as they are (experiment helper)."""
import hashlib, os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gptq_v2_amd as amd
dev = torch.device("cuda:0")
out = {}
for c in (128, 384, 1024, 4096):
    gen = torch.Generator(device=dev).manual_seed(c)
    x = (torch.randn(2048, c, generator=gen, device=dev) * (1 + 3 * torch.rand(c, generator=gen, device=dev))).half()
    hook = amd.HessianHook(); hook.add_batch(x)
    u = hook.invert(damp_ratio=1e-2, act_order=True)
    out[c] = hashlib.sha256(u.cpu().numpy().tobytes() + hook.perm.cpu().numpy().tobytes()).hexdigest()[:16]
print(json.dumps(out))
