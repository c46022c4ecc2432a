"""Packed GEMM against expand-once + dense GEMM, per shape and token count, and q/k/v / gate/up as one layer
(experiment helper)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gptq_v2_amd as amd
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(7)

def make(r, c, perm_inv):
    meta = {"qweight": torch.randint(0, 16, (r, c), generator=gen, device=dev, dtype=torch.uint8),
            "scale": (torch.rand(r, c // 128, generator=gen, device=dev) * 0.01 + 0.002).half(),
            "qzero": torch.randint(0, 16, (r, c // 128), generator=gen, device=dev, dtype=torch.uint8),
            "perm_inv": perm_inv, "group_sizes": torch.full([c // 128], 128, dtype=torch.int16),
            "group_bit_widths": torch.full([c // 128], 4, dtype=torch.uint8)}
    return amd.QuantLinear.from_quant_meta(meta, dev, validate=False, has_perm=True)

def timed(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps

pi = {c: torch.argsort(torch.randperm(c, generator=gen, device=dev)).to(torch.int16) for c in (4096, 11008)}
shapes = [("4096x4096", 4096, 4096), ("12288x4096", 12288, 4096), ("11008x4096", 11008, 4096), ("22016x4096", 22016, 4096),
          ("4096x11008", 4096, 11008)]
layers = {name: make(r, c, pi[c]) for name, r, c in shapes}
for m in (2048, 4096, 8192, 16384):
    x = {c: torch.randn(m, c, generator=gen, device=dev).half() for c in (4096, 11008)}
    for name, r, c in shapes:
        l = layers[name]
        l.expand_min_tokens = 0
        tp = timed(lambda: l.forward_permuted(x[c]))
        l.expand_min_tokens = 1
        te = timed(lambda: l.forward_permuted(x[c]))
        tx = timed(lambda: amd.ops.wq_expand(l.qweight, l.scales, l.qzeros, l.group_size))
        fl = 2.0 * m * r * c
        print(f"M={m:6d} {name:11s} packed {tp:.4f} ms {fl/tp/1e9:7.1f} TF | expand+dense {te:.4f} ms {fl/te/1e9:7.1f} TF (expand alone {tx:.4f} ms)",
              flush=True)
