import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gptq_v2_amd as amd
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
for r, c in ((4096, 4096), (11008, 4096), (4096, 11008)):
    meta = {"qweight": torch.randint(0, 16, (r, c), generator=g, device=dev, dtype=torch.uint8),
            "scale": (torch.rand(r, c // 128, generator=g, device=dev) * 0.01 + 0.002).half(),
            "qzero": torch.randint(0, 16, (r, c // 128), generator=g, device=dev, dtype=torch.uint8),
            "perm_inv": torch.argsort(torch.randperm(c, generator=g, device=dev)).to(torch.int16),
            "group_sizes": torch.full([c // 128], 128, dtype=torch.int16), "group_bit_widths": torch.full([c // 128], 4, dtype=torch.uint8)}
    for _ in range(3): w = amd.quant.reconstruct_weight(meta, torch.float16, dev)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): w = amd.quant.reconstruct_weight(meta, torch.float16, dev)
    e.record(); torch.cuda.synchronize()
    print(r, c, round(s.elapsed_time(e) / 10 * 1000, 1), "us per reconstruct_weight")
