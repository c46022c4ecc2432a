"""H += X^T X at C = 4096 / 8192: one workgroup per tile against the token split (experiment helper)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gptq_v2_amd as amd
dev = torch.device("cuda:0")
for c, n in ((4096, 32768), (4096, 16384), (5120, 32768), (8192, 32768)):
    x = (torch.randn(n, c, device=dev) * (1 + 3 * torch.arange(c, device=dev) / c)).half()
    out = {}
    for parts in (1, 2, 3, 4, 5, 7, 9, 11, 13, 15):
        p = torch.zeros(parts, c, c, device=dev)
        f = (lambda: amd.ops.xtx_accumulate(p[0], x)) if parts == 1 else (lambda: amd.ops.xtx_accumulate_split(p, x))
        for _ in range(3): f()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(16): f()
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 16
        out[parts] = (round(ms, 4), round(2.0 * n * c * c / ms / 1e9, 1))
        del p
    print(c, n, "hint", amd.ops.xtx_split_hint(c, n), json.dumps(out), flush=True)
