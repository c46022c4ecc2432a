"""H += X^T X per call against the tokens of a call, at the token split the hint picks and around it
(experiment helper): how much of a call is the read-modify-write of H and the part-full last round."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gptq_v2_amd as amd
dev = torch.device("cuda:0")
for c in (4096, 11008):
    for n in (32768, 65536, 131072):
        if n * c * 2 >= (1 << 32):
            continue
        x = (torch.randn(n, c, device=dev) * (1 + 3 * torch.rand(c, device=dev))).half()
        out = {}
        hint = amd.ops.xtx_split_hint(c, n)
        for parts in sorted({1, hint, 3, 5, 7, 9, 13, 15}):
            if c == 11008 and parts > 4:
                continue
            p = torch.zeros(parts, c, c, device=dev)
            f = (lambda: amd.ops.xtx_accumulate(p[0], x)) if parts == 1 else (lambda: amd.ops.xtx_accumulate_split(p, x))
            for _ in range(2): f()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(8): f()
            e.record(); torch.cuda.synchronize()
            ms = s.elapsed_time(e) / 8
            out[parts] = (round(ms, 4), round(2.0 * n * c * c / ms / 1e9, 1))
            del p
        print(c, n, "hint", hint, json.dumps(out), flush=True)
        del x
