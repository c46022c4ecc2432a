"""Does an XtX launch cost its tiles or its rounds?  One 32768-token call at hidden sizes whose lower-triangle tile
counts straddle multiples of the 256 CUs (experiment helper)."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gptq_v2_amd as amd
dev = torch.device("cuda:0")
out = []
for c in (9216, 9728, 10240, 10752, 11008, 11264, 11520, 11776):
    x = torch.randn(32768, c, device=dev).half()
    h = torch.zeros(c, c, device=dev)
    for _ in range(2):
        amd.ops.xtx_accumulate(h, x)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(6):
        amd.ops.xtx_accumulate(h, x)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 6
    nt = c // 256
    tiles = nt * (nt + 1) // 2
    out.append({"c": c, "tiles": tiles, "rounds": round(tiles / 256, 2), "ms": round(ms, 3), "us_per_tile_per_cu": round(ms * 1e3 / (tiles / 256), 1),
                "us_per_round_up": round(ms * 1e3 / -(-tiles // 256), 1)})
    del x, h
for o in out: print(json.dumps(o))
