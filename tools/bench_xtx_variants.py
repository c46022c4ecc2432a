"""One 32768-token XtX call (plain gq_xtx_accumulate, no token split) per hidden size, with the variant the library
picks and with the 256-tile and 128-tile kernels forced (run three times: GQ_XTX_VARIANT unset / 256 / 32x2)."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gptq_v2_amd as amd
dev = torch.device("cuda:0")
res = {}
for c in [int(a) for a in sys.argv[1:]] or (768, 1024, 1536, 2048, 3072, 5120, 6144, 8192, 10240, 10752, 11520, 13824):
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
    res[c] = round(s.elapsed_time(e) / 6, 4)
    del x, h
print(json.dumps(res))
