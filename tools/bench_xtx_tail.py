"""XtX with and without the tail split (gq_xtx_accumulate_tail): per-call time at hidden sizes that leave a partly
filled last round, 32768 and 16384 tokens per call (experiment helper)."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gptq_v2_amd as amd
dev = torch.device("cuda:0")


def timed(fn, reps=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


for c in [int(a) for a in sys.argv[1:]] or (11008, 5888, 9728, 13824, 14336):
    need = amd.ops.xtx_tail_workspace(c)
    if not need:
        print(json.dumps({"c": c, "tail": None})); continue
    for n in (32768, 16384):
        x = torch.randn(n, c, device=dev).half()
        h = torch.zeros(c, c, device=dev)
        ws = torch.zeros(need // 4, device=dev)
        plain = timed(lambda: amd.ops.xtx_accumulate(h, x))
        tail = timed(lambda: amd.ops.xtx_accumulate_tail(h, ws, x))
        fold = timed(lambda: amd.ops.xtx_fold_tail(h, ws))
        nt = c // 256
        print(json.dumps({"c": c, "n": n, "tiles": nt * (nt + 1) // 2, "runs_x_tail_tiles": need // (256 * 256 * 4),
                          "plain_ms": round(plain, 3), "tail_ms": round(tail, 3), "fold_ms": round(fold, 3)}), flush=True)
        del x, h, ws
