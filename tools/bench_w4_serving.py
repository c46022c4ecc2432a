"""W4A16 GEMM at the QuantLinear.forward shapes of a Llama-2-7B block, seq 2048 x batch 1 / 8
(experiment helper).  GQ_W4_TILE=128 / 256 / 384 forces a tile shape (384 = 256 tokens x 128 rows); argv[1] = f16 (default) or bf16."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gptq_v2_amd as amd
dev = torch.device("cuda:0")
dt = torch.bfloat16 if len(sys.argv) > 1 and sys.argv[1] == "bf16" else torch.float16
res = {}
for m in (2048, 16384):
    tot_ms, tot_fl = 0.0, 0.0
    for name, r, c, n in (("qkvo", 4096, 4096, 4), ("gate_up", 11008, 4096, 2), ("down", 4096, 11008, 1)):
        q = torch.randint(0, 16, (r, c), device=dev, dtype=torch.uint8)
        pk = amd.ops.pack_rows(q, 4, amd.ops.PACK_W4_KBLOCK)
        scale = (torch.rand(c // 128, r, device=dev) * 0.01 + 0.002).to(dt)
        zero = torch.randint(0, 16, (c // 128, r), device=dev, dtype=torch.uint8)
        x = torch.randn(m, c, device=dev).to(dt)
        for _ in range(3): amd.ops.w4a16_gemm(x, pk, scale, zero, 128)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): amd.ops.w4a16_gemm(x, pk, scale, zero, 128)
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 20
        res[f"m{m}_{name}"] = (round(ms, 4), round(2.0 * m * r * c / ms / 1e9, 1))
        tot_ms += n * ms; tot_fl += n * 2.0 * m * r * c
    res[f"m{m}_block"] = (round(tot_ms, 3), round(tot_fl / tot_ms / 1e9, 1))
print(json.dumps(res))
