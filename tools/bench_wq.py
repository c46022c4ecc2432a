"""Micro-benchmark of the packed GEMM: split kernel against stream kernel at 4 bits, stream kernel at
dense 3 bits, fp16 and bf16 (experiment helper, not part of the product)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gptq_v2_amd as amd

dev = torch.device("cuda:0")
lib = amd._lib.load()
shapes = ((2048, 4096, 4096), (2048, 11008, 4096), (2048, 4096, 11008), (16384, 4096, 4096), (16384, 11008, 4096),
          (16384, 4096, 11008), (16384, 12288, 4096), (16384, 8192, 8192), (16384, 28672, 8192))
if len(sys.argv) > 1:
    shapes = shapes[:int(sys.argv[1])]
res = {}
for (m, r, c) in shapes:
    for dtype in (torch.float16, torch.bfloat16):
        scale = (torch.rand(c // 128, r, device=dev) * 0.01 + 0.002).to(dtype)
        x = torch.randn(m, c, device=dev).to(dtype)
        for bits, variant in ((4, 2), (4, 1), (3, 1)):
            q = torch.randint(0, 1 << bits, (r, c), device=dev, dtype=torch.uint8)
            pk = amd.ops.pack_rows(q, bits, amd.ops.PACK_W4_KBLOCK if bits == 4 else amd.ops.PACK_W3_KBLOCK)
            zero = torch.randint(0, 1 << bits, (c // 128, r), device=dev, dtype=torch.uint8)
            lib.gq_set_wq_kernel(variant)
            for _ in range(2):
                amd.ops.w4a16_gemm(x, pk, scale, zero, 128, bits=bits)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 10 if m <= 2048 else 4
            s.record()
            for _ in range(n):
                amd.ops.w4a16_gemm(x, pk, scale, zero, 128, bits=bits)
            e.record()
            torch.cuda.synchronize()
            ms = s.elapsed_time(e) / n
            key = f"{m}x{r}x{c} {'f16' if dtype == torch.float16 else 'bf16'} b{bits} {'stream' if variant == 1 else 'split'}"
            res[key] = (round(ms, 3), round(2.0 * m * r * c / ms / 1e9, 1))
            print(key, res[key], flush=True)
            del q, pk, zero
lib.gq_set_wq_kernel(0)
print(json.dumps(res))
