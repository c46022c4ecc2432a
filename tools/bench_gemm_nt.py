"""Dense x W^T kernel (gq_gemm_nt) at the replay shapes: check against an fp64 sample, then time it
beside the packed kernel and the library GEMM, interleaved in one process (experiment helper)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gptq_v2_amd as amd
dev = torch.device("cuda:0")
M = int(os.environ.get("GQ_M", "32768"))
shapes = [(M, 4096, 4096), (M, 11008, 4096), (M, 4096, 11008), (M, 12288, 4096), (M, 22016, 4096)]
if os.environ.get("GQ_SHAPES"):
    shapes = [tuple(int(v) for v in s.split("x")) for s in os.environ["GQ_SHAPES"].split(",")]
res = {}
for (m, r, c) in shapes:
    gen = torch.Generator(device=dev).manual_seed(r + c)
    x = (torch.randn(m, c, device=dev, generator=gen) * (1 + 3 * torch.arange(c, device=dev) / c)).half()
    q = torch.randint(0, 16, (r, c), device=dev, dtype=torch.uint8, generator=gen)
    scale = (torch.rand(r, c // 128, device=dev, generator=gen) * 0.01 + 0.002).half()
    zero = torch.randint(0, 16, (r, c // 128), device=dev, dtype=torch.uint8, generator=gen)
    gs = [128] * (c // 128)
    w = amd.ops.dequantize(q, scale, zero, None, gs)
    pk = amd.ops.pack_rows(q, 4, amd.ops.PACK_W4_KBLOCK)
    st, zt = scale.t().contiguous(), zero.t().contiguous()
    y = amd.ops.gemm_nt(x, w)
    rows = torch.randint(0, m, (64,), device=dev); cols = torch.randint(0, r, (256,), device=dev)
    ref = x[rows].double() @ w[cols].double().t()
    err = float((y[rows][:, cols].double() - ref).abs().max() / ref.abs().max())
    y4 = amd.ops.w4a16_gemm(x, pk, st, zt, 128)
    same = bool(torch.equal(y, y4))
    fns = {"nt": lambda: amd.ops.gemm_nt(x, w), "w4": lambda: amd.ops.w4a16_gemm(x, pk, st, zt, 128),
           "lib": lambda: torch.matmul(x, w.t())}
    if os.environ.get("GQ_ONLY"):
        fns = {k: v for k, v in fns.items() if k in os.environ["GQ_ONLY"].split(",")}
    times = {k: [] for k in fns}
    for rnd in range(4):
        for k, f in fns.items():
            f(); torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5): f()
            e.record(); torch.cuda.synchronize()
            times[k].append(s.elapsed_time(e) / 5)
    out = {"err": round(err, 6), "nt_equals_w4": same}
    for k, v in times.items():
        ms = sorted(v)[len(v) // 2]
        out[k] = (round(ms, 3), round(2.0 * m * r * c / ms / 1e9, 1))
    res[f"{m}x{r}x{c}"] = out
    print(f"{m}x{r}x{c}", json.dumps(out), flush=True)
    del x, w, y, y4, q, pk
    torch.cuda.empty_cache()
