"""Does a chain of short kernels on one stream run beside a chip-filling GEMM on another?  Four inversions at
C = 1024 (leaves of one workgroup, small products; enqueued whole, no host wait) alone, and started right after an
8192^3 fp32 product on a second stream.  GQ_PROBE_PRIO=1 gives the chain's stream the high priority.
Measured on MI355X: chain alone 1.6 ms, product alone 8.6 ms, chain beside the product 10.3 ms -- with or without the
priority, and also with the product held to one workgroup per CU: while a kernel has workgroups waiting to be
placed, another queue's kernels do not start.  Streams overlap only where a kernel leaves CUs unclaimed."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import gptq_v2_amd as amd
dev = torch.device("cuda:0")
c = 1024
a = torch.randn(c, 2 * c, device=dev)
h0 = (a @ a.t()).contiguous()
xbig = torch.randn(8192, 8192, device=dev)      # fp32 XtX: an 8192 x 8192 x 8192 product on sgemm_dma (4096 tiles)
hbig = torch.zeros(8192, 8192, device=dev)
sa, sb = torch.cuda.Stream(dev, priority=-1 if os.environ.get("GQ_PROBE_PRIO") else 0), torch.cuda.Stream(dev)
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else None, "chain stream priority", sa.priority)
keep = []
def chain(n=4):
    for _ in range(n):
        h = h0.clone()
        keep.append(amd.ops.hessian_invert_begin(h, 2.0, 0.01, False))   # enqueued whole, no host wait
def timed(with_big):
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    if with_big:
        with torch.cuda.stream(sb):
            amd.ops.xtx_accumulate_f32(hbig, xbig) if hasattr(amd.ops, "xtx_accumulate_f32") else None
            e2.record()
    with torch.cuda.stream(sa):
        e0.record(); chain(); e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1), 3)
print([n for n in dir(amd.ops) if "xtx" in n])
for rep in range(2):
    print("chain alone", timed(False), "ms; chain beside the big product", timed(True), "ms")
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(); amd.ops.xtx_accumulate_f32(hbig, xbig); e.record(); torch.cuda.synchronize()
print("big product alone", round(s.elapsed_time(e), 3), "ms")
