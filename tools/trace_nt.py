"""Per-workgroup timeline of the dense x W^T kernel (experiment helper).  `build` compiles a private
copy with -DGQ_NT_TRACE into tools/_build/; on a GPU, runs one GEMM and prints, from the 100 MHz
real-time stamps of every workgroup: prologue / k-loop / epilogue times, and per CU the gaps between
consecutive workgroups."""
import ctypes, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SO = os.path.join(HERE, "_build", "libnttrace.so")

def build():
    csrc = os.path.join(ROOT, "gptq_v2_amd", "csrc")
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
           "-DGQ_NT_TRACE", *sys.argv[2:], "-I" + os.path.join(ROOT, "include"), "-I" + csrc, "-shared", "-o", SO,
           os.path.join(csrc, "gq_core.hip"), os.path.join(csrc, "gemm_nt.hip")]
    subprocess.check_call(cmd)

def run():
    import torch, numpy as np
    lib = ctypes.CDLL(SO)
    dev = torch.device("cuda:0")
    m, r, c = [int(v) for v in os.environ.get("GQ_TRACE_SHAPE", "32768,4096,4096").split(",")]
    x = (torch.randn(m, c, device=dev) * (1 + 3 * torch.arange(c, device=dev) / c)).half()
    w = (torch.randn(r, c, device=dev) * 0.02).half()
    y = torch.empty(m, r, device=dev, dtype=torch.float16)
    tiles = ((m + 255) // 256) * ((r + 255) // 256)
    trace = torch.zeros(tiles * 8, device=dev, dtype=torch.int64)
    assert lib.gq_debug_nt_trace(ctypes.c_void_p(trace.data_ptr())) == 0
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    call = lambda: lib.gq_gemm_nt(p(y), p(x), p(w), ctypes.c_int64(m), ctypes.c_int64(r), ctypes.c_int64(c), 0, None)
    for _ in range(3):
        assert call() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): call()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"shape {m}x{r}x{c}: {ms:.4f} ms per launch = {2.0*m*r*c/ms/1e9:.1f} TFLOP/s; {tiles} tiles = {tiles/256:.2f} per CU")
    t = trace.cpu().numpy().reshape(tiles, 8)
    t0 = t[:, 0].min()
    us = lambda v: v / 100.0
    pro, loop, epi = us(t[:, 1] - t[:, 0]), us(t[:, 2] - t[:, 1]), us(t[:, 3] - t[:, 2])
    q = lambda a: f"min {a.min():.1f} med {np.median(a):.1f} max {a.max():.1f}"
    print(f"prologue us: {q(pro)}\nk-loop us:   {q(loop)}\nepilogue us: {q(epi)}\nlaunch span: {us(t[:,3].max()-t0):.1f} us")
    cyc = t[:, 7].astype(np.float64)
    nk = c // 64
    print(f"k-loop cycles: {q(cyc)}  per k-step med {np.median(cyc)/nk:.0f}; clock GHz: {q(cyc / (loop * 1e3))}")
    cu = (t[:, 5] & 0xF) * 4096 + (t[:, 4] & 0xFFF00 >> 0)   # xcc id, then se/sh/cu bits of HW_ID
    cu = (t[:, 5] & 0xF) * 65536 + ((t[:, 4] >> 8) & 0xFFFF)
    gaps, first, counts = [], [], []
    for c_ in np.unique(cu):
        rows = t[cu == c_]
        rows = rows[np.argsort(rows[:, 0])]
        counts.append(len(rows))
        first.append(us(rows[0, 0] - t0))
        for a, b in zip(rows[:-1], rows[1:]):
            gaps.append(us(b[0] - a[3]))
    gaps = np.array(gaps)
    print(f"CUs seen: {len(counts)}, workgroups per CU min {min(counts)} max {max(counts)}; first start us: {q(np.array(first))}")
    if len(gaps):
        print(f"gap end->next start on the same CU, us: {q(gaps)}  mean {gaps.mean():.2f}")
    order = np.argsort(t[:, 0])
    for i in list(order[:3]) + list(order[len(order)//2:len(order)//2+3]) + list(order[-3:]):
        print(f"  wg bid {t[i,6]:5d} start {us(t[i,0]-t0):8.1f} pro {pro[i]:6.1f} loop {loop[i]:7.1f} epi {epi[i]:6.1f}")

if __name__ == "__main__":
    build() if sys.argv[1:2] == ["build"] else run()
