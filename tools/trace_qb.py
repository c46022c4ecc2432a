"""Cycle stamps of the column-loop block kernel (experiment helper).  `build` compiles a private
copy with -DGQ_QB_TRACE into tools/_build/; run on a GPU prints, for workgroup 0 and each wave,
the cycles of: prologue | per 16-column sub-block: walk, stores, chunk updates, barrier."""
import ctypes, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SO = os.path.join(HERE, "_build", "libqbtrace.so")

def build():
    csrc = os.path.join(ROOT, "gptq_v2_amd", "csrc")
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
           "-DGQ_QB_TRACE", *sys.argv[2:], "-I" + os.path.join(ROOT, "include"), "-I" + csrc, "-shared", "-o", SO] + \
          [os.path.join(csrc, f) for f in ("gq_core.hip", "quantize.hip", "sgemm.hip")]
    subprocess.check_call(cmd)

def run():
    import torch
    lib = ctypes.CDLL(SO)
    dev = torch.device("cuda:0")
    c, r = 1024, 4096
    g = torch.Generator(device=dev).manual_seed(0)
    a = torch.randn(c, c, device=dev, generator=g)
    u = torch.linalg.cholesky(a @ a.t() / c + torch.eye(c, device=dev), upper=True).contiguous()
    wt = (torch.randn(c, r, device=dev, generator=g) * 0.02).contiguous()
    quant = torch.empty_like(wt); err = torch.empty_like(wt)
    outq = torch.empty(c, r, device=dev, dtype=torch.int16)
    scale = (wt[:128].amax(0) - wt[:128].amin(0)) / 15 + 1e-12
    zero = torch.round(-wt[:128].amin(0) / scale)
    trace = torch.zeros(4 * 40, device=dev, dtype=torch.int64)
    assert lib.gq_debug_qb_trace(ctypes.c_void_p(trace.data_ptr())) == 0
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    for _ in range(2):
        rc = lib.gq_quantize_range(p(quant), p(scale), p(outq), p(zero), ctypes.c_float(15.0), p(u), p(wt), p(err),
                                   ctypes.c_int64(c), ctypes.c_int64(r), ctypes.c_int64(0), ctypes.c_int64(128), None)
        assert rc == 0
    torch.cuda.synchronize()
    t = trace.cpu().view(4, 40)
    for w in range(4):
        T = [int(v) for v in t[w]]
        print(f"wave {w}: prologue {T[1]-T[0]} = issue loads {T[34]-T[0]}, lds stores (wait for loads) {T[35]-T[34]}, recip {T[36]-T[35]}, barrier {T[37]-T[36]}, grid {T[1]-T[37]}")
        for jb in range(8):
            b = 2 + jb * 4
            prev = T[1] if jb == 0 else T[b - 1]
            print(f"   sub {jb}: walk {T[b]-prev:6d} stores {T[b+1]-T[b]:6d} chunks {T[b+2]-T[b+1]:6d} barrier {T[b+3]-T[b+2]:6d}")
        print(f"   total {T[33]-T[0]}")

if __name__ == "__main__":
    build() if sys.argv[1:2] == ["build"] else run()
