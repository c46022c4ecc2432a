"""Timing of the library's fp32 MFMA GEMM at square sizes and at the trailing-update shape
(experiment helper).  `build` compiles sgemm.hip with a C shim into tools/_build/."""
import ctypes, os, subprocess, sys, json
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SO = os.path.join(HERE, "_build", "libsgemm_dbg.so")

def build():
    csrc = os.path.join(ROOT, "gptq_v2_amd", "csrc")
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
                           *sys.argv[2:], "-I" + os.path.join(ROOT, "include"), "-I" + csrc, "-shared", "-o", SO,
                           os.path.join(csrc, "gq_core.hip"), os.path.join(csrc, "sgemm.hip"), os.path.join(HERE, "src", "sgemm_shim.hip")])

def run():
    import torch
    lib = ctypes.CDLL(SO)
    dev = torch.device("cuda:0")
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    res = {}
    cases = [("sq1024", 1024, 1024, 1024, 0, 0), ("sq4096", 4096, 4096, 4096, 0, 0), ("sq5504", 5504, 5504, 5504, 0, 0),
             ("trail_far", 4096, 10496, 512, 3, 1), ("trail_near", 4096, 384, 128, 3, 1), ("kn4096", 4096, 4096, 4096, 2, 0)]
    for name, m, n, k, layout, acc in cases:
        a = torch.randn((k, m) if layout & 1 else (m, k), device=dev)
        b = torch.randn((k, n) if layout & 2 else (n, k), device=dev)
        c = torch.zeros(m, n, device=dev)
        call = lambda: lib.gq_debug_sgemm(p(a), p(b), p(c), m, n, k, layout, 0, 0, acc, 0, None)
        for _ in range(2): assert call() == 0
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5): call()
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 5
        res[name] = (round(ms, 4), round(2.0 * m * n * k / ms / 1e9, 1))
        if name == "sq1024":
            ref = a.double() @ b.double().t()
            c.zero_(); call(); torch.cuda.synchronize()
            res["check_rel"] = float(((c.double() - ref).abs().max() / ref.abs().max()).item())
    print(json.dumps(res))
    if hasattr(lib, "gq_debug_sg_trace"):
        trace = torch.zeros(32, device=dev, dtype=torch.int64)
        lib.gq_debug_sg_trace(p(trace))
        for (m, n, k, layout, acc) in [(4096, 4096, 4096, 2, 0), (4096, 10496, 512, 3, 1)]:
            a = torch.randn((k, m) if layout & 1 else (m, k), device=dev); b = torch.randn((k, n) if layout & 2 else (n, k), device=dev)
            c = torch.zeros(m, n, device=dev)
            for _ in range(2): lib.gq_debug_sgemm(p(a), p(b), p(c), m, n, k, layout, 0, 0, acc, 0, None)
            torch.cuda.synchronize()
            t = trace.cpu().view(4, 8)
            print((m, n, k), "layout", layout, "accumulate", acc)
            for w in range(4):
                T = [int(v) for v in t[w]]
                print(f"wave {w}: prologue {T[1]-T[0]} loop {T[2]-T[1]} ({(T[2]-T[1])/max(T[6],1):.0f}/step of 32 k; pure MFMA issue 4096) epilogue {T[3]-T[2]} "
                      f"clock {(T[3]-T[0])/max((T[5]-T[4])*10,1):.2f} GHz")

if __name__ == "__main__":
    build() if sys.argv[1:2] == ["build"] else run()
