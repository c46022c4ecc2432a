"""Cycle stamps of the XtX 256-tile kernel (experiment helper): `build`, then run on a GPU."""
import ctypes, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SO = os.path.join(HERE, "_build", "libxtxtrace.so")

def build():
    csrc = os.path.join(ROOT, "gptq_v2_amd", "csrc")
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
                           "-DGQ_XTX_TRACE", *sys.argv[2:], "-I" + os.path.join(ROOT, "include"), "-I" + csrc, "-shared", "-o", SO,
                           os.path.join(csrc, "gq_core.hip"), os.path.join(csrc, "xtx.hip")])

def run():
    import torch
    lib = ctypes.CDLL(SO)
    dev = torch.device("cuda:0")
    c = int(os.environ.get("C", "11008")); n = 32768
    x = torch.randn(n, c, device=dev).half()
    h = torch.zeros(c, c, device=dev)
    trace = torch.zeros(8 * 16, device=dev, dtype=torch.int64)
    assert lib.gq_debug_xtx_trace(ctypes.c_void_p(trace.data_ptr())) == 0
    for _ in range(2):
        assert lib.gq_xtx_accumulate(ctypes.c_void_p(h.data_ptr()), ctypes.c_void_p(x.data_ptr()), 0, ctypes.c_int64(n),
                                     ctypes.c_int64(c), 0, 1, None) == 0
    torch.cuda.synchronize()
    t = trace.cpu().view(8, 16)
    for w in range(8):
        T = [int(v) for v in t[w]]
        ns = (T[12] - T[11]) * 10
        print(f"wave {w}: prologue {T[1]-T[0]} loop {T[9]-T[1]} ({(T[9]-T[1])/1024:.0f}/step) epilogue {T[10]-T[9]} total {T[10]-T[0]} = {ns} ns "
              f"({(T[10]-T[0])/max(ns,1):.2f} GHz) | last step: reads {T[3]-T[2]} vmcnt {T[4]-T[3]} lgkm {T[5]-T[4]} bar {T[6]-T[5]} mfma {T[7]-T[6]} bar {T[8]-T[7]}")

if __name__ == "__main__":
    build() if sys.argv[1:2] == ["build"] else run()
