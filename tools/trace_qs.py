"""Cycle stamps of the super-block column-loop kernel (experiment helper).  `build` compiles a private copy of the
library with -DGQ_QS_TRACE into tools/_build/; run on a GPU prints, for workgroup 0 and each wave, the cycles of the
first phases: wave 0: look-ahead tile | walk | exports | barrier; waves 1-3: request operands | tiles | stores | barrier."""
import ctypes, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SO = os.path.join(HERE, "_build", "libqstrace.so")

def build():
    csrc = os.path.join(ROOT, "gptq_v2_amd", "csrc")
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    sys.path.insert(0, os.path.join(ROOT, "gptq_v2_amd"))
    import build as b
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
           "-DGQ_QS_TRACE", *sys.argv[2:], "-I" + os.path.join(ROOT, "include"), "-I" + csrc, "-shared", "-o", SO] + \
          [os.path.join(csrc, f) for f in b.SOURCES]
    subprocess.check_call(cmd)

def run():
    import torch
    sys.path.insert(0, ROOT)
    os.environ["GQ_LIB_PATH"] = SO
    from gptq_v2_amd import _lib
    import gptq_v2_amd as amd
    from gptq_v2_amd import ops
    lib = _lib.load()
    lib.gq_debug_qs_trace.argtypes = [ctypes.c_void_p]
    dev = torch.device("cuda:0")
    c, r = 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    x = (torch.randn(4096, c, device=dev) * (1 + 3 * torch.rand(c, device=dev))).half()
    hook = amd.HessianHook(); hook.add_batch(x.view(2, 2048, c)); hook.invert()
    w = (torch.randn(r, c, device=dev) * 0.02).half()
    trace = torch.zeros(4 * 12 * 5, device=dev, dtype=torch.int64)
    assert lib.gq_debug_qs_trace(ctypes.c_void_p(trace.data_ptr())) == 0
    for _ in range(2):
        ops.quantize_layer(w, hook.perm.int(), hook.hessian_inv, [128] * (c // 128), [4] * (c // 128))
    torch.cuda.synchronize()
    t = trace.cpu().view(4, 12, 5)
    for wv in range(4):
        print(f"wave {wv}:")
        for j in range(12):
            T = [int(v) for v in t[wv, j]]
            names = ("look-ahead", "walk", "exports") if wv == 0 else ("requests", "tiles", "stores+fetch")
            print(f"   phase {j:2d}: {names[0]} {T[1]-T[0]:6d} {names[1]} {T[2]-T[1]:6d} {names[2]} {T[3]-T[2]:6d} barrier {T[4]-T[3]:6d}   phase {T[4]-T[0]:6d}")

if __name__ == "__main__":
    build() if sys.argv[1:2] == ["build"] else run()
