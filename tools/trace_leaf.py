"""Cycles per phase of the 128x128 Cholesky + inverse leaf (experiment helper).  `build` compiles a
private copy with -DGQ_LEAF_TRACE into tools/_build/."""
import ctypes, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SO = os.path.join(HERE, "_build", "libleaftrace.so")

def build():
    csrc = os.path.join(ROOT, "gptq_v2_amd", "csrc")
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
           "-DGQ_LEAF_TRACE", *sys.argv[2:], "-I" + os.path.join(ROOT, "include"), "-I" + csrc, "-shared", "-o", SO] + \
          [os.path.join(csrc, f) for f in ("gq_core.hip", "hessian.hip", "sgemm.hip", "gemm_nt.hip")]
    subprocess.check_call(cmd)

def run():
    import torch
    lib = ctypes.CDLL(SO)
    lib.gq_hessian_invert_workspace.restype = ctypes.c_size_t
    dev = torch.device("cuda:0")
    c = 512
    a = torch.randn(c, 2 * c, device=dev)
    h = (a @ a.t()).contiguous()
    u = torch.empty_like(h)
    nbytes = lib.gq_hessian_invert_workspace(ctypes.c_int64(c))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    trace = torch.zeros(8, device=dev, dtype=torch.int64)
    assert lib.gq_debug_leaf_trace(ctypes.c_void_p(trace.data_ptr())) == 0
    tries = ctypes.c_int(0)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    rc = lib.gq_hessian_invert(p(h), p(u), None, None, ctypes.c_int64(c), ctypes.c_double(2.0), ctypes.c_float(0.01), 0, 5,
                               p(ws), ctypes.c_size_t(nbytes), ctypes.byref(tries), None)
    assert rc == 0
    T = [int(v) for v in trace.cpu()]
    names = ["load + store", "(a) diagonal blocks: load + factorisation", "(b) panels", "(c1) next panel's columns", "inverse pass 1", "inverse pass 2", "(a) inverse of the diagonal block + writes", "(a) wait for (c2) on waves 1-3"]
    tot = sum(T[:8])
    for n, v in zip(names, T):
        print(f"{n:42s} {v:8d} cycles  {100.0 * v / tot:5.1f} %")
    print(f"{'total':42s} {tot:8d} cycles")

if __name__ == "__main__":
    build() if sys.argv[1:2] == ["build"] else run()
