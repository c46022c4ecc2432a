"""Per-segment cycle stamps of the W4A16 256x256 kernel's main loop (experiment helper).
Builds a private copy of the kernel with -DGQ_W4_TRACE into tools/_build/ (run with `build` on a
machine with hipcc), then on a GPU runs one GEMM and prints, per wave, the cycles spent in each
segment of the last steady-state k-step:  R1 | bar | M1 | bar | R2 | bar | M2 | bar."""
import ctypes, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SO = os.path.join(HERE, "_build", os.environ.get("GQ_TRACE_SO", "libw4trace.so"))

def build():
    csrc = os.path.join(ROOT, "gptq_v2_amd", "csrc")
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
           "-DGQ_W4_TRACE", *sys.argv[2:], "-I" + os.path.join(ROOT, "include"), "-I" + csrc, "-shared", "-o", SO,
           os.path.join(csrc, "gq_core.hip"), os.path.join(csrc, "w4a16.hip"), os.path.join(csrc, "pack.hip")]
    subprocess.check_call(cmd)

def run():
    import torch
    lib = ctypes.CDLL(SO)
    dev = torch.device("cuda:0")
    m, r, c = [int(v) for v in os.environ.get("GQ_TRACE_SHAPE", "8192,4096,4096").split(",")]
    q = torch.randint(0, 16, (r, c), device=dev, dtype=torch.uint8)
    pk = torch.empty(r, c // 8, device=dev, dtype=torch.int32)
    assert lib.gq_pack_rows(ctypes.c_void_p(q.data_ptr()), ctypes.c_int64(r), ctypes.c_int64(c), 4, 2,
                            ctypes.c_void_p(pk.data_ptr()), None) == 0
    scale = (torch.rand(c // 128, r, device=dev) * 0.01 + 0.002).half()
    zero = torch.randint(0, 16, (c // 128, r), device=dev, dtype=torch.uint8)
    x = torch.randn(m, c, device=dev).half()
    y = torch.empty(m, r, device=dev, dtype=torch.float16)
    trace = torch.zeros(8 * 32, device=dev, dtype=torch.int64)
    assert lib.gq_debug_w4_trace(ctypes.c_void_p(trace.data_ptr())) == 0
    for _ in range(3):
        rc = lib.gq_w4a16_gemm(ctypes.c_void_p(y.data_ptr()), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(pk.data_ptr()),
                               ctypes.c_void_p(scale.data_ptr()), ctypes.c_void_p(zero.data_ptr()), ctypes.c_int64(m),
                               ctypes.c_int64(r), ctypes.c_int64(c), 128, None)
        assert rc == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n_rep = 20
    e0.record()
    for _ in range(n_rep):
        lib.gq_w4a16_gemm(ctypes.c_void_p(y.data_ptr()), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(pk.data_ptr()),
                          ctypes.c_void_p(scale.data_ptr()), ctypes.c_void_p(zero.data_ptr()), ctypes.c_int64(m),
                          ctypes.c_int64(r), ctypes.c_int64(c), 128, None)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n_rep
    tiles = ((m + 255) // 256) * ((r + 255) // 256)
    print(f"shape {m}x{r}x{c}: {ms:.4f} ms per launch = {2.0*m*r*c/ms/1e9:.1f} TFLOP/s; {tiles} tiles = {tiles/256:.2f} per CU")
    full = trace.cpu().view(8, 32)
    for w in (0, 4):
        k0, k1, k2, k3, r0, r3 = [int(v) for v in full[w, 13:19]]
        if w == 0:
            print(f"  tiles per CU x whole tile time = {tiles/256*(r3-r0)*10/1e6:.4f} ms")
        print(f"wave {w}: prologue {k1-k0} cyc, loop {k2-k1} cyc, epilogue {k3-k2} cyc, whole {k3-k0} cyc = "
              f"{(r3-r0)*10} ns realtime -> {(k3-k0)/((r3-r0)*10):.3f} GHz")
    t = full[:, :10]
    base = int(t.min())
    print("wave start |  R1: reads expand wait | bar |  M1: mfma loads | bar |  R2 | bar |  M2: mfma copy+dma | bar | total")
    for w in range(8):
        T = [int(v) for v in full[w, :13]]
        seq = [T[10]-T[0], T[11]-T[10], T[1]-T[11], T[2]-T[1], T[12]-T[2], T[3]-T[12], T[4]-T[3], T[5]-T[4], T[6]-T[5],
               T[7]-T[6], T[9]-T[7], T[8]-T[9], T[8]-T[0]]
        print(f"{w:4d} {T[0]-base:6d} | " + " ".join(str(v).rjust(5) for v in seq))

if __name__ == "__main__":
    build() if sys.argv[1:2] == ["build"] else run()
