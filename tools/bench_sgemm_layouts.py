"""fp32 GEMM timing per operand layout, with and without padded leading dimensions (experiment helper;
build with `python tools/bench_sgemm.py build`)."""
import ctypes, os, sys
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "_build", "libsgemm_dbg.so"))
dev = torch.device("cuda:0")
p = lambda t: ctypes.c_void_p(t.data_ptr())
for (m, n, k) in [(4096, 4096, 4096), (4096, 10496, 512), (11008, 10496, 512)]:
    for pad in (0, 32):
        row = []
        for layout in range(4):
            ra, ca = ((k, m) if layout & 1 else (m, k))
            rb, cb = ((k, n) if layout & 2 else (n, k))
            a = torch.randn(ra, ca + pad, device=dev); b = torch.randn(rb, cb + pad, device=dev); c = torch.zeros(m, n + pad, device=dev)
            res = []
            for plain in (1, 0):
                lib.gq_debug_sgemm_plain(plain)
                call = lambda: lib.gq_debug_sgemm_ld(p(a), p(b), p(c), m, n, k, layout, 0, 0, 1, 0, None, pad, 1)
                for _ in range(2): assert call() == 0
                torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(5): call()
                e.record(); torch.cuda.synchronize()
                res.append(round(2.0 * m * n * k / (s.elapsed_time(e) / 5) / 1e9, 1))
            row.append(f"L{layout} {res[0]}->{res[1]}")
        print((m, n, k), "pad", pad, " | ".join(row), flush=True)
