"""Bit-for-bit comparison of the hand-scheduled fp32 GEMM kernel with the compiler-scheduled one, over
every layout, K trim, sign, accumulate flag, lower-only, batch and padded leading dimensions, and against
float64 (experiment helper; build with `python tools/bench_sgemm.py build`)."""
import ctypes, os, sys
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "_build", "libsgemm_dbg.so"))
dev = torch.device("cuda:0")
p = lambda t: ctypes.c_void_p(t.data_ptr())
bad = 0
n_cases = 0
for layout in range(4):
    for trim, negate, acc, lower, pad, batch, (m, n, k) in [
            (0, 0, 0, 0, 0, 1, (2048, 2048, 64)), (0, 1, 1, 0, 0, 1, (2048, 2560, 96)), (1, 0, 1, 0, 0, 1, (2048, 2048, 2048)),
            (2, 1, 0, 0, 0, 1, (2048, 2048, 2048)), (3, 0, 0, 0, 0, 1, (2048, 2048, 2048)), (0, 1, 1, 1, 0, 1, (3072, 3072, 160)),
            (0, 0, 1, 0, 8, 1, (2048, 2304, 128)), (0, 1, 0, 0, 4, 3, (1280, 1280, 256)), (0, 0, 1, 0, 0, 1, (2048, 2048, 32)),
            # short products on 64-tiles (fewer than 256 tiles of 128): one round trip per tile
            (0, 1, 1, 0, 0, 1, (1024, 384, 128)), (1, 0, 0, 0, 0, 1, (1984, 128, 128)), (2, 0, 1, 0, 0, 1, (960, 128, 128)),
            (3, 1, 0, 0, 0, 1, (128, 1024, 128)), (0, 1, 1, 1, 0, 1, (768, 768, 96)), (0, 0, 1, 0, 12, 5, (192, 320, 64)),
            (0, 1, 1, 0, 0, 1, (64, 64, 32))]:
        g = torch.Generator(device="cpu").manual_seed(layout * 100 + trim * 10 + negate)
        ra, ca = ((k, m) if layout & 1 else (m, k))
        rb, cb = ((k, n) if layout & 2 else (n, k))
        a = torch.randn(batch, ra, ca + pad, generator=g).to(dev)
        b = torch.randn(batch, rb, cb + pad, generator=g).to(dev)
        c0 = torch.randn(batch, m, n + pad, generator=g).to(dev)
        out = []
        for plain in (1, 0):
            lib.gq_debug_sgemm_plain(plain)
            c = c0.clone()
            rc = lib.gq_debug_sgemm_ld(p(a), p(b), p(c), m, n, k, layout, trim, negate, acc, lower, None, pad, batch)
            assert rc == 0
            torch.cuda.synchronize()
            out.append(c)
        same = torch.equal(out[0], out[1])
        n_cases += 1
        if not same:
            bad += 1
            d = (out[0] != out[1])
            idx = d.nonzero()[:3].tolist()
            print("DIFF layout", layout, "trim", trim, "neg", negate, "acc", acc, "lower", lower, "pad", pad, "batch", batch, (m, n, k),
                  "count", int(d.sum()), "first", idx)
        if trim == 0 and not lower:
            A = a[..., :ca].double(); B = b[..., :cb].double()
            A = A.transpose(1, 2) if layout & 1 else A
            B = B if layout & 2 else B.transpose(1, 2)
            ref = (-1 if negate else 1) * (A @ B) + (c0[..., :n].double() if acc else 0)
            err = ((out[1][..., :n].double() - ref).abs().max() / ref.abs().max()).item()
            assert err < 1e-5, err
lib.gq_debug_sgemm_plain(0)
print("cases", n_cases, "mismatching", bad)
sys.exit(1 if bad else 0)
