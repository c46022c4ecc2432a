"""Five launches each of the XtX kernel (C = 4096, 11008) and of the W4A16 GEMM (M = 32768; 4096x4096,
11008x4096, 4096x11008) for a rocprofv3 --pmc pass with SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CU_CYCLES and
GRBM_GUI_ACTIVE; `summarize <counter_collection.csv>` prints, per kernel and grid, the share of
SIMD cycles in which the matrix pipe was busy: MFMA_BUSY / (4 SIMDs x 256 CUs x GUI_ACTIVE / 8 XCDs)."""
import collections, csv, os, sys


def run():
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import gptq_v2_amd as amd
    dev = torch.device("cuda:0")
    for c in (4096, 11008):
        x = torch.randn(32768, c, device=dev).half()
        h = torch.zeros(c, c, device=dev)
        for _ in range(5):
            amd.ops.xtx_accumulate(h, x)
        torch.cuda.synchronize()
        del x, h
    for (m, r, c) in ((32768, 4096, 4096), (32768, 11008, 4096), (32768, 4096, 11008)):
        q = torch.randint(0, 16, (r, c), device=dev, dtype=torch.uint8)
        pk = amd.ops.pack_rows(q, 4, amd.ops.PACK_W4_KBLOCK)
        scale = (torch.rand(c // 128, r, device=dev) * 0.01 + 0.002).half()
        zero = torch.randint(0, 16, (c // 128, r), device=dev, dtype=torch.uint8)
        x = torch.randn(m, c, device=dev).half()
        for _ in range(5):
            amd.ops.w4a16_gemm(x, pk, scale, zero, 128)
        torch.cuda.synchronize()


def summarize(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if "xtx256" not in name and "w4a16" not in name:
            continue
        key = (name, int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])))
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            calls[key] += 1
    print(f"{'kernel':24s} {'WGs':>6s} {'calls':>5s} {'cycles/launch':>14s} {'MFMA busy':>10s} {'CU busy':>8s}")
    for key, v in sorted(agg.items()):
        n = max(1, calls[key])
        cyc = v["GRBM_GUI_ACTIVE"] / 8.0 / n
        mfma = v["SQ_VALU_MFMA_BUSY_CYCLES"] / n / (4 * 256 * cyc) if cyc else 0.0
        cu = v.get("SQ_BUSY_CU_CYCLES", 0.0) / n / (256 * cyc) if cyc else 0.0
        print(f"{key[0][:24]:24s} {key[1]:6d} {n:5d} {cyc:14.0f} {mfma:10.3f} {cu:8.3f}")


if __name__ == "__main__":
    summarize(sys.argv[2]) if sys.argv[1:2] == ["summarize"] else run()
