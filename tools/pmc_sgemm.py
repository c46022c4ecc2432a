"""Three launches of the fp32 GEMM per shape (4096^3 in two layouts, the accumulating far update of the column
loop, a near update) for rocprofv3 --pmc passes; `summarize <dir>` prints, per kernel and workgroup count, the
share of the launch's cycles in which the matrix pipe was busy (SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE,
the former summed over the chip's 4 x 256 SIMDs, the latter over its 8 XCDs).
Needs tools/_build/libsgemm_dbg.so (`python tools/bench_sgemm.py build`)."""
import collections, csv, ctypes, glob, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))


def run():
    import torch
    lib = ctypes.CDLL(os.path.join(HERE, "_build", "libsgemm_dbg.so"))
    dev = torch.device("cuda:0")
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    for m, n, k, layout, acc in ((4096, 4096, 4096, 2, 0), (4096, 4096, 4096, 3, 0), (4096, 10496, 512, 3, 1), (4096, 384, 128, 3, 1)):
        a = torch.randn((k, m) if layout & 1 else (m, k), device=dev)
        b = torch.randn((k, n) if layout & 2 else (n, k), device=dev)
        c = torch.zeros(m, n, device=dev)
        for _ in range(3):
            assert lib.gq_debug_sgemm(p(a), p(b), p(c), m, n, k, layout, 0, 0, acc, 0, None) == 0
        torch.cuda.synchronize()


def summarize(d):
    csv.field_size_limit(1 << 30)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "sgemm" in row["Kernel_Name"]:
                key = (row["Kernel_Name"].replace("void (anonymous namespace)::", "").split("(")[0], int(row["Grid_Size"]) // int(row["Workgroup_Size"]))
                acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for key, c in sorted(acc.items()):
        busy, active = c.get("SQ_VALU_MFMA_BUSY_CYCLES"), c.get("GRBM_GUI_ACTIVE")
        if busy and active:
            b, a = sum(busy) / len(busy), sum(active) / len(active)
            print(f"{key[0]:32s} workgroups {key[1]:6d}  launches {len(busy)}  MFMA busy {b:.3e}  GUI active {a:.3e}  busy share {b / (a / 8 * 4 * 256):.3f}")


if __name__ == "__main__":
    summarize(sys.argv[2]) if sys.argv[1:2] == ["summarize"] else run()
