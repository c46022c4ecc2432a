"""Three launches of the dense replay GEMM per launch shape of a Llama-2-7B block at the bench's M
(32768 tokens; GQ_PMC_LIB=1: the library GEMM on the same data): merged q/k/v (12288 x 4096), o (4096 x 4096), gate and up (11008 x 4096 each), down
(4096 x 11008); for rocprofv3 --pmc passes.  `summarize <dir> <counter>` prints the per-shape mean."""
import collections, csv, glob, os, sys


def run():
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import gptq_v2_amd as amd
    dev = torch.device("cuda:0")
    m = 32768
    for r, c in ((12288, 4096), (4096, 4096), (11008, 4096), (4096, 11008)):
        x = (torch.randn(m, c, device=dev) * (1 + 3 * torch.arange(c, device=dev) / c)).half()
        w = (torch.randn(r, c, device=dev) * 0.02).half()
        for _ in range(3):
            torch.matmul(x, w.t()) if os.environ.get("GQ_PMC_LIB") else amd.ops.gemm_nt(x, w)
        torch.cuda.synchronize()
        del x, w


def summarize(d, counter):
    csv.field_size_limit(1 << 30)
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if ("gemm_nt" in row["Kernel_Name"] or "Cijk" in row["Kernel_Name"]) and row["Counter_Name"] == counter:
                acc[(row["Kernel_Name"][:24], int(row["Grid_Size"]) // int(row["Workgroup_Size"]))].append(float(row["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(counter, "workgroups", k, "n", len(v), "mean", sum(v) / len(v))


if __name__ == "__main__":
    summarize(sys.argv[2], sys.argv[3]) if sys.argv[1:2] == ["summarize"] else run()
