"""Five XtX launches at the bench shape (C = 4096, 32768 tokens; token split as HessianHook uses it)
and five at C = 11008, for rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE); nothing else touches the
GPU.  `summarize <dir> <counter>` prints the mean per launch, by workgroup count."""
import collections, csv, glob, os, sys


def run():
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import gptq_v2_amd as amd
    dev = torch.device("cuda:0")
    for c in (4096, 11008):
        x = torch.randn(32768, c, device=dev).half()
        parts = amd.ops.xtx_split_hint(c, x.shape[0])
        h = torch.zeros(parts, c, c, device=dev)
        for _ in range(5):
            if parts > 1:
                amd.ops.xtx_accumulate_split(h, x)
            else:
                amd.ops.xtx_accumulate(h[0], x)
        torch.cuda.synchronize()


def summarize(d, counter):
    csv.field_size_limit(1 << 30)
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "xtx256" in row["Kernel_Name"] and row["Counter_Name"] == counter:
                acc[int(row["Grid_Size"]) // 512].append(float(row["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(counter, "workgroups", k, "n", len(v), "mean", sum(v) / len(v))


if __name__ == "__main__":
    summarize(sys.argv[2], sys.argv[3]) if sys.argv[1:2] == ["summarize"] else run()
