"""One gptq_quant of a 4096x4096 and a 4096x11008 layer (run under `rocprofv3 --kernel-trace`), or
`summarize <kernel_trace.csv>`: per kernel and grid size the launch count and total time of each
layer call (experiment helper)."""
import csv, os, sys, collections


def run():
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import gptq_v2_amd as amd
    dev = torch.device("cuda:0")
    for c in (4096, 11008):
        x = (torch.randn(4096, c, device=dev) * (1 + 3 * torch.arange(c, device=dev) / c)).half()
        w = (torch.randn(4096, c, device=dev) * 0.02).half()
        gs = torch.full([c // 128], 128, dtype=torch.int32); gb = torch.full([c // 128], 4, dtype=torch.int32)
        hook = amd.HessianHook(); hook.add_batch(x.view(2, 2048, c)); hook.invert()
        for it in range(2):
            torch.cuda.synchronize()
            amd.gptq_quant(w, hook, gs, gb, gptq_use_kernel=True, gptq_block_sizes=gs, quant_mse=False, save_device=dev)
            torch.cuda.synchronize()


def summarize(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    runs, cur = [], None
    for r in rows:
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if "prep_kernel" in name:
            cur = []
            runs.append(cur)
        if cur is not None:
            cur.append((name, int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    for run in runs:
        ends = [i for i, k in enumerate(run) if "metrics_flat" in k[0]]
        run = run[:ends[0] + 1] if ends else run
        span = run[-1][3] - run[0][2]
        busy = sum(k[3] - k[2] for k in run)
        print(f"--- layer: {len(run)} launches, span {span/1e6:.3f} ms, kernel time {busy/1e6:.3f} ms, idle {100*(span-busy)/span:.0f} %")
        agg = collections.defaultdict(lambda: [0, 0])
        for name, wgs, s, e in run:
            bucket = 1
            while bucket < wgs: bucket *= 4
            agg[(name[:40], bucket)][0] += 1; agg[(name[:40], bucket)][1] += e - s
        for (name, bucket), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
            print(f"  {name:40s} <= {bucket:6d} WGs  x{n:4d}  {t/1e6:7.3f} ms  avg {t/n/1e3:7.1f} us")


if __name__ == "__main__":
    summarize(sys.argv[2]) if sys.argv[1:2] == ["summarize"] else run()
