"""One HessianHook.invert at C = 4096 and at C = 11008 (run under `rocprofv3 --kernel-trace`), or
`summarize <kernel_trace.csv>`: per kernel and grid size the launch count, total time and the idle
gaps between consecutive launches (experiment helper)."""
import csv, os, sys, collections


def run():
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import gptq_v2_amd as amd
    dev = torch.device("cuda:0")
    for c in (4096, 11008):
        x = (torch.randn(4096, c, device=dev) * (1 + 3 * torch.arange(c, device=dev) / c)).half()
        for it in range(2):
            hook = amd.HessianHook(); hook.add_batch(x.view(2, 2048, c))
            torch.cuda.synchronize()
            hook.invert(damp_ratio=1e-2, act_order=True)
            torch.cuda.synchronize()


def summarize(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # split into inverts: a scale_kernel launch starts one
    runs, cur = [], None
    for r in rows:
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if "scale_kernel" in name:
            cur = []
            runs.append(cur)
        if cur is not None:
            cur.append((name, int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    for run in runs:
        end = max(i for i, k in enumerate(run) if "flip_out" in k[0]) + 1 if any("flip_out" in k[0] for k in run) else len(run)
        run = run[:end]
        span = run[-1][3] - run[0][2]
        busy = sum(k[3] - k[2] for k in run)
        print(f"--- invert: {len(run)} launches, span {span/1e6:.3f} ms, kernel time {busy/1e6:.3f} ms, idle {100*(span-busy)/span:.0f} %")
        agg = collections.defaultdict(lambda: [0, 0])
        for name, wgs, s, e in run:
            short = name.replace("(anonymous namespace)::", "").replace("void ", "")
            bucket = 1
            while bucket < wgs: bucket *= 4
            key = (short[:40], bucket)
            agg[key][0] += 1; agg[key][1] += e - s
        for (name, bucket), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
            print(f"  {name:40s} <= {bucket:6d} WGs  x{n:4d}  {t/1e6:7.3f} ms  avg {t/n/1e3:7.1f} us")


if __name__ == "__main__":
    summarize(sys.argv[2]) if sys.argv[1:2] == ["summarize"] else run()
