"""Kernel time inside the inversion of tools/bench_invert.py's last C = 11008 (or given C) pass, from a rocprofv3
kernel-trace CSV: window from the last scale_kernel with that grid to the following prep_kernel; per kernel
name and workgroup count: launches, total time; plus the window's wall time and idle.
usage: invert_breakdown.py <csv>"""
import collections, csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
starts = [i for i, r in enumerate(rows) if "scale_kernel" in r["Kernel_Name"]]
for si in starts[-2:]:
    j = next(k for k in range(si, len(rows)) if "prep_kernel" in rows[k]["Kernel_Name"])
    win = rows[si:j]
    t0, t1 = int(win[0]["Start_Timestamp"]), int(win[-1]["End_Timestamp"])
    agg = collections.defaultdict(lambda: [0, 0])
    busy, cover = 0, t0
    for r in win:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        k = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])) * max(1, int(r["Grid_Size_Y"]) // max(1, int(r["Workgroup_Size_Y"]))))
        agg[k][0] += 1; agg[k][1] += e - s
        lo = max(s, cover)
        if e > lo: busy += e - lo; cover = e
    print(f"window of {len(win)} kernels: wall {(t1 - t0) / 1e6:.3f} ms, some kernel running {busy / 1e6:.3f} ms")
    byname = collections.defaultdict(lambda: [0, 0])
    for (n, g), v in agg.items():
        byname[n][0] += v[0]; byname[n][1] += v[1]
    for n, v in sorted(byname.items(), key=lambda kv: -kv[1][1]):
        print(f"  {n:46s} {v[0]:5d} launches {v[1] / 1e6:8.3f} ms")
    print("  largest (kernel, workgroups):")
    for (n, g), v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
        print(f"    {n:44s} {g:6d} WGs {v[0]:5d} launches {v[1] / 1e6:8.3f} ms  avg {v[1] / v[0] / 1e3:7.1f} us")
