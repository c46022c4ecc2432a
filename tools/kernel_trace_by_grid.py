"""Per (kernel, workgroup count) launch count / average / total from a rocprofv3 kernel-trace CSV:
the --stats summary lumps all launches of a kernel together, this separates the XtX launches at
C = 4096 (136 workgroups) from those at C = 11008 (946).  Usage: kernel_trace_by_grid.py <csv> [min_total_ms]"""
import collections, csv, sys

rows = csv.DictReader(open(sys.argv[1]))
floor_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
agg = collections.defaultdict(lambda: [0, 0])
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    wgs = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])) * max(1, int(r.get("Grid_Size_Y", 1)) // max(1, int(r.get("Workgroup_Size_Y", 1))))
    a = agg[(name, wgs)]
    a[0] += 1
    a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print(f"{'kernel':44s} {'WGs':>7s} {'calls':>6s} {'avg us':>10s} {'total ms':>10s}")
for (name, wgs), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if t / 1e6 >= floor_ms:
        print(f"{name[:44]:44s} {wgs:7d} {n:6d} {t / n / 1e3:10.1f} {t / 1e6:10.3f}")
