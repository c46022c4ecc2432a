"""Per-kernel, per-grid-size totals from a rocprofv3 kernel trace CSV (experiment helper)."""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2].split(",") if len(sys.argv) > 2 else None
d = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if pat and not any(p in n for p in pat): continue
    short = n.split("(")[0].split("::")[-1][:40]
    key = (short, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]))
    d[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in d.values())
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print(f"{k[0]:42s} grid {k[1]:6d} n {len(v):5d} tot {sum(v)/1e3:8.2f} ms avg {sum(v)/len(v):9.1f} us")
print(f"total {tot/1e3:.2f} ms")
