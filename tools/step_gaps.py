"""GPU idle gaps in a rocprofv3 kernel-trace CSV: every interval in which no kernel runs, longer than a
threshold, with the kernels on either side; and the total.  usage: step_gaps.py <csv> [min_us=20] [skip_first_ms=0]"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
min_ns = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 20e3
t_first = int(rows[0]["Start_Timestamp"])
skip = float(sys.argv[3]) * 1e6 if len(sys.argv) > 3 else 0.0
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
cover, last = None, None
gaps = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s - t_first < skip:
        cover, last = max(cover or e, e), r
        continue
    if cover is not None and s > cover:
        gaps.append((s - cover, (cover - t_first) / 1e6, short(last["Kernel_Name"]), short(r["Kernel_Name"])))
    if cover is None or e > cover:
        cover, last = e, r
span = (cover - t_first) - skip
tot = sum(g[0] for g in gaps)
by_pair = {}
for g in gaps:
    k = (g[2], g[3])
    by_pair.setdefault(k, [0, 0])
    by_pair[k][0] += g[0]; by_pair[k][1] += 1
print(f"span {span / 1e6:.1f} ms, idle {tot / 1e6:.2f} ms ({100 * tot / span:.1f} %), {len(gaps)} gaps")
for k, v in sorted(by_pair.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f"  {v[0] / 1e6:8.3f} ms in {v[1]:5d} gaps  after {k[0]:40s} before {k[1]}")
print("longest:")
for g in sorted(gaps, reverse=True)[:15]:
    if g[0] >= min_ns:
        print(f"  {g[0] / 1e3:8.1f} us at {g[1]:9.2f} ms  after {g[2]:40s} before {g[3]}")
