"""Launch chains in a rocprofv3 kernel-trace CSV: for every run of kernels from `first` to `last`
(kernel-name substrings), the wall time, the time inside kernels, the idle time between them, and the
per-kernel totals.  Usage: chain_gaps.py <csv> <first> <last>"""
import collections
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
first, last = sys.argv[2], sys.argv[3]
chain = None
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if chain is None and first in name:
        chain = {"start": s, "busy": 0, "n": 0, "by": collections.defaultdict(lambda: [0, 0]), "prev_end": s, "gap": 0}
    if chain is not None:
        chain["busy"] += e - s
        chain["n"] += 1
        chain["gap"] += max(0, s - chain["prev_end"])
        chain["prev_end"] = max(chain["prev_end"], e)
        wgs = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
        b = chain["by"][(name[:40], wgs if wgs < 4 else ">=4")]
        b[0] += 1
        b[1] += e - s
        if last in name:
            print(f"chain: {chain['n']} kernels, wall {(e - chain['start']) / 1e6:.3f} ms, in kernels {chain['busy'] / 1e6:.3f} ms, "
                  f"idle between {chain['gap'] / 1e6:.3f} ms")
            for k, (n, t) in sorted(chain["by"].items(), key=lambda kv: -kv[1][1])[:12]:
                print(f"    {k[0]:40s} WGs {str(k[1]):>4s} x{n:5d} {t / 1e6:8.3f} ms  avg {t / n / 1e3:7.1f} us")
            chain = None
