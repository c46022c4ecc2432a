"""GPU idle time inside the phases of a bench step, from a rocprofv3 kernel-trace CSV: a phase is the
window from the end of one marker kernel to the start of the next (markers given as name substrings);
prints wall, the union of kernel run times inside it and the idle remainder.
usage: phase_idle.py <csv> <start-marker> <end-marker>"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
a, b = sys.argv[2], sys.argv[3]
win = None
out = []
for r in rows:
    name = r["Kernel_Name"]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if win is not None and b in name:
        wall = s - win["t0"]
        out.append((wall, win["busy"], win["n"]))
        win = None
    if win is not None:
        lo = max(s, win["cover"])
        if e > lo:
            win["busy"] += e - lo
            win["cover"] = e
        win["n"] += 1
    if a in name:
        win = {"t0": e, "cover": e, "busy": 0, "n": 0}
for wall, busy, n in out:
    print(f"window: {n:5d} kernels, wall {wall / 1e6:8.3f} ms, some kernel running {busy / 1e6:8.3f} ms, idle {(wall - busy) / 1e6:7.3f} ms")
print(f"total wall {sum(o[0] for o in out) / 1e6:.3f} ms, idle {sum(o[0] - o[1] for o in out) / 1e6:.3f} ms over {len(out)} windows")
