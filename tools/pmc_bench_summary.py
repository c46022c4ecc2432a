"""Per-kernel MFMA-busy and CU-busy fractions from one `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES
GRBM_GUI_ACTIVE --output-format csv` pass over a bench run: python tools/pmc_bench_summary.py <dir>.
busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE / 8 XCDs), summed over a kernel's launches."""
import collections, csv, glob, sys
csv.field_size_limit(1 << 30)
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:44]
        key = (name, int(row["Grid_Size"]) // max(int(row["Workgroup_Size"]), 1))
        acc[key][row["Counter_Name"]] += float(row["Counter_Value"])
        if row["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cnt[key] += 1
rows = []
for key, c in acc.items():
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    if gui <= 0:
        continue
    denom = 4 * 256 * gui / 8
    rows.append((gui, key, cnt[key], c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / denom, c.get("SQ_BUSY_CU_CYCLES", 0.0) / (256 * gui / 8) if "SQ_BUSY_CU_CYCLES" in c else float("nan")))
print(f"{'kernel':46s} {'WGs':>6s} {'n':>5s} {'cycles (sum)':>14s} {'MFMA busy':>10s} {'CU busy':>8s}")
for gui, key, n, mf, cu in sorted(rows, reverse=True)[:24]:
    print(f"{key[0]:46s} {key[1]:6d} {n:5d} {gui:14.0f} {mf:10.3f} {cu:8.3f}")
