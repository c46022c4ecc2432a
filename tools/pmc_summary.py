"""Per-kernel mean of one PMC counter from a rocprofv3 --pmc output directory, grouped by grid size:
python tools/pmc_summary.py <dir> <counter> <kernel-name-substring>."""
import csv, glob, sys, collections
d, counter, sub = sys.argv[1], sys.argv[2], sys.argv[3]
csv.field_size_limit(1 << 30)
acc = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if sub in row["Kernel_Name"] and row["Counter_Name"] == counter:
            acc[(row["Grid_Size"], row.get("Workgroup_Size", ""))].append(float(row["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(counter, "grid", k[0], "wg", k[1], "n", len(v), "mean", sum(v) / len(v))
