"""Memory-side traffic of the two MFMA kernels the bench reports a roofline for, from rocprofv3 --pmc passes
(FETCH_SIZE and WRITE_SIZE in separate passes: they do not fit one), written as the JSON files bench.py reads:
profiles/gemm_nt_traffic.json and profiles/xtx_traffic.json, each stamped with the SHA-256 of the kernel's source so
that bench.py can tell a measurement of another build from one of this build.
Run on a GPU box from the repository root:  python3 tools/measure_traffic.py <output directory>
(each pass is a child process: `rocprofv3 ... -- python3 tools/pmc_nt.py`)."""
import collections, csv, glob, hashlib, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one_pass(tool, counter, out):
    d = os.path.join(out, "%s_%s" % (os.path.basename(tool).replace(".py", ""), counter))
    cmd = ["timeout", "-s", "KILL", "180", "rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "p", "--",
           "python3", os.path.join(ROOT, "tools", tool)]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
    csv.field_size_limit(1 << 30)
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                acc[(row["Kernel_Name"], int(row["Grid_Size"]) // int(row["Workgroup_Size"]))].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def sha(name):
    return hashlib.sha256(open(os.path.join(ROOT, "gptq_v2_amd", "csrc", name), "rb").read()).hexdigest()


def main():
    out = os.path.abspath(sys.argv[1])
    os.makedirs(out, exist_ok=True)
    correction = ("FETCH_SIZE (KiB) doubled: on gfx950 it tallies the 128-byte requests of 16-byte-per-lane reads at 64 bytes; "
                  "WRITE_SIZE (KiB) taken as reported; Infinity-Cache hits are included in both counters")
    # ---- dense replay GEMM: launch shapes of a Llama-2-7B block at 32768 tokens (tools/pmc_nt.py) ----
    m = 32768
    shapes = {"qkv_12288x4096": (12288, 4096, 1), "o_4096x4096": (4096, 4096, 1), "gate_up_11008x4096": (11008, 4096, 2),
              "down_4096x11008": (4096, 11008, 1)}
    f_nt = {k: v for k, v in one_pass("pmc_nt.py", "FETCH_SIZE", out).items() if "gemm_nt" in k[0]}
    w_nt = {k: v for k, v in one_pass("pmc_nt.py", "WRITE_SIZE", out).items() if "gemm_nt" in k[0]}
    # the persistent grid has the same workgroup count for every shape: launches are told apart by their order
    # in the tool (three per shape), which the per-dispatch rows keep; fall back to the mean if they cannot be
    rec = {"what": "memory-side bytes of one gemm_nt_kernel launch at 32768 tokens, mean over the five launches of a Llama-2-7B block "
                   "(q/k/v as one 12288-row launch, o, gate, up, down), rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate "
                   "passes of tools/pmc_nt.py (3 launches per shape)", "correction": correction, "source": "gemm_nt.hip",
           "source_sha256": sha("gemm_nt.hip")}
    per_dispatch = per_dispatch_rows(out, "pmc_nt")
    if per_dispatch:
        total, n = 0.0, 0
        rec["shapes"] = {}
        for i, (name, (r, c, per_block)) in enumerate(shapes.items()):
            fetch = 2048.0 * mean(per_dispatch["FETCH_SIZE"][3 * i:3 * i + 3])
            write = 1024.0 * mean(per_dispatch["WRITE_SIZE"][3 * i:3 * i + 3])
            rec["shapes"][name] = {"launches_per_block": per_block, "fetch_bytes_corrected_x2": round(fetch), "write_bytes": round(write),
                                   "hbm_bytes_per_launch": round(fetch + write), "algorithmic_bytes": 2 * (m * c + r * c + m * r)}
            total += per_block * (fetch + write)
            n += per_block
        rec["hbm_bytes_per_launch"] = round(total / n)
        rec["algorithmic_bytes_per_launch"] = round(sum(v["launches_per_block"] * v["algorithmic_bytes"] for v in rec["shapes"].values()) / n)
    json.dump(rec, open(os.path.join(out, "gemm_nt_traffic.json"), "w"), indent=1)
    # ---- XtX: C = 4096 (token split) and C = 11008, 32768 tokens (tools/pmc_xtx.py) ----
    f_x = {k: v for k, v in one_pass("pmc_xtx.py", "FETCH_SIZE", out).items() if "xtx256" in k[0]}
    w_x = {k: v for k, v in one_pass("pmc_xtx.py", "WRITE_SIZE", out).items() if "xtx256" in k[0]}
    recx = {"what": "memory-side bytes of one xtx256_kernel launch (32768 tokens), rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in "
                    "separate passes of tools/pmc_xtx.py, mean of 5 launches", "correction": correction, "source": "xtx.hip",
            "source_sha256": sha("xtx.hip")}
    for (name, wgs), v in sorted(f_x.items(), key=lambda kv: kv[0][1]):
        c = 4096 if wgs < 600 else 11008
        w = w_x.get((name, wgs), 0.0)
        recx["c%d" % c] = {"workgroups": wgs, "fetch_bytes_corrected_x2": round(2048.0 * v), "write_bytes": round(1024.0 * w),
                           "hbm_bytes_per_launch": round(2048.0 * v + 1024.0 * w), "algorithmic_bytes": 2 * 32768 * c + 8 * c * c}
    if "c4096" in recx:
        recx["hbm_bytes_per_launch"] = recx["c4096"]["hbm_bytes_per_launch"]
    json.dump(recx, open(os.path.join(out, "xtx_traffic.json"), "w"), indent=1)
    print(json.dumps({"gemm_nt": rec.get("hbm_bytes_per_launch"), "xtx_c4096": recx.get("hbm_bytes_per_launch")}))


def mean(v):
    return sum(v) / max(len(v), 1)


def per_dispatch_rows(out, tool):
    """counter -> values of the gemm_nt launches in dispatch order"""
    res = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        rows = []
        for f in glob.glob(os.path.join(out, "%s_%s" % (tool, counter)) + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                if row["Counter_Name"] == counter and "gemm_nt" in row["Kernel_Name"]:
                    rows.append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
        rows.sort()
        res[counter] = [v for _, v in rows]
    return res if all(len(v) == 12 for v in res.values()) else None


if __name__ == "__main__":
    main()
