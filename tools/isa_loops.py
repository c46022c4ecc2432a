"""Per-kernel summary of the gfx950 ISA hipcc emits: for every backward branch (a loop), the number of
MFMA, scratch (spill) and VALU instructions between its target and the branch.

usage: python tools/isa_loops.py gptq_v2_amd/csrc/wq_stream.hip [name-filter]"""
import re
import subprocess
import sys
import tempfile


def main():
    src, filt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    with tempfile.NamedTemporaryFile(suffix=".s") as out:
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Iinclude", "-S",
                        "--cuda-device-only", src, "-o", out.name] + sys.argv[3:], check=True, stderr=subprocess.DEVNULL)
        text = open(out.name).read()
    for func in re.split(r"\n(?=_Z\w+:)", text)[1:]:
        lines = func.split("\n")
        name = lines[0].split(":")[0]
        if filt not in name:
            continue
        labels = {m.group(1): i for i, l in enumerate(lines) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
        print(name, len(lines), "lines")
        for i, l in enumerate(lines):
            m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
            if m and labels.get(m.group(1), i) < i:
                body = lines[labels[m.group(1)]:i]
                count = lambda pat: sum(1 for b in body if re.search(pat, b))
                valu = count(r"^\s+v_(?!mfma)")
                print(f"  loop {m.group(1)} [{labels[m.group(1)]}..{i}]: mfma {count('v_mfma')}, scratch_load {count('scratch_load')}, "
                      f"scratch_store {count('scratch_store')}, valu {valu}, ds_read {count('ds_read')}, "
                      f"ds_write {count('ds_write')}, waitcnt {count('s_waitcnt')}")


if __name__ == "__main__":
    main()
