"""Kernel resource usage of one .hip unit (VGPRs, scratch, occupancy, LDS): python tools/kres.py attention_fused.hip [hipcc flags]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
         "-Wall", "-Wno-unused-function", "-Rpass-analysis=kernel-resource-usage"]
KEYS = {"VGPRs:": "vgpr", "AGPRs:": "agpr", "ScratchSize [bytes/lane]:": "scratch", "Occupancy [waves/SIMD]:": "occ",
        "LDS Size [bytes/block]:": "lds"}


def main():
    src = os.path.join(ROOT, "openseg3d_amd", "csrc", sys.argv[1])
    out = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + sys.argv[2:] + ["-c", src, "-o", "/tmp/kres.o"],
                         capture_output=True, text=True).stderr
    cur, vals = None, {}
    for line in out.splitlines():
        if "error" in line or "warning:" in line:
            print(line)
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            # kernel name without its namespace and parameter list (parameters may name anonymous-namespace types too)
            cur, vals = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60], {}
            continue
        for key, short in KEYS.items():
            if key in line and "Spill" not in line and "Total" not in line:
                vals[short] = line.split(key)[1].split()[0]
        if "LDS Size" in line and cur:
            print(f"{cur:60s} " + " ".join(f"{k} {vals.get(k)}" for k in KEYS.values()))


if __name__ == "__main__":
    main()
