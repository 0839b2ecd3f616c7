#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS table of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage), to spot
spills and occupancy cliffs before going to the GPU:   python tools/kernel_resources.py csrc/conv_igemm.hip [filter]"""
import os
import re
import subprocess
import sys

FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-fno-gpu-rdc", "-ffp-contract=off", "-x", "hip", "-c"]


def main():
    src, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + [src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"],
                       capture_output=True, text=True)
    name, rec = None, {}
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            rec[name] = {}
        for k in ("VGPRs", "AGPRs", "ScratchSize", "Occupancy", "SGPRs", "LDS Size"):
            m = re.search(re.escape(k) + r"[^:]*: (\d+)", line)
            if m and name:
                rec[name].setdefault(k, int(m.group(1)))
    names = list(rec)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    for n, d in zip(names, dem):
        if flt and flt not in d:
            continue
        v = rec[n]
        print("%-110s vgpr %3d agpr %3d scratch %4d occ %d" % (d[:110], v.get("VGPRs", -1), v.get("AGPRs", -1), v.get("ScratchSize", -1), v.get("Occupancy", -1)))


if __name__ == "__main__":
    main()
