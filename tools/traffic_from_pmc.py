#!/usr/bin/env python3
"""HBM bytes per embed step from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as
MI355X_MICROARCH.md prescribes) of `tools/pmc_one_step.py` run from a tune cache (so every vnf kernel dispatch
belongs to one of its identical steps).  usage: traffic_from_pmc.py <fetch_csv> <write_csv> <steps> <out_json>"""
import csv, json, sys

def total_kib(path, counter):
    tot = conv = 0.0
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter or "vnf" not in r["Kernel_Name"]:
            continue
        v = float(r["Counter_Value"])
        tot += v
        if "conv_" in r["Kernel_Name"]:
            conv += v
    return tot, conv

fetch, fetch_conv = total_kib(sys.argv[1], "FETCH_SIZE")
write, write_conv = total_kib(sys.argv[2], "WRITE_SIZE")
steps = int(sys.argv[3])
out = {
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over python3 tools/pmc_one_step.py "
              "(%d identical embed steps, bs=256 bf16, tile choices from a tune cache so no autotune launches); "
              "all vnf kernels, per step" % steps,
    "fetch_size_kib_per_step": fetch / steps,
    "write_size_kib_per_step": write / steps,
    "fetch_correction": "x2 (gfx950 counts 64 B per 128-B request, MI355X_MICROARCH.md HBM section)",
    "hbm_bytes_per_step": (2.0 * fetch + write) / steps * 1024.0,
    "conv_only_fetch_kib": fetch_conv / steps,
    "conv_only_write_kib": write_conv / steps,
}
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps(out))
