#!/usr/bin/env python3
"""MFMA-busy summary per kernel from a rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, SQ_INSTS_VALU_MFMA_MOPS_BF16,
GRBM_GUI_ACTIVE, SQ_BUSY_CYCLES) over tools/pmc_one_step.py.

    mfma_busy_from_pmc.py <counter_collection.csv> [kernel_trace.csv] > summary.txt

MfmaUtil follows ROCm's derived-counter definition (counter_defs.yaml): sum(SQ_VALU_MFMA_BUSY_CYCLES) /
(GRBM_GUI_ACTIVE x 1024 SIMDs); rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS
section), hence the /8.  MOPS are 512-FLOP units: executed TFLOP = MOPS * 512."""
import csv, sys
from collections import defaultdict

c = defaultdict(lambda: defaultdict(float))
n = defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "vnf" not in k:
        continue
    k = k.split("(")[0]
    c[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[k].add(r["Dispatch_Id"])
dur = defaultdict(float)
if len(sys.argv) > 2:
    for r in csv.DictReader(open(sys.argv[2])):
        dur[r["Kernel_Name"].split("(")[0]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = defaultdict(float)
print("%-72s %8s %14s %12s %9s %12s" % ("kernel", "launches", "MFMA_BUSY_CYC", "GUI_ACT/8", "MfmaUtil%", "exec TFLOP"))
for k in sorted(c, key=lambda k: -c[k].get("GRBM_GUI_ACTIVE", 0)):
    busy, gui, mops = c[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c[k].get("GRBM_GUI_ACTIVE", 0.0) / 8.0, c[k].get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0)
    util = 100.0 * busy / (gui * 1024.0) if gui else 0.0
    print("%-72s %8d %14.0f %12.0f %9.2f %12.4f" % (k[:72], len(n[k]), busy, gui, util, mops * 512 / 1e12))
    for key, v in (("busy", busy), ("gui", gui), ("mops", mops)):
        tot[key] += v
print("%-72s %8s %14.0f %12.0f %9.2f %12.4f" % ("ALL vnf kernels", "", tot["busy"], tot["gui"], 100.0 * tot["busy"] / (tot["gui"] * 1024.0) if tot["gui"] else 0, tot["mops"] * 512 / 1e12))
