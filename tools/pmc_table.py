#!/usr/bin/env python3
"""Per-kernel table of any rocprofv3 --pmc pass: python tools/pmc_table.py <counter_collection.csv> [min_share]
Sums every counter per kernel name (template arguments kept), prints each counter's per-dispatch mean and, for the SQ
wait counters, its share of SQ_WAVE_CYCLES when that counter is in the pass."""
import csv
import sys
from collections import defaultdict

c = defaultdict(lambda: defaultdict(float))
n = defaultdict(set)
names = []
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "vnf" not in k:
        continue
    k = k.split("(")[0].replace("void vnf::", "")
    cn = r["Counter_Name"]
    if cn not in names:
        names.append(cn)
    c[k][cn] += float(r["Counter_Value"])
    n[k].add(r["Dispatch_Id"])
key = "SQ_WAVE_CYCLES" if "SQ_WAVE_CYCLES" in names else names[0]
tot = sum(v.get(key, 0) for v in c.values()) or 1.0
print("%-64s %6s " % ("kernel", "n") + " ".join("%16s" % x[-16:] for x in names))
for k in sorted(c, key=lambda k: -c[k].get(key, 0)):
    if c[k].get(key, 0) / tot < float(sys.argv[2] if len(sys.argv) > 2 else 0.01):
        continue
    cells = []
    for x in names:
        v = c[k].get(x, 0.0)
        if key == "SQ_WAVE_CYCLES" and x != key and x.startswith("SQ_") and c[k].get(key):
            cells.append("%9.3g (%4.1f%%)" % (v / len(n[k]), 100.0 * v / c[k][key]))
        else:
            cells.append("%16.4g" % (v / len(n[k])))
    print("%-64s %6d " % (k[:64], len(n[k])) + " ".join(cells))
