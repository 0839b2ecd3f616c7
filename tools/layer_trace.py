#!/usr/bin/env python3
"""Per-layer table of a rocprofv3 --kernel-trace of tools/pmc_one_step.py (bench.py's 3-lane embed mode).

    layer_trace.py <plan.txt (stdout of pmc_one_step.py with VNF_PRINT_PLAN=1)> <kernel_trace.csv> <steps> <lanes> > table

Dispatches of one stream follow the plan order, so the i-th vnf dispatch of a step on a stream is plan launch i; rows
report the average device time of that launch over the traced steps, its algorithmic GFLOP and the achieved TFLOP/s --
measured while the other lanes' kernels share the GPU (that is the benchmarked mode), so the per-row times sum to more
than the step time."""
import csv, sys
from collections import defaultdict

plan = []
for line in open(sys.argv[1]):
    if line.startswith("PLAN "):
        _, n, gf, label = line.rstrip("\n").split(" ", 3)
        for k in range(int(n)):
            plan.append((label + (" #%d" % k if int(n) > 1 else ""), float(gf) / int(n)))
    elif line.startswith("GROUP_END "):
        # the launches listed so far repeat once per sub-batch of the stem group
        rep = int(line.split()[1])
        if rep > 1:
            grp = [(l + " [sub-batch %d/%d]" % (r + 1, rep), g / rep) for r in range(rep) for (l, g) in plan]
            plan = grp
steps, lanes = int(sys.argv[3]), int(sys.argv[4])
rows = [r for r in csv.DictReader(open(sys.argv[2])) if "vnf" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
by_stream = defaultdict(list)
for r in rows:
    by_stream[r.get("Stream_Id", r.get("Queue_Id"))].append(r)
per = len(plan)
acc = defaultdict(list)
names = {}
used = 0
for sid, rs in by_stream.items():
    if len(rs) < per:
        continue
    tail = rs[-(len(rs) // per) * per:] if len(rs) % per else rs     # warm-up dispatches (tuning) come first
    # keep only the last `steps`-worth of whole steps of this stream
    nst = min(len(tail) // per, (steps + lanes - 1) // lanes)
    tail = tail[-nst * per:]
    for i, r in enumerate(tail):
        acc[i % per].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        names[i % per] = r["Kernel_Name"].split("(")[0][:60]
    used += nst
print("# per-launch device time by plan position, %d steps over %d streams (3-lane mode: kernels of other lanes share the GPU)" % (used, len(by_stream)))
print("%-34s %10s %10s %10s  %s" % ("layer", "avg us", "GFLOP", "TFLOP/s", "kernel"))
tot_us = tot_gf = 0.0
for i in range(per):
    if not acc[i]:
        continue
    us = sum(acc[i]) / len(acc[i])
    gf = plan[i][1]
    tot_us += us; tot_gf += gf
    print("%-34s %10.2f %10.2f %10.1f  %s" % (plan[i][0], us, gf, gf / us * 1e3 if us > 0 else 0.0, names[i]))
print("%-34s %10.2f %10.2f %10.1f" % ("SUM of launches (one step)", tot_us, tot_gf, tot_gf / tot_us * 1e3))
