#!/usr/bin/env python3
"""Per-layer table of a rocprofv3 --kernel-trace of tools/pmc_one_step.py (bench.py's 3-lane embed mode).

    layer_trace.py <plan.txt (stdout of pmc_one_step.py with VNF_PRINT_PLAN=1)> <kernel_trace.csv> <steps> <lanes> > table

Dispatches of one stream follow the plan order.  A step ends with its l2norm dispatch, so every stream's dispatch list is
cut into steps at the l2norm kernels; only steps with exactly the plan's number of launches count (tuning launches at
create time form irregular groups and drop out), and EVERY dispatch must be of the kernel family its plan row names
(stem_conv1a / stem_mid / block35 / block17_trunk / maxpool / avgpool / l2norm / a conv_* kernel) -- a mismatch aborts:
a shifted table is worse than none.  Rows report the average device time of that launch over the traced steps, its
algorithmic GFLOP and the achieved TFLOP/s -- measured while the other lanes' kernels share the GPU (the benchmarked
mode), so the per-row times sum to more than the step time."""
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


def family(label):
    if label.startswith("conv2d_1a (direct"):
        return ("stem_conv1a",)
    if label.startswith("conv2d_2a+"):
        return ("stem_mid",)
    if label.startswith("repeat_1 (fused"):
        return ("block35",)
    if label.startswith("repeat_2 (persistent"):
        return ("block17_trunk",)
    if label.startswith("maxpool"):
        return ("maxpool3s2",)
    if label.startswith("avgpool"):
        return ("avgpool",)
    if label.startswith("l2norm"):
        return ("l2norm",)
    if label.startswith("pack"):
        return ("pack_input",)
    return ("conv_igemm", "conv_patch")     # conv_igemm_dma / conv_igemm_ws / conv_igemm (fallback) / conv_patch


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
    groups, cur = [], []
    for r in rs:
        cur.append(r)
        if "l2norm_kernel" in r["Kernel_Name"]:
            groups.append(cur)
            cur = []
    good = [g for g in groups if len(g) == per]
    good = good[-((steps + lanes - 1) // lanes):]          # the last whole steps of this stream: the traced ones
    for g in good:
        for i, r in enumerate(g):
            kn = r["Kernel_Name"]
            if not any(f in kn for f in family(plan[i][0])):
                sys.exit("layer_trace: dispatch %d of a step on stream %s is %s, the plan row is %r -- refusing to print a shifted table"
                         % (i, sid, kn[:80], plan[i][0]))
            acc[i].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
            names[i] = kn.split("(")[0].replace("void vnf::", "").replace("vnf::", "")[:64]
    used += len(good)
if used == 0:
    sys.exit("layer_trace: no stream holds a whole step of %d launches" % per)
print("# per-launch device time by plan position, %d steps over %d streams (%d-lane mode: kernels of other lanes share the GPU);" % (used, len(by_stream), lanes))
print("# every dispatch checked against its plan row's kernel family")
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
