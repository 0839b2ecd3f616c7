#!/usr/bin/env python3
"""Warm encoder steps for counter / trace collection (GPU box), in bench.py's embed mode (3 lanes / activation contexts):
    rocprofv3 --pmc ... -- python3 tools/pmc_one_step.py [bs] [steps] [dtype] [lanes]
Prints the plan's op list (one line per launch group: "PLAN <launches> <GFLOP per step> <label>") so a kernel trace of
the same run can be keyed by layer (tools/layer_trace.py)."""
import os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vn_celeb_face_recognition_amd.models import InceptionResnetV1
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 2
DT = sys.argv[3] if len(sys.argv) > 3 else "bf16"
LANES = int(sys.argv[4]) if len(sys.argv) > 4 else 3
tdt = {"bf16": torch.bfloat16, "f16": torch.float16}.get(DT, torch.float32)
m = InceptionResnetV1(pretrained=None, device="cuda:0", compute_dtype=DT, max_batch=bs).eval()
x = torch.randn((bs, 3, 160, 160), generator=torch.Generator().manual_seed(0)).cuda().to(tdt)
if os.environ.get("VNF_PRINT_PLAN"):
    m.set_streams(1)
    for line in m.profile(x).splitlines():
        if line.startswith("TOTAL") or not line.strip():
            continue
        mm = re.search(r"([0-9.]+) ms\s+([0-9.]+) GFLOP", line)
        gf = float(mm.group(2)) if mm else 0.0
        label = line[:28].strip() or line.split()[0]
        launches = 5 if ("fused blocks" in line and "one launch per block" in line) else 1   # the stack kernel: one launch
        print("PLAN %d %.3f %s" % (launches, gf, label), flush=True)
        if label.startswith("conv2d_4b"):
            # the ops up to here form the stem group, which the engine runs once per sub-batch (engine.cpp groups:
            # 128 images unfused, the whole batch when conv2d_2a/2b/maxpool are fused)
            fuse = int(os.environ.get("VNF_FUSE", "31"))
            fused_stem = (DT in ("bf16", "f16") and (fuse & 4)) or (DT == "f16x2" and (fuse & 12) == 12)
            chunk = int(os.environ.get("VNF_STEM_CHUNK", "256" if fused_stem else "128"))
            print("GROUP_END %d" % ((bs + chunk - 1) // chunk), flush=True)
    torch.cuda.synchronize()
    print("PLAN_END", flush=True)
m.set_streams(1)
m.set_contexts(LANES)
lanes = [torch.cuda.Stream() for _ in range(LANES)]
for i in range(LANES):          # first use of every context (allocation + tuning) outside the marked region
    with torch.cuda.stream(lanes[i]):
        m(x)
torch.cuda.synchronize()
print("STEPS_BEGIN", flush=True)
for i in range(STEPS):
    with torch.cuda.stream(lanes[i % LANES]):
        m(x)
torch.cuda.synchronize()
