#!/bin/bash
# One GPU-box call: per-kernel time of the MTCNN cascade on 16 synthetic 1080p frames (rocprofv3 kernel trace) and the
# stage table of vnf_mtcnn_stage_times.  Usage: bash tools/profile_detect.sh [tag]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-det}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT -o k --output-format csv -- python3 $ROOT/bench.py --workload detect --steps 50 --warmup 5 --no-cpu-baseline > $OUT/run.log 2>&1 || exit 1
python3 - <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob("$OUT/**/k_kernel_stats.csv", recursive=True)[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open("$OUT/kernel_stats.txt", "w") as f:
    for r in rows[:28]:
        f.write("%-90s calls %6s  avg %9.1f us  total/step %8.1f us  %5.1f %%\n" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3,
                float(r["TotalDurationNs"]) / 55e3, 100 * float(r["TotalDurationNs"]) / tot))
    f.write("all kernels: %.1f us per step (55 steps)\n" % (tot / 55e3))
print(open("$OUT/kernel_stats.txt").read())
PY
cd $ROOT && python3 - <<PY
import torch, sys
sys.path.insert(0, "$ROOT")
from vn_celeb_face_recognition_amd import models
from vn_celeb_face_recognition_amd.synth import make_frames
frames, _ = make_frames(16, 8, seed=0)
det = models.MTCNN(keep_all=True, min_face_size=50, device="cuda:0", max_batch=16, max_height=1080, max_width=1920)
fd = torch.from_numpy(frames).cuda()
st = det.stage_times(fd)
tot = 0
for k, v in st.items():
    print("%-22s %8.4f ms %12d B" % (k, v["ms"], int(v["bytes"]))); tot += v["ms"]
print("sum %.4f ms" % tot)
PY
