#!/usr/bin/env python3
"""Per-layer device time of R-Net / O-Net inside one detection of 16 synthetic 1080p frames (VNF_MTCNN_LAYERS=1)."""
import os
import sys

os.environ["VNF_MTCNN_LAYERS"] = "1"
import torch  # noqa: E402

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from vn_celeb_face_recognition_amd import models  # noqa: E402
from vn_celeb_face_recognition_amd.synth import make_frames  # noqa: E402

frames, _ = make_frames(16, 8, seed=0)
det = models.MTCNN(keep_all=True, min_face_size=50, device="cuda:0", max_batch=16, max_height=1080, max_width=1920)
fd = torch.from_numpy(frames).cuda()
det.detect_device(fd)
det.detect_device(fd)
st = det.stage_times(fd, reps=1)
for k, v in st.items():
    print("%-22s %8.4f ms %12d B" % (k, v["ms"], int(v["bytes"])))
