#!/usr/bin/env python3
"""Time the RetinaFace detector on synthetic frames: python tools/retina_time.py [H W B reps]."""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from vn_celeb_face_recognition_amd import synth  # noqa: E402
from vn_celeb_face_recognition_amd.models import RetinaFace  # noqa: E402

H, W, B, reps = (int(a) for a in (sys.argv[1:5] + ["1080", "1920", "8", "10"][len(sys.argv) - 1:]))
frames, _ = synth.make_frames(n_frames=B, faces_per_frame=8, height=H, width=W, seed=3)
fd = torch.from_numpy(frames).cuda()
det = RetinaFace("cfg_mnet", device="cuda:0", max_batch=B)
for _ in range(3):
    counts, *_ = det.detect_device(fd)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(reps):
    det.detect_device(fd)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / reps
print("retina %dx%d batch %d: %.3f ms/call, %.1f frames/s, faces/frame %s" % (H, W, B, dt * 1e3, B / dt, counts[:4]))
