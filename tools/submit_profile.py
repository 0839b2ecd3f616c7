#!/usr/bin/env python3
"""GPU box: host-side profile of FacePipeline.submit (where does the host block?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vn_celeb_face_recognition_amd import models
from vn_celeb_face_recognition_amd.pipeline import FacePipeline
from vn_celeb_face_recognition_amd.synth import make_frames
dev = torch.device("cuda:0")
NF, PER = 16, 8
frames, _ = make_frames(NF * 2, PER, seed=0)
det = models.MTCNN(keep_all=True, min_face_size=50, device=dev, max_batch=NF, max_height=1080, max_width=1920)
enc = models.InceptionResnetV1(pretrained=None, compute_dtype="bf16", max_batch=256).to(dev).eval()
clf = models.MLPModel(512, 1001).to(dev).eval()
pipe = FacePipeline(det, enc, clf, {"label": list(range(1001)), "name": ["c%d" % i for i in range(1001)]}, 160, 0.0)
batches = [torch.from_numpy(frames[i * NF:(i + 1) * NF]).to(dev) for i in range(2)]
for i in range(5): pipe.submit(batches[i & 1])
torch.cuda.synchronize()
import cProfile, pstats
K = 30
pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable()
for i in range(K): pipe.submit(batches[i & 1])
pr.disable(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host %.3f ms/step, total %.3f ms/step" % ((t1 - t0) / K * 1e3, (t2 - t0) / K * 1e3))
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
