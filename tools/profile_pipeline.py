#!/usr/bin/env python3
"""Stage times of the resident pipeline on synthetic 1080p frames (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vn_celeb_face_recognition_amd import models
from vn_celeb_face_recognition_amd.pipeline import FacePipeline
from vn_celeb_face_recognition_amd.synth import make_frames

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
frames, truth = make_frames(B, 8)
dev = "cuda:0"
det = models.MTCNN(keep_all=True, min_face_size=50, device=dev, max_batch=B, max_height=1080, max_width=1920)
enc = models.InceptionResnetV1(pretrained=None, compute_dtype="bf16", max_batch=256).to(dev).eval()
clf = models.MLPModel(512, 1001).to(dev).eval()
pipe = FacePipeline(det, enc, clf, {"label": list(range(1001)), "name": ["c%d" % i for i in range(1001)]}, 160, 0.0)
fd = torch.from_numpy(frames).to(dev)
def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e
for it in range(3):
    e0 = ev(); counts, boxes, probs, points = det.detect_device(fd); e1 = ev()
    counts2, boxes2, emb = pipe.embed_frames(fd); e2 = ev()
    _, am, pr = clf.classify(emb, want_logp=False); e3 = ev()
    torch.cuda.synchronize()
    print("iter %d: detect %.2f ms | detect+align+embed %.2f ms | classify %.3f ms | faces %d (pasted %d)" % (
        it, e0.elapsed_time(e1), e1.elapsed_time(e2), e2.elapsed_time(e3), sum(counts), sum(len(t) for t in truth)))
t0 = time.perf_counter()
for _ in range(5):
    counts, boxes, emb = pipe.embed_frames(fd); clf.classify(emb, want_logp=False)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print("steady: %.2f ms per batch of %d frames -> %.1f frames/s, %.1f faces/s" % (dt * 1e3, B, B / dt, sum(counts) / dt))
