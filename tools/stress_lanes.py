#!/usr/bin/env python3
"""GPU box: soak test of the overlap machinery -- every output of many overlapped steps must equal the serial one.
  (a) encoder activation contexts: 3 inputs x 3 lanes x 400 rounds, bit-exact vs one stream / one context;
  (b) FacePipeline.submit (streams, embed micro-batching, 2 detector threads): 60 rounds vs embed_frames."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vn_celeb_face_recognition_amd import models
from vn_celeb_face_recognition_amd.pipeline import FacePipeline
from vn_celeb_face_recognition_amd.synth import make_frames
dev = torch.device("cuda:0")
m = models.InceptionResnetV1(pretrained=None, device=dev, compute_dtype="bf16", max_batch=96).eval()
xs = [torch.randn((96 - 7 * i, 3, 160, 160), generator=torch.Generator().manual_seed(20 + i)).to(dev).to(torch.bfloat16) for i in range(3)]
want = [m(x).clone() for x in xs]
m.set_streams(1); m.set_contexts(3)
lanes = [torch.cuda.Stream(device=dev) for _ in range(3)]
torch.cuda.synchronize()
bad = 0
for rnd in range(400):
    outs = []
    for i, x in enumerate(xs):
        with torch.cuda.stream(lanes[(rnd + i) % 3]):
            outs.append(m(x))
    if rnd % 20 == 19:
        torch.cuda.synchronize()
    for i, o in enumerate(outs):
        with torch.cuda.stream(lanes[(rnd + i) % 3]):
            bad += int((o != want[i]).any().item())
torch.cuda.synchronize()
print("contexts: mismatching outputs:", bad, flush=True)

frames, _ = make_frames(8, 4, seed=9, height=540, width=960)
det = [models.MTCNN(keep_all=True, min_face_size=40, device=dev, max_batch=2, max_height=540, max_width=960) for _ in range(2)]
enc = models.InceptionResnetV1(pretrained=None, compute_dtype="bf16", max_batch=64).to(dev).eval()
clf = models.MLPModel(512, 1001).to(dev).eval()
l2n = {"label": list(range(1001)), "name": ["c%d" % i for i in range(1001)]}
batches = [torch.from_numpy(frames[i * 2:(i + 1) * 2]).to(dev) for i in range(4)]
ref = FacePipeline(det[0], enc, clf, l2n, 160, 0.0)
want = []
for b in batches:
    c, bx, e = ref.embed_frames(b)
    want.append((c, bx.copy(), e.cpu().numpy()))
bad2 = 0
for name, pipe in (("1 det, micro-batch 16", FacePipeline(det[0], enc, clf, l2n, 160, 0.0, embed_batch=16)),
                   ("2 det threads, micro-batch 16, 2 lanes", FacePipeline(det, enc, clf, l2n, 160, 0.0, embed_batch=16, embed_lanes=2)),
                   ("2 det threads, immediate", FacePipeline(det, enc, clf, l2n, 160, 0.0))):
    for rnd in range(20):
        ts = [pipe.submit(b) for b in batches]
        pipe.flush()
        for t, w in zip(ts, want):
            c, bx, e, am, pr = t.result()
            if c != w[0] or not np.array_equal(bx, w[1]) or not np.array_equal(e.cpu().numpy(), w[2]):
                bad2 += 1
    pipe.close()
    print("pipeline [%s]: mismatching batches so far: %d" % (name, bad2), flush=True)
torch.cuda.synchronize()
sys.exit(1 if bad or bad2 else 0)
