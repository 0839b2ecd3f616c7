#!/usr/bin/env python3
"""GPU box: host time of one encoder forward call (enqueue only) against its device time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vn_celeb_face_recognition_amd import models
dev = torch.device("cuda:0")
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
enc = models.InceptionResnetV1(pretrained=None, compute_dtype="bf16", max_batch=256).to(dev).eval()
enc.set_streams(1)
x = torch.randn((bs, 3, 160, 160), device=dev).to(torch.bfloat16)
for _ in range(5):
    enc(x)
torch.cuda.synchronize()
# idle GPU: one call at a time
ts = []
for _ in range(20):
    t0 = time.perf_counter(); y = enc(x); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append((t1 - t0, t2 - t0))
print("bs %d idle GPU : host enqueue %.3f ms, to completion %.3f ms" % (bs, 1e3 * sorted(t[0] for t in ts)[10], 1e3 * sorted(t[1] for t in ts)[10]))
# back to back (queue fills)
torch.cuda.synchronize(); t0 = time.perf_counter()
hs = []
for _ in range(30):
    a = time.perf_counter(); enc(x); hs.append(time.perf_counter() - a)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("back to back   : host enqueue median %.3f ms (first %.3f), total host %.3f ms, device done after %.3f ms for 30 calls" % (1e3 * sorted(hs)[15], 1e3 * hs[0], 1e3 * (t1 - t0), 1e3 * (t2 - t0)))
