#!/usr/bin/env python3
"""3-lane embed step time without any output check (timing experiments): python tools/lane_time.py [lanes] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vn_celeb_face_recognition_amd.models import InceptionResnetV1
from vn_celeb_face_recognition_amd.streams import side_streams

LANES = int(sys.argv[1]) if len(sys.argv) > 1 else 3
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = torch.device("cuda", 0)
m = InceptionResnetV1(pretrained=None, device=dev, compute_dtype="bf16", max_batch=256).eval()
x = torch.randn((256, 3, 160, 160), generator=torch.Generator().manual_seed(0)).to(dev).to(torch.bfloat16)
lanes = side_streams(dev, LANES)
m.set_streams(1); m.set_contexts(LANES)
outs = [torch.empty((256, 512), device=dev) for _ in range(LANES)]
def step(i):
    with torch.cuda.stream(lanes[i % LANES]):
        m.embed_into(x, outs[i % LANES]) if hasattr(m, "embed_into") else outs[i % LANES].copy_(m(x))
for i in range(20):
    step(i)
torch.cuda.synchronize()
t = time.perf_counter()
for i in range(STEPS):
    step(i)
torch.cuda.synchronize()
print("%d lanes: %.4f ms/step" % (LANES, (time.perf_counter() - t) * 1e3 / STEPS))
