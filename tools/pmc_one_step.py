#!/usr/bin/env python3
"""Warm encoder steps for PMC collection (GPU box), in bench.py's embed mode (3 lanes / activation contexts):
rocprofv3 --pmc ... -- python3 tools/pmc_one_step.py [bs] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vn_celeb_face_recognition_amd.models import InceptionResnetV1
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 2
LANES = 3
m = InceptionResnetV1(pretrained=None, device="cuda:0", compute_dtype="bf16", max_batch=bs).eval()
m.set_streams(1)
m.set_contexts(LANES)
x = torch.randn((bs, 3, 160, 160), generator=torch.Generator().manual_seed(0)).cuda().to(torch.bfloat16)
lanes = [torch.cuda.Stream() for _ in range(LANES)]
torch.cuda.synchronize()
for i in range(STEPS):
    with torch.cuda.stream(lanes[i % LANES]):
        m(x)
torch.cuda.synchronize()
