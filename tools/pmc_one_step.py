#!/usr/bin/env python3
"""One warm encoder step for PMC collection (GPU box): rocprofv3 --pmc ... -- python3 tools/pmc_one_step.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vn_celeb_face_recognition_amd.models import InceptionResnetV1
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
m = InceptionResnetV1(pretrained=None, device="cuda:0", compute_dtype="bf16", max_batch=bs).eval()
x = torch.randn((bs, 3, 160, 160), generator=torch.Generator().manual_seed(0)).cuda().to(torch.bfloat16)
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 2
for _ in range(STEPS):
    m(x)
torch.cuda.synchronize()
