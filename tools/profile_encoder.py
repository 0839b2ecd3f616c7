#!/usr/bin/env python3
"""Per-layer device-time table of one encoder step (GPU box): python tools/profile_encoder.py [bs] [dtype]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vn_celeb_face_recognition_amd.models import InceptionResnetV1

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dt = sys.argv[2] if len(sys.argv) > 2 else "bf16"
m = InceptionResnetV1(pretrained=None, device="cuda:0", compute_dtype=dt, max_batch=bs).eval()
x = torch.randn((bs, 3, 160, 160), generator=torch.Generator().manual_seed(0)).cuda()
x = x.to({"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32, "f16x2": torch.float32}[dt])
for _ in range(3):
    m(x)
torch.cuda.synchronize()
print(m.profile(x))
