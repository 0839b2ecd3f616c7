#!/usr/bin/env python3
"""Per-launch device-time table of one bs-256 forward (vnf_encoder_profile): python tools/run_layers.py [dtype] [model]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vn_celeb_face_recognition_amd.models import InceptionResnetV1, iresnet100

dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
model = sys.argv[2] if len(sys.argv) > 2 else "irv1"
tdt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32, "f16x2": torch.float32}[dt]
size = 160 if model == "irv1" else 112
if model == "irv1":
    m = InceptionResnetV1(pretrained=None, device="cuda:0", compute_dtype=dt, max_batch=256).eval()
else:
    m = iresnet100(pretrained=False, compute_dtype=dt, max_batch=256).to("cuda:0").eval()
m.set_streams(1)
x = torch.randn((256, 3, size, size), generator=torch.Generator().manual_seed(0)).cuda().to(tdt)
for _ in range(3):
    m(x)
torch.cuda.synchronize()
print(m.profile(x))
