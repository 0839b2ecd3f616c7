#!/usr/bin/env python3
"""GPU box: VNF_PATCH_STAMP=<file> python tools/stamp_patch.py -> in-kernel cycle stamps of the patch kernel's
{256,192,4,2,3} bf16 launches (conv2d_4a), see conv_patch.hip launch_patch_stamped."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vn_celeb_face_recognition_amd.models import InceptionResnetV1
m = InceptionResnetV1(pretrained=None, device="cuda:0", compute_dtype="bf16", max_batch=256).eval()
m.set_streams(1)
x = torch.randn((256, 3, 160, 160), generator=torch.Generator().manual_seed(0)).cuda().to(torch.bfloat16)
for _ in range(3):
    m(x)
torch.cuda.synchronize()
