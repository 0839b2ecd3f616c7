#!/usr/bin/env python3
"""GPU box: VNF_WS_STAMP=<file> VNF_AUTOTUNE=0 VNF_FORCE_CFG=56 python tools/stamp_ws.py -> in-kernel stamps of the
wave-specialised kernel on the layers it fits (see conv_ws.hip launch_stamped)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vn_celeb_face_recognition_amd.models import InceptionResnetV1
m = InceptionResnetV1(pretrained=None, device="cuda:0", compute_dtype="bf16", max_batch=256).eval()
x = torch.randn((256, 3, 160, 160), generator=torch.Generator().manual_seed(0)).cuda().to(torch.bfloat16)
m(x)
torch.cuda.synchronize()
