#!/usr/bin/env python3
"""GPU box: print a digest of the IRv1 embeddings of a fixed batch (compare plans: VNF_FUSE17=0/1, VNF_FORCE_CFG=...)."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vn_celeb_face_recognition_amd.models import InceptionResnetV1
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 37
dt = sys.argv[2] if len(sys.argv) > 2 else "bf16"
m = InceptionResnetV1(pretrained=None, device="cuda:0", compute_dtype=dt, max_batch=max(bs, 64)).eval()
x = torch.randn((bs, 3, 160, 160), generator=torch.Generator().manual_seed(1)).cuda()
e = m(x).cpu().numpy()
print(bs, dt, hashlib.sha256(e.tobytes()).hexdigest()[:16], float(abs(e).sum()))
