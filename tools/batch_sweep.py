#!/usr/bin/env python3
"""GPU box: IRv1 bf16 embeddings/s vs batch size (is bs=256 latency- or throughput-bound?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vn_celeb_face_recognition_amd.models import InceptionResnetV1
for bs in (64, 128, 256, 512, 1024):
    m = InceptionResnetV1(pretrained=None, device="cuda:0", compute_dtype="bf16", max_batch=bs).eval()
    x = torch.randn((bs, 3, 160, 160), generator=torch.Generator().manual_seed(0)).cuda().to(torch.bfloat16)
    for _ in range(5): m(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K = 20
    for _ in range(K): m(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    print("bs %4d: %.3f ms/step  %.0f emb/s" % (bs, dt * 1e3, bs / dt), flush=True)
    del m
