#!/usr/bin/env python3
"""Diagnostic: default (forked) mode vs 3-context mode on one handle, per-stage taps.  usage: diag_modes.py [dtype]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vn_celeb_face_recognition_amd.models import InceptionResnetV1
dt = sys.argv[1] if len(sys.argv) > 1 else "f16x2"
order = sys.argv[2] if len(sys.argv) > 2 else "AB"
dev = torch.device("cuda:0")
x = torch.randn((256, 3, 160, 160), generator=torch.Generator().manual_seed(0)).to(dev)
NAMES = ["conv2d_1a", "conv2d_2a", "conv2d_2b", "maxpool_3a", "conv2d_3b", "conv2d_4a", "conv2d_4b", "repeat_1", "mixed_6a",
         "repeat_2", "mixed_7a", "repeat_3", "block8"]
m = InceptionResnetV1(pretrained=None, device=dev, compute_dtype=dt, max_batch=256).eval()
res = {}
for mode in order:
    if mode == "A":
        m.set_contexts(1); m.set_streams(4)
        y = m(x)
    else:
        m.set_streams(1); m.set_contexts(3)
        lanes = [torch.cuda.Stream(device=dev) for _ in range(3)]
        torch.cuda.synchronize()
        with torch.cuda.stream(lanes[0]):
            y = m(x)
    torch.cuda.synchronize()
    res[mode] = (y.cpu().numpy(), {n: m.tap(n, 256) for n in NAMES})
    print("mode", mode, "emb[0,:4]", res[mode][0][0, :4], "emb[200,:4]", res[mode][0][200, :4])
    print(m.profile(x).replace("\n\n", "\n")[:0])
if len(res) == 2:
    a, b = res["A"], res["B"]
    for n in NAMES:
        d = np.abs(a[1][n] - b[1][n])
        bad = np.nonzero(d.reshape(256, -1).max(axis=1) > 0)[0]
        print("%-12s max|diff| %.3e  images differing: %d %s" % (n, d.max(), len(bad), bad[:12]))
    print("emb max diff", np.abs(a[0] - b[0]).max())
small = InceptionResnetV1(pretrained=None, device=dev, compute_dtype=dt, max_batch=8).eval()
want = torch.cat([small(x[i:i + 8]) for i in range(0, 256, 8)]).cpu().numpy()
for k, v in res.items():
    print("mode", k, "vs serial-8: max diff", np.abs(v[0] - want).max())
