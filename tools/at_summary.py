#!/usr/bin/env python3
"""Summarise a VNF_AUTOTUNE_LOG=1 stderr dump: best ring vs best patch configuration per layer."""
import re, sys, collections
d = collections.OrderedDict()
for l in open(sys.argv[1]):
    m = re.match(r'autotune (\S+) cfg (-?\d+): ([\d.]+) ms', l)
    if m:
        d.setdefault(m.group(1), {})[int(m.group(2))] = float(m.group(3))
NR = int(sys.argv[2]) if len(sys.argv) > 2 else 35
tot_r = tot_b = 0
for name, v in d.items():
    ring = {c: t for c, t in v.items() if c < NR}
    patch = {c: t for c, t in v.items() if c >= NR}
    br = min(ring, key=ring.get)
    tot_r += ring[br]
    tot_b += min(v.values())
    if patch:
        bp = min(patch, key=patch.get)
        print('%-26s ring cfg%-3d %.4f  patch cfg%-3d %.4f  | %s' % (name, br, ring[br], bp, patch[bp], ' '.join('%d:%.3f' % (c, t) for c, t in sorted(patch.items()))))
    else:
        print('%-26s ring cfg%-3d %.4f' % (name, br, ring[br]))
print('sum best ring %.4f  sum best overall %.4f' % (tot_r, tot_b))
