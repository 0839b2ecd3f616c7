"""CPU oracle: a restatement of the reference's detect -> align -> embed -> classify path.

TEST INFRASTRUCTURE ONLY.  Nothing under the product package imports this; only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may call it, and only as the checker.

Every function cites the /root/reference file:line it restates.  The floating-point nets are
written with plain torch-CPU fp32 functional ops in the reference's op order (that is what the
reference itself executes on CPU); box arithmetic, NMS, crop/resize bins, Umeyama and the affine
warp are written out in numpy.

Pinning: the reference has no tests or golden vectors.  The oracle is pinned by
outputs of the reference itself, produced in the build container by tools/make_golden.py
(imports /root/reference with an in-memory torchvision shim for batched_nms) and committed
under tests/golden/.  Two third-party boundaries stay "parity unpinned" because the libraries
are absent offline: cv2.warpAffine (align_face.py:55) and torchvision.ops.batched_nms
(detect_face.py:79,93,128) -- both are restated from their documented algorithms
(oracle/align.py, oracle/mtcnn.py headers).
"""
