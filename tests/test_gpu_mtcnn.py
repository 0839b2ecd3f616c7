"""GPU parity of the HIP MTCNN cascade (through the C ABI) against the reference-generated goldens
and the oracle: identical box sets and order, coordinates / landmarks within 1e-3 px."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_image, mtcnn_state_dicts
from oracle import mtcnn as om

pytestmark = pytest.mark.gpu

with open(os.path.join(GOLDEN, "mtcnn_ref.json")) as _f:
    _MT = json.load(_f)


@pytest.mark.parametrize("case", _MT, ids=["%s@%d" % (c["file"], c["min_face_size"]) for c in _MT])
def test_mtcnn_matches_reference_golden(case):
    from vn_celeb_face_recognition_amd.models import MTCNN
    g = np.load(os.path.join(GOLDEN, "mtcnn_ref.npz"))
    key = "%s@%d" % (case["file"], case["min_face_size"])
    img = load_image(case["file"])
    det = MTCNN(image_size=160, keep_all=True, min_face_size=case["min_face_size"], device="cuda:0",
                max_batch=1, max_height=img.shape[0], max_width=img.shape[1]).eval()
    boxes, probs, points = det.inference(img, landmark=True)
    assert len(boxes) == case["n"]
    boxes = np.asarray(boxes).reshape(-1, 4)
    # the oracle with the pinned tie rule ("table"): exact parity, always
    p, r, o = mtcnn_state_dicts()
    st = {}
    ob, op_, ol = om.mtcnn_detect(img, p, r, o, min_face_size=case["min_face_size"], ties="table", stages=st)
    assert np.abs(boxes - np.asarray(ob).reshape(-1, 4)).max() <= 1e-3
    assert np.abs(np.asarray(probs).reshape(-1) - np.asarray(op_).reshape(-1)).max() <= 1e-5
    assert np.abs(np.asarray(points).reshape(-1, 5, 2) - np.asarray(ol).reshape(-1, 5, 2)).max() <= 1e-3
    # the reference's own output.  Its final "Min" NMS visits candidates in np.argsort order -- NumPy's default
    # UNSTABLE sort, so the visiting order of exactly equal O-Net scores (softmax saturates to 1.0f on clear faces) is
    # implementation defined (oracle/mtcnn.py nms_min).  Only faces whose score is one of the tied values may differ
    # from the golden, and then only by being ANOTHER member of the same tied candidate group:
    #   * every face with an untied score: box and landmarks within 1e-3 px of the golden face it overlaps;
    #   * a face with a tied score: IoU >= 0.7 with a golden face of the SAME score, and both it and that golden
    #     face are rows (box and landmarks, 1e-3) of the reference's own pre-NMS candidate table with that score.
    pts = np.asarray(points).reshape(-1, 5, 2)
    prb = np.asarray(probs).reshape(-1)
    gb, gp, gs = g[key + "/boxes"].reshape(-1, 4), g[key + "/points"].reshape(-1, 5, 2), g[key + "/probs"].reshape(-1)
    pre_b, _, pre_p = st["stage3_pre_nms"]
    s3 = pre_b[:, 4]
    vals, cnts = np.unique(s3, return_counts=True)
    tied = set(vals[cnts > 1].tolist())

    def iou(a, b):
        iw = max(0.0, min(a[2], b[2]) - max(a[0], b[0])); ih = max(0.0, min(a[3], b[3]) - max(a[1], b[1]))
        return iw * ih / ((a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - iw * ih)

    def is_candidate(box, lm, score):
        rows = np.nonzero(s3 == score)[0]
        return any(np.abs(pre_b[r, :4] - box).max() <= 1e-3 and np.abs(pre_p[r] - lm).max() <= 1e-3 for r in rows)

    used = set()
    for k in range(len(boxes)):
        j = int(np.argmax([iou(boxes[k], gbj) for gbj in gb]))
        assert j not in used
        used.add(j)
        assert prb[k] == gs[j] or abs(prb[k] - gs[j]) <= 1e-5
        if float(gs[j]) not in tied:
            assert np.abs(boxes[k] - gb[j]).max() <= 1e-3, (k, boxes[k], gb[j])
            assert np.abs(pts[k] - gp[j]).max() <= 1e-3, (k, pts[k], gp[j])
        else:
            assert iou(boxes[k], gb[j]) >= 0.7
            assert is_candidate(boxes[k], pts[k], gs[j]) and is_candidate(gb[j], gp[j], gs[j])
    if not tied:
        assert np.abs(boxes - gb).max() <= 1e-3 and np.abs(pts - gp).max() <= 1e-3     # same order too
    assert np.abs(np.asarray(probs).reshape(-1) - g[key + "/probs"]).max() <= 1e-5


def test_min_nms_tie_rule_on_a_deliberate_exact_tie():
    """detect_face.py:221-257 with exactly tied scores: the HIP kernel visits equal scores in table order (the rule the
    oracle's ties="table" pins; the reference leaves it to np.argsort's unstable sort).  Synthetic single-frame table:
    three overlapping candidates of one face with the SAME score 1.0f (whichever is visited first suppresses the
    others: 'Min' overlap > 0.7), two tied but disjoint candidates, an untied one that a tied one suppresses, one below
    the threshold, and equal AREAS among survivors for the final area ordering (mtcnn.py:334-340)."""
    from vn_celeb_face_recognition_amd.models import MTCNN
    det = MTCNN(keep_all=True, min_face_size=20, device="cuda:0", max_batch=1, max_height=64, max_width=64)
    rng = np.random.default_rng(3)
    boxes = np.array([[100, 100, 200, 200], [104, 98, 203, 199], [97, 103, 198, 204],      # one face, tied 1.0
                      [400, 120, 460, 180], [600, 300, 660, 360],                           # disjoint, tied 0.98, equal areas
                      [402, 122, 458, 178],                                                 # inside box 3, lower score
                      [800, 50, 900, 150], [10, 10, 50, 50]], dtype=np.float32)
    score = np.array([1.0, 1.0, 1.0, 0.98, 0.98, 0.95, 0.5, 0.9], dtype=np.float32)
    oo = np.zeros((len(boxes), 15), dtype=np.float32)
    oo[:, 0] = score
    oo[:, 1:5] = rng.uniform(-0.03, 0.03, size=(len(boxes), 4)).astype(np.float32)
    oo[3, 1:5] = oo[4, 1:5]                                   # equal regression -> exactly equal areas after bbreg
    oo[:, 5:15] = rng.uniform(0.2, 0.8, size=(len(boxes), 10)).astype(np.float32)
    for order in (np.arange(len(boxes)), np.arange(len(boxes))[::-1].copy(), rng.permutation(len(boxes))):
        b, o_ = boxes[order], oo[order]
        gb, gpr, gpt = det.debug_stage3(b, o_)
        # the oracle's O-stage decode (oracle/mtcnn.py detect_face tail) with ties="table"
        keep = o_[:, 0] > np.float32(0.7)
        bk, ok = b[keep], o_[keep]
        w_i = bk[:, 2] - bk[:, 0] + np.float32(1); h_i = bk[:, 3] - bk[:, 1] + np.float32(1)
        px = w_i[:, None] * ok[:, 5:10] + bk[:, 0:1] - np.float32(1)
        py = h_i[:, None] * ok[:, 10:15] + bk[:, 1:2] - np.float32(1)
        pts = np.stack([px, py], axis=2).astype(np.float32)
        reg = om.bbreg(np.concatenate([bk, ok[:, 0:1]], axis=1), ok[:, 1:5])
        pick = om.batched_nms_min(reg[:, :4], reg[:, 4], np.zeros(len(reg), np.int64), 0.7, "table")
        reg, pts = reg[pick], pts[pick]
        fin = np.argsort((reg[:, 2] - reg[:, 0]) * (reg[:, 3] - reg[:, 1]))[::-1]
        assert len(gb) == len(fin) == 4
        assert np.abs(gb - reg[fin, :4]).max() <= 1e-3
        assert np.array_equal(gpr, reg[fin, 4])
        assert np.abs(gpt - pts[fin]).max() <= 1e-3
        # the tied face is represented by its FIRST table row
        first = int(np.nonzero(b[:, 0] < 300)[0][np.argmax(o_[b[:, 0] < 300, 0] == 1.0)])
        assert any(np.abs(gb[k] - om.bbreg(np.concatenate([b[first:first + 1], o_[first:first + 1, 0:1]], axis=1),
                                              o_[first:first + 1, 1:5])[0, :4]).max() <= 1e-3 for k in range(len(gb)))


def test_pnet_level_maps_match_reference():
    """Pyramid level bit-exact (area bins of 8-bit data are exact in fp32); P-Net maps to 1e-5."""
    from vn_celeb_face_recognition_amd.models import MTCNN
    g = np.load(os.path.join(GOLDEN, "mtcnn_ref.npz"))
    img = load_image("mrDam_HaHo_recog.jpg")
    det = MTCNN(min_face_size=50, device="cuda:0", max_batch=1, max_height=img.shape[0], max_width=img.shape[1])
    lvl, prob, reg = det.debug_pnet_level(img, 3)
    want = (g["pnet_level/level"][0] - 127.5) * 0.0078125
    assert lvl.shape == want.shape
    assert np.array_equal(lvl, want.astype(np.float32))
    assert np.abs(prob - g["pnet_level/prob"][0, 1]).max() <= 1e-5
    assert np.abs(reg - g["pnet_level/reg"][0]).max() <= 1e-5


def test_batch_of_frames_equals_per_frame_and_oracle():
    """A batch of equal-size frames (one with no face) gives, per frame, what single calls give, and
    matches the oracle run on the batch."""
    from vn_celeb_face_recognition_amd.models import MTCNN
    from oracle import mtcnn as om
    a = load_image("mrDam_HaHo_recog.jpg")
    h, w = a.shape[:2]
    b = np.ascontiguousarray(a[:, ::-1])                     # mirrored
    rng = np.random.default_rng(0)
    c = rng.integers(0, 40, size=a.shape, dtype=np.uint8)    # dark noise: no faces
    d = np.zeros_like(a); d[60:60 + 181, 200:200 + 181] = load_image("041bc30432964f95871d4c223eba8f7c.png")
    frames = [a, b, c, d]
    det = MTCNN(keep_all=True, min_face_size=40, device="cuda:0", max_batch=4, max_height=h, max_width=w)
    bb, pp, ll = det.inference(frames, landmark=True)
    p, r, o = mtcnn_state_dicts()
    ob, op_, ol = om.mtcnn_detect(frames, p, r, o, min_face_size=40, ties="table")
    for i, f in enumerate(frames):
        sb, sp, sl = det.inference(f, landmark=True)
        assert len(sb) == len(bb[i]) == len(ob[i])
        if len(sb):
            assert np.array_equal(np.asarray(sb), np.asarray(bb[i]))
            assert np.abs(np.asarray(bb[i]) - ob[i]).max() <= 1e-3
            assert np.abs(np.asarray(pp[i]) - op_[i]).max() <= 1e-5
            assert np.abs(np.asarray(ll[i]) - ol[i]).max() <= 1e-3
    assert len(bb[2]) == 0 and bb[2] == []
    assert len(bb[0]) >= 2 and len(bb[3]) >= 1


def test_mixed_sizes_raise_like_the_reference():
    from vn_celeb_face_recognition_amd.models import MTCNN
    det = MTCNN(device="cuda:0", max_batch=2, max_height=64, max_width=64)
    with pytest.raises(Exception, match="equal-dimension"):
        det.inference([np.zeros((40, 40, 3), np.uint8), np.zeros((41, 40, 3), np.uint8)])


def test_tiny_and_blank_images():
    from vn_celeb_face_recognition_amd.models import MTCNN
    det = MTCNN(min_face_size=20, device="cuda:0", max_batch=1, max_height=64, max_width=64)
    for shape in [(8, 8, 3), (12, 12, 3), (30, 17, 3)]:
        boxes, probs = det.inference(np.full(shape, 128, np.uint8), landmark=False)
        assert len(boxes) == 0


def test_detect_then_align_resident_pipeline_matches_oracle():
    """parallel_detect_and_align (demo_image.py:273-306) end to end on the device vs the oracle."""
    from vn_celeb_face_recognition_amd.models import MTCNN
    from vn_celeb_face_recognition_amd.pipeline import parallel_detect_and_align, center_point_dict
    from oracle import mtcnn as om, align as oalign
    img = load_image("dam_vinh_hung_2_recog.jpg")
    det = MTCNN(keep_all=True, min_face_size=40, device="cuda:0", max_batch=1, max_height=img.shape[0], max_width=img.shape[1])
    faces, chosen = parallel_detect_and_align([img], det, center_point_dict["(160, 160)"], (160, 160))
    p, r, o = mtcnn_state_dicts()
    ob, _, ol = om.mtcnn_detect([img], p, r, o, min_face_size=40, ties="table")
    want = oalign.detect_align_faces(img, np.asarray(chosen[0]), ol[0], oalign.CENTER_POINTS["(160, 160)"], 160, 160)
    assert len(faces[0]) == len(want) == 2
    # identical boxes would give identical bytes; device boxes differ from the CPU ones in the last
    # bits (conv summation order), so allow the warp to move by at most a few grey levels on a few pixels
    for f, wnt in zip(faces[0], want):
        diff = np.abs(f.astype(np.int32) - wnt.astype(np.int32))
        assert (diff > 2).mean() < 0.01


def test_pyramid_fast_path_bit_exact_and_synthetic_1080p_batch():
    """Rows of W*3 % 16 == 0 take the vectorised pyramid kernel: levels must stay bit-identical to
    interpolate(mode='area'); and a batch of synthetic 1080p frames finds every pasted face."""
    from vn_celeb_face_recognition_amd.models import MTCNN
    from vn_celeb_face_recognition_amd.synth import make_frames
    img = load_image("hoai_linh_4_recog.jpg")           # 1280 x 720
    h, w = img.shape[:2]
    det = MTCNN(min_face_size=50, device="cuda:0", max_batch=1, max_height=h, max_width=w)
    scales = om.scale_pyramid(h, w, 50, 0.709)
    x = torch.from_numpy(img.copy()).permute(2, 0, 1).unsqueeze(0).float()
    for li in (0, 3, len(scales) - 1):
        lvl, prob, reg = det.debug_pnet_level(img, li)
        want = (om.imresample(x, om.level_size(h, w, scales[li])) - 127.5) * 0.0078125
        assert np.array_equal(lvl, want[0].numpy()), li
    frames, truth = make_frames(2, 6, seed=3)
    det2 = MTCNN(keep_all=True, min_face_size=50, device="cuda:0", max_batch=2)
    boxes, probs = det2.inference(list(frames), landmark=False)
    for b, t in zip(boxes, truth):
        assert len(b) >= len(t)
        for (x0, y0, x1, y1) in t:                       # every pasted crop contains a detection centre
            cx, cy = (np.asarray(b)[:, 0] + np.asarray(b)[:, 2]) / 2, (np.asarray(b)[:, 1] + np.asarray(b)[:, 3]) / 2
            assert ((cx > x0) & (cx < x1) & (cy > y0) & (cy < y1)).any()


def test_result_readback_paths_agree(monkeypatch):
    """The one-copy pinned read-back (<= 32 faces per frame) and the 2-D copy taken by busier frames return the
    same detections (VNF_FIN_FAST lowers the limit so ordinary frames take the second path), and both equal the
    oracle on a 1080p batch: exercises the large-box crop path (row groups split over workgroups) as well."""
    from vn_celeb_face_recognition_amd.models import MTCNN
    from vn_celeb_face_recognition_amd.synth import make_frames
    frames, _ = make_frames(2, 6, seed=5)
    det = MTCNN(keep_all=True, min_face_size=50, device="cuda:0", max_batch=2)
    b1, p1, l1 = det.inference(list(frames), landmark=True)
    monkeypatch.setenv("VNF_FIN_FAST", "1")
    det_slow = MTCNN(keep_all=True, min_face_size=50, device="cuda:0", max_batch=2)
    b2, p2, l2 = det_slow.inference(list(frames), landmark=True)
    assert max(len(b) for b in b1) > 1
    # the device-resident copy of the detections (vnf_mtcnn_results_device) is the host arrays, frame by frame
    n = sum(len(b) for b in b1)
    fidx, bd, pd, ld = (x.cpu().numpy() for x in det.results_device(n))
    assert np.array_equal(fidx, np.repeat(np.arange(2), [len(b) for b in b1]))
    assert np.array_equal(bd, np.concatenate([np.asarray(b).reshape(-1, 4) for b in b1]))
    assert np.array_equal(pd, np.concatenate([np.asarray(p).reshape(-1) for p in p1]))
    assert np.array_equal(ld, np.concatenate([np.asarray(l).reshape(-1, 10) for l in l1]))
    for i in range(2):
        assert np.array_equal(np.asarray(b1[i]), np.asarray(b2[i]))
        assert np.array_equal(np.asarray(p1[i]), np.asarray(p2[i]))
        assert np.array_equal(np.asarray(l1[i]), np.asarray(l2[i]))
    p, r, o = mtcnn_state_dicts()
    ob, op_, ol = om.mtcnn_detect(list(frames), p, r, o, min_face_size=50, ties="table")
    for i in range(2):
        assert len(ob[i]) == len(b1[i])
        assert np.abs(np.asarray(b1[i]).reshape(-1, 4) - np.asarray(ob[i]).reshape(-1, 4)).max() <= 1e-3
        assert np.abs(np.asarray(l1[i]).reshape(-1, 10) - np.asarray(ol[i]).reshape(-1, 10)).max() <= 1e-3


@pytest.mark.parametrize("hw,mfs,thr1", [((233, 317), 20, 0.6), ((301, 403), 30, 0.6), ((360, 641), 40, 0.5), ((487, 353), 24, 0.6)])
def test_random_frame_sizes_match_the_oracle(hw, mfs, thr1):
    """Odd frame sizes (rows that are not a whole number of 16-byte chunks take the per-pixel pyramid / crop paths; odd
    level sizes exercise the ceil-mode pooling edges of the MFMA P-Net front and the R/O-Net front bands), a lowered
    stage-1 threshold (more candidates through the sort / NMS paths), two frames per batch."""
    from PIL import Image
    from vn_celeb_face_recognition_amd.models import MTCNN
    from oracle import mtcnn as om
    h, w = hw
    rng = np.random.default_rng(h * 1000 + w)
    faces = [Image.fromarray(load_image(f)) for f in ("041bc30432964f95871d4c223eba8f7c.png", "318c7ec3b94b451c813a5665cfcfbda3.png",
                                                       "33f2891da9694198a67aabd1660517c3.png")]
    frames = []
    for k in range(2):
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
        img = np.stack([110 + 50 * np.sin(xx / w * (2 + c) + k) * np.cos(yy / h * (1.5 + c)) for c in range(3)], axis=-1)
        img = np.clip(img + rng.normal(0, 6, img.shape), 0, 255).astype(np.uint8)
        for j in range(3):
            s = int(rng.integers(48, min(h, w) // 2))
            x0, y0 = int(rng.integers(0, w - s)), int(rng.integers(0, h - s))
            img[y0:y0 + s, x0:x0 + s] = np.asarray(faces[(k + j) % 3].resize((s, s), Image.BICUBIC))
        frames.append(img)
    thr = [thr1, 0.7, 0.7]
    det = MTCNN(keep_all=True, min_face_size=mfs, thresholds=thr, device="cuda:0", max_batch=2, max_height=h, max_width=w)
    bb, pp, ll = det.inference(frames, landmark=True)
    p, r, o = mtcnn_state_dicts()
    ob, op_, ol = om.mtcnn_detect(frames, p, r, o, min_face_size=mfs, thresholds=thr, ties="table")
    assert sum(len(b) for b in ob) >= 2
    for i in range(2):
        assert len(bb[i]) == len(ob[i]), (i, len(bb[i]), len(ob[i]))
        if len(ob[i]):
            assert np.abs(np.asarray(bb[i]) - ob[i]).max() <= 1e-3
            assert np.abs(np.asarray(pp[i]) - op_[i]).max() <= 1e-5
            assert np.abs(np.asarray(ll[i]) - ol[i]).max() <= 1e-3


def test_candidate_tables_have_no_fixed_cap():
    """The reference has no cap on candidates (detect_face.py:79-93,203-218).  A frame where more than 8192 P-Net cells
    of one pyramid level pass thresholds[0] (more than the LDS sort tables hold: global-memory sort), whose per-scale
    NMS keeps more than 2048 boxes (more than the LDS kept-box tables hold: global-memory kept list) and whose
    cross-scale NMS leaves more rows than the default stage-2 table has (the host layer grows the table and retries)
    must still give the oracle's detections -- no VNF_E_CAPACITY, no truncation.  factor = 0.1 leaves two pyramid
    levels; on a dense grid the 0.5-IoU NMS keeps every other cell in each direction."""
    from vn_celeb_face_recognition_amd.models import MTCNN
    from oracle import mtcnn as om
    rng = np.random.default_rng(3)
    h, w = 400, 560
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.stack([120 + 60 * np.sin(xx / 9.0 + c) * np.cos(yy / 7.0 - c) for c in range(3)], axis=-1)
    img = np.clip(img + rng.normal(0, 25, img.shape), 0, 255).astype(np.uint8)
    probe = MTCNN(min_face_size=20, factor=0.1, device="cuda:0", max_batch=1, max_height=h, max_width=w)
    _, prob, _ = probe.debug_pnet_level(img, 0)
    assert prob.size > 16000
    thr1 = float(np.sort(prob.ravel())[::-1][11000])      # ~11000 cells of level 0 at or above it
    thr = [thr1, 0.7, 0.7]
    p, r, o = mtcnn_state_dicts()
    st = {}
    ob, op_, ol = om.mtcnn_detect(img, p, r, o, min_face_size=20, thresholds=thr, factor=0.1, ties="table", stages=st)
    assert st["n_stage1_raw"] > 8192 and len(st["scales"]) == 2
    n_stage2 = len(st["stage1"][0]) if "stage1" in st else None
    det = MTCNN(keep_all=True, min_face_size=20, factor=0.1, thresholds=thr, device="cuda:0", max_batch=1, max_height=h, max_width=w)
    bb, pp, ll = det.inference(img, landmark=True)
    grown = det._max_candidates
    assert len(bb) == len(ob), (len(bb), len(ob), n_stage2, grown)
    if len(ob):
        assert np.abs(np.asarray(bb) - np.asarray(ob)).max() <= 1e-3
        assert np.abs(np.asarray(pp) - np.asarray(op_)).max() <= 1e-5
    # the same through a handle created large enough: identical, and the second call of the grown detector too
    big = MTCNN(keep_all=True, min_face_size=20, factor=0.1, thresholds=thr, device="cuda:0", max_batch=1, max_height=h, max_width=w,
                max_candidates=32768)
    b2, p2, l2 = big.inference(img, landmark=True)
    b3, p3, l3 = det.inference(img, landmark=True)
    assert np.array_equal(np.asarray(bb), np.asarray(b2)) and np.array_equal(np.asarray(bb), np.asarray(b3))
    assert np.array_equal(np.asarray(ll), np.asarray(l2))
    print("stage-1 raw %d, table rows grown to %d" % (st["n_stage1_raw"], grown))
    assert grown >= 4096, grown          # more than 2048 rows survived stage 1: the default table had to grow


def test_speculative_stage_sizing_is_exact_on_hits_and_misses(monkeypatch):
    """vnf_mtcnn_detect sizes stages 2 / 3 from the previous call's candidate counts and synchronises once
    (detect_face.py:96-146 synchronises at both stage boundaries).  Hit (same frames again), miss (a far busier batch
    after a blank one: the bounds are too small, stages 2 / 3 re-run with exact bounds) and the always-exact path
    (VNF_MTCNN_SPEC=0) must return identical detections, equal to the oracle's."""
    from vn_celeb_face_recognition_amd.models import MTCNN
    from vn_celeb_face_recognition_amd.synth import make_frames
    busy, _ = make_frames(2, 8, seed=11)
    sparse, _ = make_frames(2, 1, seed=12)
    blank = np.full_like(busy, 90)

    def run(det, fr):
        b, p, l = det.inference(list(fr), landmark=True)
        return [np.asarray(x).reshape(-1, 4) for x in b], [np.asarray(x).reshape(-1) for x in p], [np.asarray(x).reshape(-1, 10) for x in l]

    def same(a, b):
        return all(np.array_equal(x, y) for u, v in zip(a, b) for x, y in zip(u, v))

    det = MTCNN(keep_all=True, min_face_size=50, device="cuda:0", max_batch=2)
    first = run(det, busy)                 # first call of this frame size: exact path
    assert sum(len(x) for x in first[0]) >= 12
    assert same(run(det, busy), first)     # hit
    assert same(run(det, busy), first)
    assert all(len(x) == 0 for x in run(det, blank)[0])   # no candidates at all: the bounds shrink to their floor
    assert same(run(det, busy), first)     # miss: far more candidates than the previous call's bounds
    sp = run(det, sparse)                  # over-provisioned bounds: the nets also run on unused tail rows
    assert same(run(det, busy), first) and same(run(det, sparse), sp)
    monkeypatch.setenv("VNF_MTCNN_SPEC", "0")
    exact = MTCNN(keep_all=True, min_face_size=50, device="cuda:0", max_batch=2)
    assert same(run(exact, busy), first) and same(run(exact, sparse), sp) and same(run(exact, busy), first)
    p, r, o = mtcnn_state_dicts()
    ob, _, ol = om.mtcnn_detect(list(busy), p, r, o, min_face_size=50, ties="table")
    for i in range(2):
        assert len(ob[i]) == len(first[0][i])
        assert np.abs(first[0][i] - np.asarray(ob[i]).reshape(-1, 4)).max() <= 1e-3


def test_fused_conv2_pool_kernel_gives_the_plans_detections(monkeypatch):
    """R-Net / O-Net conv2 + PReLU + ceil-mode pool (mtcnn.py:88-90, 142-144) in one kernel per candidate
    (net_mid_kernel: the conv map only in LDS, the plan's split-f16 products in the plan's k order, pooled in fp32 and
    split once) against the plan's conv2 and maxpool_ceil launches (VNF_MTCNN_MID=0), which split the conv map first and
    pool the split values: the two differ by the split format's rounding (2^-22 relative) on the rare element where
    that double rounding is not monotonic, so the detections must be the same faces with boxes within 1e-3 px,
    probabilities within 1e-6 and landmarks within 1e-3 px -- on busy 1080p frames (hundreds of stage-2 candidates per
    frame, ragged last tiles in both nets)."""
    from vn_celeb_face_recognition_amd.models import MTCNN
    from vn_celeb_face_recognition_amd.synth import make_frames
    frames, _ = make_frames(3, 8, seed=21)

    def run(det):
        b, p, l = det.inference(list(frames), landmark=True)
        return [np.asarray(x).reshape(-1, 4) for x in b], [np.asarray(x).reshape(-1) for x in p], [np.asarray(x).reshape(-1, 10) for x in l]

    fused = run(MTCNN(keep_all=True, min_face_size=50, device="cuda:0", max_batch=3))
    monkeypatch.setenv("VNF_MTCNN_MID", "0")
    plain = run(MTCNN(keep_all=True, min_face_size=50, device="cuda:0", max_batch=3))
    assert sum(len(x) for x in fused[0]) >= 18
    for i in range(3):
        assert fused[0][i].shape == plain[0][i].shape
        assert np.abs(fused[0][i] - plain[0][i]).max() <= 1e-3
        assert np.abs(fused[1][i] - plain[1][i]).max() <= 1e-6
        assert np.abs(fused[2][i] - plain[2][i]).max() <= 1e-3
