"""The oracle pinned against outputs of the reference itself (tests/golden, tools/make_golden.py)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_image, seeded_normal, mtcnn_state_dicts
from vn_celeb_face_recognition_amd.weights import generate_state_dict, irv1_spec
from oracle import irv1, mlp, iresnet, mtcnn, align


def test_generator_is_deterministic_and_complete():
    a = generate_state_dict("irv1", 0)
    b = generate_state_dict("irv1", 0)
    assert list(a) == [n for n, _, _ in irv1_spec()]
    assert all(np.array_equal(a[k], b[k]) for k in a)
    n_params = sum(v.size for k, v in a.items() if not k.endswith("num_batches_tracked"))
    # 714 entries incl. num_batches_tracked
    assert len(a) == 714
    assert n_params > 23_000_000
    c = generate_state_dict("irv1", 1)
    assert not np.array_equal(a["conv2d_1a.conv.weight"], c["conv2d_1a.conv.weight"])


def test_irv1_oracle_matches_reference():
    g = np.load(os.path.join(GOLDEN, "irv1_seed0.npz"))
    x = seeded_normal((6, 3, 160, 160), g["input_seed"])
    x[4:6] = torch.from_numpy(g["real_inputs"].astype(np.float32))
    sd = generate_state_dict("irv1", 0, as_torch=True)
    taps = {}
    y = irv1.irv1_forward(sd, x, taps=taps).numpy()
    # same torch build, same op sequence: expect bit-equality; tolerance 1e-6 absolute
    assert np.abs(y - g["embeddings"]).max() <= 1e-6
    assert np.abs(taps["last_bn"].numpy() - g["last_bn"]).max() <= 1e-4
    assert np.abs(taps["block8"][:, ::64].numpy() - g["block8_sample"]).max() <= 1e-4
    assert np.allclose(np.linalg.norm(y, axis=1), 1.0, atol=1e-6)


def test_mlp_oracle_matches_reference():
    g = np.load(os.path.join(GOLDEN, "mlp_seed0.npz"))
    e = torch.nn.functional.normalize(seeded_normal((32, 512), g["input_seed"]), dim=1)
    lp = mlp.mlp_forward(generate_state_dict("mlp", 0, as_torch=True), e).numpy()
    assert np.abs(lp - g["logp"]).max() <= 1e-5


def test_identify_person_threshold_and_unknown():
    g = np.load(os.path.join(GOLDEN, "mlp_seed0.npz"))
    lp = torch.from_numpy(g["logp"])
    labels = list(range(0, 1001, 2))            # only even labels have names
    names = ["celeb_%d" % l for l in labels]
    out, pred = mlp.identify_person(lp, labels, names, 0.35)
    p = lp.exp().max(dim=1)[0].numpy()
    am = lp.argmax(dim=1).numpy()
    assert (pred[p < 0.35] == 1001).all() and (pred[p >= 0.35] == am[p >= 0.35]).all()
    assert 0 < (pred == 1001).sum() < len(pred)     # fixture straddles the threshold
    for n, q in zip(out, pred):
        assert n == ("celeb_%d" % q if (q % 2 == 0 and q < 1001) else "Unknown")
    # dict thresholds (celeb_statistic.py:128-136)
    thr = {str(i): (0.0 if i % 2 == 0 else 2.0) for i in range(1001)}
    _, pred2 = mlp.identify_person(lp, labels, names, thr)
    assert ((pred2 == 1001) == (am % 2 == 1)).all()


def test_ir100_oracle_matches_reference():
    g = np.load(os.path.join(GOLDEN, "ir100_seed0.npz"))
    x = seeded_normal((2, 3, 112, 112), g["input_seed"])
    y = iresnet.iresnet_forward(generate_state_dict("iresnet100", 0, as_torch=True), x).numpy()
    assert np.abs(y - g["features"]).max() <= 1e-4 * max(1.0, np.abs(g["features"]).max())


def test_umeyama_matches_skimage_vectors():
    g = np.load(os.path.join(GOLDEN, "align_umeyama.npz"))
    for lm, ref in zip(g["landmarks"], g["skimage_0_18_3_params"]):
        T = align.umeyama(lm, g["template"])
        # scikit-image 0.18.3 computes means/covariance in the input dtype (float32); the
        # restatement is float64: tolerance 2e-4 on the translation column, 2e-6 on the 2x2
        assert np.abs(T[:2, :2] - ref[:2, :2]).max() <= 5e-6
        assert np.abs(T[:2, 2] - ref[:2, 2]).max() <= 5e-4


with open(os.path.join(GOLDEN, "mtcnn_ref.json")) as _f:
    _MT = json.load(_f)


@pytest.mark.parametrize("case", _MT, ids=["%s@%d" % (c["file"], c["min_face_size"]) for c in _MT])
def test_mtcnn_oracle_matches_reference(case):
    g = np.load(os.path.join(GOLDEN, "mtcnn_ref.npz"))
    key = "%s@%d" % (case["file"], case["min_face_size"])
    img = load_image(case["file"])
    assert list(img.shape) == case["shape"]
    p, r, o = mtcnn_state_dicts()
    boxes, probs, points = mtcnn.mtcnn_detect(img, p, r, o, min_face_size=case["min_face_size"])
    assert len(boxes) == case["n"]
    # identical box set and order; coordinates to 1e-3 px (SURVEY.md 8d parity gates)
    assert np.abs(np.asarray(boxes).reshape(-1, 4) - g[key + "/boxes"]).max() <= 1e-3
    assert np.abs(np.asarray(probs).reshape(-1) - g[key + "/probs"]).max() <= 1e-6
    assert np.abs(np.asarray(points).reshape(-1, 5, 2) - g[key + "/points"]).max() <= 1e-3


def test_pnet_level_matches_reference():
    g = np.load(os.path.join(GOLDEN, "mtcnn_ref.npz"))
    img = load_image("mrDam_HaHo_recog.jpg")
    h, w = img.shape[:2]
    oh, ow = mtcnn.level_size(h, w, float(g["pnet_level/scale"]))
    x = torch.from_numpy(img.copy()).permute(2, 0, 1).unsqueeze(0).float()
    lvl = mtcnn.imresample(x, (oh, ow))
    assert np.array_equal(lvl.numpy(), g["pnet_level/level"])
    # explicit numpy restatement of the area bins == torch's kernel, bit for bit
    mine = mtcnn.area_resample(x.numpy()[0], oh, ow)
    assert np.array_equal(mine, g["pnet_level/level"][0])
    p, _, _ = mtcnn_state_dicts()
    with torch.no_grad():
        reg, prob = mtcnn.pnet_forward(p, (lvl - 127.5) * 0.0078125)
    assert np.abs(reg.numpy() - g["pnet_level/reg"]).max() <= 1e-6
    assert np.abs(prob.numpy() - g["pnet_level/prob"]).max() <= 1e-6


def test_area_resample_upsampling_bins():
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, size=(3, 17, 11)).astype(np.float32)
    t = torch.nn.functional.interpolate(torch.from_numpy(a)[None], size=(24, 24), mode="area")[0].numpy()
    assert np.array_equal(mtcnn.area_resample(a, 24, 24), t)
    a = rng.integers(0, 256, size=(3, 131, 97)).astype(np.float32)
    t = torch.nn.functional.interpolate(torch.from_numpy(a)[None], size=(48, 48), mode="area")[0].numpy()
    assert np.array_equal(mtcnn.area_resample(a, 48, 48), t)


# ----------------------------------------------------------------------------- RetinaFace
def _retina_inputs():
    from vn_celeb_face_recognition_amd import synth
    frames, _ = synth.make_frames(n_frames=2, faces_per_frame=4, height=360, width=640, seed=1)
    return frames


def test_retina_oracle_matches_reference():
    """oracle/retina.py against the reference's RetinaFace run in the build container (tools/make_golden.py
    golden_retina): raw heads bit-exact, detections bit-exact under both tie rules (the fixtures hold no tied scores)."""
    import hashlib
    from oracle import retina
    g = np.load(os.path.join(GOLDEN, "retina_ref.npz"))
    with open(os.path.join(GOLDEN, "retina_ref.json")) as f:
        meta = {m["case"]: m for m in json.load(f)}
    sd = generate_state_dict("retina", 0, as_torch=True)
    frames = _retina_inputs()
    assert hashlib.sha1(np.ascontiguousarray(frames).tobytes()).hexdigest() == meta["synth"]["sha1"]
    x = torch.stack([torch.from_numpy((np.float32(f) - retina.CHANNELS_SUBTRACT).transpose(2, 0, 1)) for f in frames]).float()
    loc, conf, ldm = retina.forward(sd, x)
    assert np.array_equal(loc.numpy(), g["synth/loc"])
    assert np.array_equal(conf.numpy(), g["synth/conf"])
    assert np.array_equal(ldm.numpy(), g["synth/ldm"])
    for ties in ("numpy", "table"):
        dets, scores, lms = retina.inference(sd, list(frames), ties=ties)
        for i in range(2):
            assert len(scores[i]) == meta["synth"]["n"][i] > 20
            assert np.array_equal(dets[i], g["synth/%d/boxes" % i])
            assert np.array_equal(scores[i], g["synth/%d/scores" % i])
            assert np.array_equal(lms[i], g["synth/%d/points" % i])
    img = load_image("hoai_linh_4_recog.jpg")
    dets, scores, lms = retina.inference(sd, [img], ties="table")
    assert np.array_equal(dets[0], g["hoai_linh_4_recog.jpg/0/boxes"])
    assert np.array_equal(lms[0], g["hoai_linh_4_recog.jpg/0/points"])
    d2, s2 = retina.inference(sd, [img], landmark=False)
    assert np.array_equal(d2[0], dets[0]) and np.array_equal(s2[0], scores[0])


def test_retina_prior_box_and_decode_small_case():
    """prior_box.py:20-34 / box_utils.py:209-247 on a hand-checkable 32x64 frame."""
    from oracle import retina
    p = retina.prior_box(32, 64)
    assert p.shape == (4 * 8 * 2 + 2 * 4 * 2 + 1 * 2 * 2, 4)
    assert np.allclose(p[0].numpy(), [4 / 64, 4 / 32, 16 / 64, 16 / 32])
    assert np.allclose(p[1].numpy(), [4 / 64, 4 / 32, 32 / 64, 32 / 32])
    assert np.allclose(p[-1].numpy(), [48 / 64, 16 / 32, 512 / 64, 512 / 32])
    z = torch.zeros(p.shape[0], 4)
    b = retina.decode(z, p, [0.1, 0.2])
    assert np.allclose(b[0].numpy(), [4 / 64 - 8 / 64, 4 / 32 - 8 / 32, 4 / 64 + 8 / 64, 4 / 32 + 8 / 32])
    lm = retina.decode_landm(torch.ones(p.shape[0], 10), p, [0.1, 0.2])
    assert np.allclose(lm[0].numpy(), [4 / 64 + 0.1 * 16 / 64, 4 / 32 + 0.1 * 16 / 32] * 5)


def test_retina_nms_tie_rules():
    """py_cpu_nms with tied scores: ties='table' visits equal scores in DEscending position (stable argsort reversed)."""
    from oracle import retina
    dets = np.array([[0, 0, 10, 10, 0.9], [1, 1, 11, 11, 0.9], [50, 50, 60, 60, 0.9], [0, 0, 10, 10, 0.95]], np.float32)
    keep = retina.py_cpu_nms(dets, 0.4, ties="table")
    assert keep == [3, 2]          # 3 first; among the 0.9 ties the highest position (2) leads, 1 and 0 overlap box 3
    dets[3, 4] = 0.5
    assert retina.py_cpu_nms(dets, 0.4, ties="table") == [2, 1]   # 1 suppresses 0, then 3 (IoU with 1 = 0.70)
