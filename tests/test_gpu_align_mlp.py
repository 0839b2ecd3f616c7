"""GPU parity: MLP classifier (fp32, <=1e-4 on log-probs) and the alignment warp (bit-exact u8)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_image, seeded_normal

pytestmark = pytest.mark.gpu


def test_mlp_matches_reference_golden_and_oracle():
    from vn_celeb_face_recognition_amd.models import MLPModel
    from vn_celeb_face_recognition_amd.weights import generate_state_dict
    from oracle import mlp as omlp
    g = np.load(os.path.join(GOLDEN, "mlp_seed0.npz"))
    e = torch.nn.functional.normalize(seeded_normal((32, 512), g["input_seed"]), dim=1)
    m = MLPModel(512, 1001).to("cuda:0").eval()
    lp = m(e.cuda()).cpu().numpy()
    assert np.abs(lp - g["logp"]).max() <= 1e-4            # vs the reference's own output
    ref = omlp.mlp_forward(generate_state_dict("mlp", 0, as_torch=True), e)
    _, amax, prob = m.classify(e.cuda(), want_logp=False)
    assert np.array_equal(amax.cpu().numpy(), ref.argmax(dim=1).numpy())
    assert np.abs(prob.cpu().numpy() - ref.exp().max(dim=1)[0].numpy()).max() <= 1e-5
    assert np.allclose(np.exp(lp).sum(axis=1), 1.0, atol=1e-5)


@pytest.mark.parametrize("f,c", [(1, 1001), (7, 1000), (300, 1020), (0, 16)])
def test_mlp_shapes_and_class_counts(f, c):
    """ragged batch sizes (incl. empty, > max_batch) and class counts that are not multiples of 8."""
    from vn_celeb_face_recognition_amd.models import MLPModel
    from oracle import mlp as omlp
    m = MLPModel(512, c, max_batch=128).to("cuda:0").eval()
    e = torch.nn.functional.normalize(seeded_normal((f, 512), 5), dim=1)
    lp = m(e.cuda()).cpu()
    assert lp.shape == (f, c)
    if f:
        ref = omlp.mlp_forward(m.state_dict(), e)
        assert (lp - ref).abs().max() <= 1e-4


def test_identify_person_names_match_oracle():
    from vn_celeb_face_recognition_amd.models import MLPModel
    from vn_celeb_face_recognition_amd.pipeline import identify_person
    from oracle import mlp as omlp
    g = np.load(os.path.join(GOLDEN, "mlp_seed0.npz"))
    e = torch.nn.functional.normalize(seeded_normal((32, 512), g["input_seed"]), dim=1)
    m = MLPModel(512, 1001).to("cuda:0").eval()
    labels = list(range(0, 1001, 2))
    names = ["celeb_%d" % l for l in labels]
    df = {"label": labels, "name": names}
    for thr in (0.0, 0.35, {str(i): (0.0 if i % 3 else 2.0) for i in range(1001)}):
        want, _ = omlp.identify_person(torch.from_numpy(g["logp"]), labels, names, thr)
        got = identify_person(e.cuda(), m, df, thr)
        assert got == want


def _random_case(rng, H, W):
    tmpl_key = "(160, 160)"
    from oracle.align import CENTER_POINTS
    t = CENTER_POINTS[tmpl_key]
    ang = rng.uniform(-0.5, 0.5)
    s = rng.uniform(0.5, 2.5)
    R = np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]])
    lm = (t @ R.T) * s + rng.uniform(-20, 60, size=2) + rng.normal(0, 1.5, size=(5, 2))
    return lm.astype(np.float32)


def test_alignment_bit_exact_vs_oracle_random():
    from vn_celeb_face_recognition_amd.pipeline import alignment, center_point_dict
    from oracle import align as oalign
    rng = np.random.default_rng(0)
    for case in range(12):
        H, W = int(rng.integers(40, 400)), int(rng.integers(40, 400))
        img = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
        size = (112, 160)[case % 2]
        key = "(%d, %d)" % (size, size)
        lm = _random_case(rng, H, W) * (size / 160.0)
        want = oalign.alignment(img, oalign.CENTER_POINTS[key], lm, size, size)
        got = alignment(img, center_point_dict[key], lm, size, size, device="cuda:0")
        assert got.shape == want.shape
        assert np.array_equal(got, want), "case %d: %d pixels differ" % (case, (got != want).sum())


def test_detect_align_on_golden_detections_bit_exact():
    """Boxes / landmarks produced by the reference MTCNN on its own pictures -> crop, move landmarks,
    Umeyama, warp: device output == oracle output, every byte; fused normalisation == transforms_default."""
    from vn_celeb_face_recognition_amd.pipeline import align_faces_device, center_point_dict
    from oracle import align as oalign
    g = np.load(os.path.join(GOLDEN, "mtcnn_ref.npz"))
    for key, size in [("mrDam_HaHo_recog.jpg@50", 160), ("dam_vinh_hung_2_recog.jpg@40", 112),
                      ("hoai_linh_4_recog.jpg@50", 160), ("QuangLe_PhuongMyChi_recog.png@30", 112)]:
        img = load_image(key.split("@")[0])
        boxes, points = g[key + "/boxes"], g[key + "/points"]
        tk = "(%d, %d)" % (size, size)
        want = oalign.detect_align_faces(img, boxes, points, oalign.CENTER_POINTS[tk], size, size)
        frames = torch.from_numpy(img.copy()).cuda().unsqueeze(0)
        u8, nm = align_faces_device(frames, np.zeros(len(boxes), np.int32), boxes, points, center_point_dict[tk], size,
                                    norm_dtype=torch.float32)
        u8 = u8.cpu().numpy()
        for k in range(len(boxes)):
            assert np.array_equal(u8[k], want[k]), key
            assert np.array_equal(nm[k].cpu().numpy(), oalign.transforms_default(want[k]))


def test_alignment_box_partly_outside_frame():
    from vn_celeb_face_recognition_amd.pipeline import align_faces_device, center_point_dict
    from oracle import align as oalign
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, size=(120, 150, 3), dtype=np.uint8)
    boxes = np.array([[-12.7, -5.2, 70.3, 88.9], [100.5, 60.1, 170.0, 140.0]], dtype=np.float32)
    t = oalign.CENTER_POINTS["(112, 112)"]
    points = np.stack([t * 0.7 + boxes[0, :2] + 3, t * 0.6 + boxes[1, :2] + 1]).astype(np.float32)
    want = oalign.detect_align_faces(img, boxes, points, t, 112, 112)
    u8, _ = align_faces_device(torch.from_numpy(img).cuda().unsqueeze(0), np.zeros(2, np.int32), boxes, points,
                               center_point_dict["(112, 112)"], 112)
    assert np.array_equal(u8.cpu().numpy(), np.stack(want))
