"""GPU: the CLI surface (config 1 plumbing, demo_image / demo_video) and the resident FacePipeline."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, REPO, load_image, mtcnn_state_dicts

pytestmark = pytest.mark.gpu
PNGS = ["041bc30432964f95871d4c223eba8f7c.png", "318c7ec3b94b451c813a5665cfcfbda3.png", "33f2891da9694198a67aabd1660517c3.png"]


def _run(args, cwd):
    env = dict(os.environ, PYTHONPATH=REPO)
    r = subprocess.run([sys.executable] + args, cwd=cwd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


def _classifier_files(tmp_path, nc=1001):
    from vn_celeb_face_recognition_amd.weights import generate_state_dict
    ck = str(tmp_path / "model_best.pth")
    torch.save({"arch": "MLPModel", "epoch": 3, "state_dict": generate_state_dict("mlp", 0, as_torch=True, num_classes=nc),
                "optimizer": {}, "monitor_best": 0.1, "config": {}}, ck)
    l2n = str(tmp_path / "label2name.csv")
    with open(l2n, "w") as f:
        f.write("label,name\n" + "".join("%d,celeb_%d\n" % (i, i) for i in range(0, nc, 2)))
    return ck, l2n


def test_find_embedding_cli_matches_oracle(tmp_path):
    """BASELINE config 1 plumbing: <stem>.npz with arr_0 (512,) fp32, values == oracle within 1e-4."""
    from oracle import irv1
    from vn_celeb_face_recognition_amd.weights import generate_state_dict
    d = tmp_path / "data"; d.mkdir()
    for f in PNGS:
        shutil.copy(os.path.join(GOLDEN, "images", f), d / f)
    out = tmp_path / "emb"
    stdout = _run([os.path.join(REPO, "find_embedding.py"), "-d", str(d), "-bz", "2", "-o", str(out), "-w", "none", "-dv", "GPU"],
                  str(tmp_path))
    assert stdout.count("Save embedding for") == 3
    sd = generate_state_dict("irv1", 0, as_torch=True)
    for f in PNGS:
        e = np.load(out / (f.split(".")[0] + ".npz"))["arr_0"]
        assert e.shape == (512,) and e.dtype == np.float32
        x = ((np.float32(load_image(f)[10:170, 10:170]) - 127.5) / 128).transpose(2, 0, 1)[None]
        ref = irv1.irv1_forward(sd, torch.from_numpy(x)).numpy()[0]
        assert np.linalg.norm(e - ref) <= 1e-4


def test_face_pipeline_resident_equals_stepwise_and_oracle(tmp_path):
    from vn_celeb_face_recognition_amd import models
    from vn_celeb_face_recognition_amd.classifier import load_model_classify
    from vn_celeb_face_recognition_amd.cli_utils import read_label2name
    from vn_celeb_face_recognition_amd.pipeline import (FacePipeline, center_point_dict, parallel_detect_and_align,
                                                        recognize_celeb, transforms_default)
    from oracle import mtcnn as om, align as oalign, irv1 as oirv1, mlp as omlp
    from vn_celeb_face_recognition_amd.weights import generate_state_dict
    ck, l2n = _classifier_files(tmp_path)
    a = load_image("mrDam_HaHo_recog.jpg")
    frames = [a, np.ascontiguousarray(a[:, ::-1]), np.zeros_like(a)]
    det = models.MTCNN(keep_all=True, min_face_size=50, device="cuda:0", max_batch=3, max_height=a.shape[0], max_width=a.shape[1])
    enc = models.InceptionResnetV1(pretrained=None, compute_dtype="f32", max_batch=16).to("cuda:0").eval()
    clf = load_model_classify(ck, models.MLPModel(512, 1001)).to("cuda:0")
    df = read_label2name(l2n)
    pipe = FacePipeline(det, enc, clf, df, 160, 0.0)
    names, boxes, emb = pipe.recognize_frames(frames)
    faces, chosen = parallel_detect_and_align(frames, det, center_point_dict["(160, 160)"], (160, 160))
    names2 = recognize_celeb(faces, "cuda:0", enc, clf, transforms_default, df, 0.0)
    assert names == names2 and [len(n) for n in names] == [2, 2, 0]
    # oracle chain on the device's boxes/landmarks (boxes differ from CPU ones only in the last bits)
    p, r, o = mtcnn_state_dicts()
    ob, _, ol = om.mtcnn_detect(frames, p, r, o, min_face_size=50, ties="table")
    sd = generate_state_dict("irv1", 0, as_torch=True)
    k = 0
    for i, f in enumerate(frames):
        for b, l in zip(ob[i], ol[i]):
            face = oalign.detect_align_faces(f, [b], [l], oalign.CENTER_POINTS["(160, 160)"], 160, 160)[0]
            e = oirv1.irv1_forward(sd, torch.from_numpy(oalign.transforms_default(face))[None]).numpy()[0]
            assert np.linalg.norm(emb[k].cpu().numpy() - e) <= 2e-2   # a few warp pixels may differ by 1 grey level
            lp = omlp.mlp_forward(generate_state_dict("mlp", 0, as_torch=True), torch.from_numpy(e)[None])
            want, _ = omlp.identify_person(lp, df["label"], df["name"], 0.0)
            assert names[i][k - sum(len(n) for n in names[:i])] == want[0]
            k += 1
    assert k == emb.shape[0] == 4


def test_face_pipeline_submit_overlaps_without_changing_results(tmp_path):
    """FacePipeline.submit (detection and embedding on separate HIP streams, batches in flight together) returns
    exactly what the one-stream embed_frames path returns, for every batch of a back-to-back sequence."""
    from vn_celeb_face_recognition_amd import models
    from vn_celeb_face_recognition_amd.pipeline import FacePipeline
    from vn_celeb_face_recognition_amd.synth import make_frames
    frames, _ = make_frames(6, 3, seed=3, height=360, width=640)
    dev = torch.device("cuda:0")
    det = models.MTCNN(keep_all=True, min_face_size=40, device=dev, max_batch=2, max_height=360, max_width=640)
    enc = models.InceptionResnetV1(pretrained=None, compute_dtype="bf16", max_batch=32).to(dev).eval()
    clf = models.MLPModel(512, 1001).to(dev).eval()
    pipe = FacePipeline(det, enc, clf, {"label": list(range(1001)), "name": ["c%d" % i for i in range(1001)]}, 160, 0.0)
    batches = [torch.from_numpy(frames[i * 2:(i + 1) * 2]).to(dev) for i in range(3)]
    want = []
    for b in batches:
        counts, boxes, emb = pipe.embed_frames(b)
        _, amax, prob = clf.classify(emb, want_logp=False)
        want.append((counts, boxes.copy(), emb.cpu().numpy(), amax.cpu().numpy(), prob.cpu().numpy()))
    assert sum(sum(w[0]) for w in want) > 0
    det2 = models.MTCNN(keep_all=True, min_face_size=40, device=dev, max_batch=2, max_height=360, max_width=640)
    pipe2 = FacePipeline([det, det2], enc, clf, pipe.label2name, 160, 0.0)   # two detection threads + streams
    pipe3 = FacePipeline(det, enc, clf, pipe.label2name, 160, 0.0, embed_batch=8)    # faces of several submits embedded together
    pipe4 = FacePipeline(det, enc, clf, pipe.label2name, 160, 0.0, embed_batch=8, embed_lanes=2)   # ... on rotating streams
    for p in (pipe, pipe2, pipe3, pipe4):
        for rep in range(3):
            tickets = [p.submit(b) for b in batches]          # all three in flight before any result is read
            for t, w in zip(tickets, want):
                counts, boxes, emb, amax, prob = t.result()
                assert counts == w[0] and np.array_equal(boxes, w[1])
                assert np.array_equal(emb.cpu().numpy(), w[2])
                assert np.array_equal(amax.cpu().numpy(), w[3]) and np.array_equal(prob.cpu().numpy(), w[4])
    pipe2.close()
    torch.cuda.synchronize()


def test_demo_image_and_demo_video_cli(tmp_path):
    ck, l2n = _classifier_files(tmp_path)
    a = load_image("mrDam_HaHo_recog.jpg")
    fd = tmp_path / "frames"; fd.mkdir()
    from PIL import Image
    for i in range(5):
        img = a if i != 3 else np.zeros_like(a)
        Image.fromarray(img).save(fd / ("f_%02d.png" % i))
    common = ["-m", ck, "-l2n", l2n, "-enc", "InceptionResnetV1", "-eargs", os.path.join(REPO, "cfg/embedding/inception_resnet_v1.json"),
              "-dargs", os.path.join(REPO, "cfg/detection/mtcnn.json"), "-tg_fs", "160", "--inference_method", "par_fd_vs_aln"]
    out_png = str(tmp_path / "demo_recognition.png")
    so = _run([os.path.join(REPO, "demo_image.py"), "-i", str(fd / "f_00.png"), "-o", out_png] + common, str(tmp_path))
    assert "Face recognized image saved at" in so and os.path.exists(out_png)
    assert "Loading checkpoint" in so and "after training for 3 epochs" in so
    trk = str(tmp_path / "tracker.csv")
    so = _run([os.path.join(REPO, "demo_video.py"), "-i", str(fd), "-o", str(tmp_path / "of"), "-ot", trk, "--n_frames", "2", "-sfr"] + common,
              str(tmp_path))
    assert "FPS for recognition face:" in so and "Saved tracker file in" in so
    lines = open(trk).read().splitlines()
    assert lines[0] == "Time,Names,Frame_idx,Bboxes" and len(lines) == 6
    assert lines[4].endswith(',"[]",4,"[]"')                   # the blank frame
    assert lines[1].split(",")[0] == "0.04" and '"[\'celeb_' in lines[1] or "Unknown" in lines[1]
    assert sorted(os.listdir(tmp_path / "of")) == ["frame_%d.png" % i for i in range(1, 6)]
    # container in, container out (demo_video.py:25-43,78-110) without OpenCV: Motion-JPEG AVI both ways
    from vn_celeb_face_recognition_amd.mjpeg_avi import read_mjpeg_avi, write_mjpeg_avi
    from vn_celeb_face_recognition_amd.cli_utils import read_rgb
    vin = str(tmp_path / "in.avi")
    write_mjpeg_avi(vin, (read_rgb(str(fd / ("f_%02d.png" % i))) for i in range(5)), 25.0, quality=97)
    trk2, vout = str(tmp_path / "tracker2.csv"), str(tmp_path / "out.avi")
    so = _run([os.path.join(REPO, "demo_video.py"), "-i", vin, "-o", str(tmp_path / "of2"), "-ot", trk2, "--n_frames", "2", "-sfr",
               "-ov", vout, "-fps", "25"] + common, str(tmp_path))
    assert "Save exported video in" in so
    lines2 = open(trk2).read().splitlines()
    assert len(lines2) == 6 and lines2[4].endswith(',"[]",4,"[]"')
    assert [l.split(",")[0] for l in lines2[1:]] == [l.split(",")[0] for l in lines[1:]]          # same time stamps (25 fps)
    assert [l.count("celeb_") + l.count("Unknown") for l in lines2[1:]] == [l.count("celeb_") + l.count("Unknown") for l in lines[1:]]
    fps, frames, n = read_mjpeg_avi(vout)
    frames = list(frames)
    assert n == 5 and fps == 25.0 and frames[0].shape == a.shape


def test_celeb_statistic_cli(tmp_path):
    """celeb_statistic.py end to end: -fidx sub-sampling, per-class thresholds file, tracker CSV, interval JSON; and the
    tracker re-use branch (celeb_statistic.py:393-399) reproduces the JSON without touching the GPU."""
    import json
    ck, l2n = _classifier_files(tmp_path)
    a = load_image("mrDam_HaHo_recog.jpg")
    fd = tmp_path / "frames"; fd.mkdir()
    from PIL import Image
    for i in range(12):
        Image.fromarray(a if i % 5 else np.zeros_like(a)).save(fd / ("f_%02d.png" % i))
    thr = tmp_path / "thr.json"
    thr.write_text(json.dumps({str(i): 0.0 for i in range(1001)}))
    common = ["-i", str(fd), "-fps", "4", "-fidx", "1", "3", "-m", ck, "-l2n", l2n, "-enc", "InceptionResnetV1", "-eargs",
              os.path.join(REPO, "cfg/embedding/inception_resnet_v1.json"), "-tg_fs", "160", "--inference_method",
              "par_fd_vs_aln", "-dargs", os.path.join(REPO, "cfg/detection/mtcnn.json"), "--track_bbox",
              "--local_thresholds", str(thr), "-tap", "2", "-nvi", "2", "--n_frames", "2",
              "-o", str(tmp_path / "of"), "-ot", str(tmp_path / "tracker.csv"), "-jst", str(tmp_path / "t.json")]
    out = _run([os.path.join(REPO, "celeb_statistic.py")] + common, str(tmp_path))
    assert "Create tracker file" in out and "Using local thresholds" in out
    rows = open(tmp_path / "tracker.csv").read().splitlines()
    assert rows[0] == "Time,Names,Frame_idx,Bboxes"
    # 12 frames at 4 fps: count % 4 in {1, 3} -> frames 1,3,5,7,9,11
    assert [r.split(",")[-0].split(",")[0] for r in rows[1:]] == ["0.25", "0.75", "1.25", "1.75", "2.25", "2.75"]
    stat = json.load(open(tmp_path / "t.json"))
    assert list(stat) == ["1", "2"] and stat["1"]["interval"][0] == "0.0h:0.0m:0.25s"
    assert sum(len(v) for itv in stat.values() for v in itv["celebrities"].values()) > 0
    first = open(tmp_path / "t.json").read()
    os.remove(tmp_path / "t.json")
    out2 = _run([os.path.join(REPO, "celeb_statistic.py")] + common + ["--statistic_mode", "dynamic_itv"], str(tmp_path))
    assert "Re-use tracker file" in out2 and open(tmp_path / "t.json").read() == first


def test_face_pipeline_with_the_retinaface_detector(tmp_path):
    """The detector is a plugin (demo_image.py:361-366): RetinaFace behind the same pipeline -- resident path == stepwise
    path, and the oracle chain (oracle RetinaFace -> warp -> IRv1 -> MLP) names the same people."""
    from vn_celeb_face_recognition_amd import models
    from vn_celeb_face_recognition_amd.classifier import load_model_classify
    from vn_celeb_face_recognition_amd.cli_utils import read_label2name
    from vn_celeb_face_recognition_amd.pipeline import (FacePipeline, center_point_dict, parallel_detect_and_align,
                                                        recognize_celeb, transforms_default)
    from vn_celeb_face_recognition_amd.synth import make_frames
    from vn_celeb_face_recognition_amd.weights import generate_state_dict
    from oracle import retina as oret, align as oalign, irv1 as oirv1, mlp as omlp
    ck, l2n = _classifier_files(tmp_path)
    frames = list(make_frames(2, 4, seed=1, height=360, width=640)[0])
    kw = dict(keep_top_k=5, vis_thres=0.9)
    det = models.RetinaFace("cfg_mnet", device="cuda:0", max_batch=2, **kw)
    enc = models.InceptionResnetV1(pretrained=None, compute_dtype="f32", max_batch=16).to("cuda:0").eval()
    clf = load_model_classify(ck, models.MLPModel(512, 1001)).to("cuda:0")
    df = read_label2name(l2n)
    pipe = FacePipeline(det, enc, clf, df, 160, 0.0)
    names, boxes, emb = pipe.recognize_frames(frames)
    faces, chosen = parallel_detect_and_align(frames, det, center_point_dict["(160, 160)"], (160, 160))
    names2 = recognize_celeb(faces, "cuda:0", enc, clf, transforms_default, df, 0.0)
    assert names == names2 and [len(n) for n in names] == [5, 5]
    rsd = generate_state_dict("retina", 0, as_torch=True)
    ob, _, ol = oret.inference(rsd, frames, ties="table", **kw)
    sd = generate_state_dict("irv1", 0, as_torch=True)
    msd = generate_state_dict("mlp", 0, as_torch=True)
    k = 0
    for i, f in enumerate(frames):
        for b, l in zip(ob[i], ol[i]):
            face = oalign.detect_align_faces(f, [b], [l], oalign.CENTER_POINTS["(160, 160)"], 160, 160)[0]
            e = oirv1.irv1_forward(sd, torch.from_numpy(oalign.transforms_default(face))[None]).numpy()[0]
            assert np.linalg.norm(emb[k].cpu().numpy() - e) <= 5e-2
            want, _ = omlp.identify_person(omlp.mlp_forward(msd, torch.from_numpy(e)[None]), df["label"], df["name"], 0.0)
            assert names[i][k - sum(len(n) for n in names[:i])] == want[0]
            k += 1
    assert k == 10


def test_demo_video_1080p_npy_stream(tmp_path):
    """BASELINE.json configs[3] on one GPU: a 1080p frame stream (.npy, memory-mapped) through demo_video.py -- one tracker
    row per frame, in order, and the pasted faces found (a box centred inside the paste rectangle: the pasted pictures
    are loose crops, the face fills part of them)."""
    import ast
    import csv
    from vn_celeb_face_recognition_amd.synth import make_frames
    ck, l2n = _classifier_files(tmp_path)
    frames, truth = make_frames(12, 4, seed=5)
    npy = str(tmp_path / "stream.npy")
    np.save(npy, frames)
    trk = str(tmp_path / "tracker.csv")
    common = ["-m", ck, "-l2n", l2n, "-enc", "InceptionResnetV1", "-eargs", os.path.join(REPO, "cfg/embedding/inception_resnet_v1.json"),
              "-dargs", os.path.join(REPO, "cfg/detection/mtcnn.json"), "-tg_fs", "160", "--inference_method", "par_fd_vs_aln"]
    so = _run([os.path.join(REPO, "demo_video.py"), "-i", npy, "-o", str(tmp_path / "of"), "-ot", trk, "--n_frames", "8"] + common,
              str(tmp_path))
    assert "Saved tracker file in" in so
    rows = list(csv.reader(open(trk)))
    assert rows[0] == ["Time", "Names", "Frame_idx", "Bboxes"] and [int(r[2]) for r in rows[1:]] == list(range(1, 13))
    assert abs(float(rows[1][0]) - 1 / 30.0) < 1e-9            # .npy streams count time at 30 fps

    def inside(t, b):
        cx, cy = (b[0] + b[2]) / 2, (b[1] + b[3]) / 2
        return t[0] <= cx <= t[2] and t[1] <= cy <= t[3] and (b[2] - b[0]) * (b[3] - b[1]) >= 0.05 * (t[2] - t[0]) * (t[3] - t[1])
    found = total = 0
    for r, tb in zip(rows[1:], truth):
        boxes = [[b[0] * 1920, b[1] * 1080, b[2] * 1920, b[3] * 1080] for b in ast.literal_eval(r[3])]   # demo_video.py:160-166
        assert len(ast.literal_eval(r[1])) == len(boxes)
        for t in tb:
            total += 1
            found += any(inside(t, b) for b in boxes)
    assert total == 48 and found >= 40, (found, total)


def test_demo_image_cli_with_the_retinaface_plugin(tmp_path):
    """-det RetinaFace -dargs cfg/detection/retina_face.json (the detector the reference's own scripts select,
    scripts/celeb_stat_*.sh): the CLI resolves it by name like the reference (demo_image.py:361-366)."""
    ck, l2n = _classifier_files(tmp_path)
    a = load_image("hoai_linh_4_recog.jpg")
    from PIL import Image
    src = str(tmp_path / "in.png")
    Image.fromarray(a).save(src)
    out_png = str(tmp_path / "out.png")
    so = _run([os.path.join(REPO, "demo_image.py"), "-i", src, "-o", out_png, "-m", ck, "-l2n", l2n, "-enc", "InceptionResnetV1",
               "-eargs", os.path.join(REPO, "cfg/embedding/inception_resnet_v1.json"), "-det", "RetinaFace",
               "-dargs", os.path.join(REPO, "cfg/detection/retina_face.json"), "-tg_fs", "160", "--inference_method", "par_fd_vs_aln"],
              str(tmp_path))
    assert "Face recognized image saved at" in so and os.path.exists(out_png)
