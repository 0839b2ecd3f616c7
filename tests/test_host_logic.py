"""CPU-only tests: host logic, multi-process sharding / all-gather over gloo, C-ABI symbol export."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest
import torch

from conftest import REPO, GOLDEN


def test_library_builds_loads_and_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()  # hipcc cross-compiles gfx950 without a GPU
    from vn_celeb_face_recognition_amd import _lib
    lib = _lib.load()
    hdr = open(os.path.join(REPO, "include", "vnface.h")).read()
    declared = set(re.findall(r"\b(vnf_[a-z0-9_]+)\s*\(", hdr))
    assert {"vnf_init", "vnf_embed", "vnf_mtcnn_detect", "vnf_align", "vnf_classify"} <= declared
    for name in sorted(declared):
        assert hasattr(lib, name), "libvnface.so does not export %s" % name
        assert name in _lib.SIGNATURES, "no ctypes signature for %s" % name
    assert lib.vnf_version().startswith(b"vnface")


def test_models_refuse_cpu_loudly():
    from vn_celeb_face_recognition_amd.models import InceptionResnetV1, MLPModel, MTCNN, iresnet100
    with pytest.raises(RuntimeError, match="MI355X only"):
        InceptionResnetV1(pretrained=None).eval()(torch.zeros(1, 3, 160, 160))
    with pytest.raises(RuntimeError, match="MI355X only"):
        iresnet100(pretrained=False).eval()(torch.zeros(1, 3, 112, 112))
    with pytest.raises(RuntimeError, match="MI355X only"):
        MLPModel(512, 10).eval()(torch.zeros(1, 512))
    with pytest.raises(RuntimeError, match="MI355X only"):
        MTCNN().inference(np.zeros((32, 32, 3), np.uint8))


def test_plugin_registry_and_checkpoint_formats(tmp_path):
    from vn_celeb_face_recognition_amd import models
    from vn_celeb_face_recognition_amd.classifier import load_model_classify
    from vn_celeb_face_recognition_amd.weights import generate_state_dict
    for name in ("InceptionResnetV1", "MLPModel", "MTCNN", "iresnet100", "resnet101", "RetinaFace", "resnet_2branch_50"):
        assert hasattr(models, name)
    with pytest.raises(NotImplementedError):
        models.RetinaFace("cfg_re50")                  # only the mobilenet0.25 configuration is built
    with pytest.raises(NotImplementedError):
        models.resnet101()
    rf = models.RetinaFace("cfg_mnet", device="cpu")
    with pytest.raises(RuntimeError, match="MI355X only"):
        rf.inference([np.zeros((32, 32, 3), np.uint8)])
    # retina_face.py:233-266 checkpoint forms: 'module.' prefixes, optional 'state_dict' level
    rsd = generate_state_dict("retina", 0, as_torch=True)
    rp = str(tmp_path / "mobilenet0.25_Final.pth")
    torch.save({"state_dict": {"module." + k: v for k, v in rsd.items()}}, rp)
    rf2 = models.RetinaFace("cfg_mnet", device="cpu", checkpoint_path=rp)
    assert sorted(rf2._sd) == sorted(rsd) and torch.equal(rf2._sd["ssh3.conv7x7_3.0.weight"], rsd["ssh3.conv7x7_3.0.weight"])
    # trainer/base_trainer.py:91-98 checkpoint dict -> demo_image.py:16-21 loader
    sd = generate_state_dict("mlp", 3, as_torch=True, num_classes=16)
    path = str(tmp_path / "model_best.pth")
    torch.save({"arch": "MLPModel", "epoch": 7, "state_dict": sd, "optimizer": {}, "monitor_best": 0.5, "config": {}}, path)
    m = load_model_classify(path, models.MLPModel(512, 16))
    assert torch.equal(m.state_dict()["dense_2.bias"], sd["dense_2.bias"])
    with pytest.raises(RuntimeError):
        models.MLPModel(512, 17).load_state_dict(sd)
    # IRv1 flat state_dict with the extra logits head of the published files
    irsd = generate_state_dict("irv1", 1, as_torch=True)
    irsd["logits.weight"] = torch.zeros(8631, 512); irsd["logits.bias"] = torch.zeros(8631)
    p2 = str(tmp_path / "vggface2.pt")
    torch.save(irsd, p2)
    enc = models.InceptionResnetV1(pretrained=p2)
    assert torch.equal(enc.state_dict()["last_linear.weight"], irsd["last_linear.weight"])
    with pytest.raises(FileNotFoundError):
        models.InceptionResnetV1(pretrained="vggface2")   # would download in the reference
    # IR-100 {'state_dict': ...}, strict=False (iresnet_encoder.py:171-172)
    p3 = str(tmp_path / "ir100.pth")
    part = {"state_dict": {"fc.bias": torch.ones(512)}}
    torch.save(part, p3)
    ir = models.iresnet100(pretrained=True, checkpoint_path=p3)
    assert torch.equal(torch.as_tensor(ir.state_dict()["fc.bias"]), torch.ones(512))
    with pytest.raises(TypeError):
        models.iresnet100(bogus=1)


def test_mtcnn_input_validation_matches_reference():
    from vn_celeb_face_recognition_amd.models import MTCNN
    det = MTCNN(min_face_size=50, keep_all=True)
    with pytest.raises(Exception, match="equal-dimension"):
        det.inference([np.zeros((40, 40, 3), np.uint8), np.zeros((41, 40, 3), np.uint8)])
    assert det.selection_method == "largest"


def test_shard_helpers():
    from vn_celeb_face_recognition_amd.dist import shard_range, round_robin_batches
    for n in (0, 1, 7, 20, 1800):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert round_robin_batches(10, 1, 4) == [1, 5, 9]


def _gloo_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    sys.path.insert(0, REPO)
    import torch.distributed as dist
    from vn_celeb_face_recognition_amd import dist as vdist
    r, w, _ = vdist.init_from_env("gloo")
    n = 3 + 2 * r                                   # ragged: 3 faces on rank 0, 5 on rank 1
    emb = torch.full((n, 512), float(r)) + torch.arange(n).view(-1, 1)
    parts, counts = vdist.all_gather_embeddings(emb)
    ok = counts == [3, 5] and all(torch.equal(parts[k], torch.full((counts[k], 512), float(k)) + torch.arange(counts[k]).view(-1, 1))
                                  for k in range(w))
    parts0, counts0 = vdist.all_gather_embeddings(torch.zeros((0, 512)) if r == 0 else emb)   # a rank with no faces
    ok = ok and counts0 == [0, 5] and parts0[0].shape == (0, 512)
    out = torch.empty((w * 4, 512))
    vdist.all_gather_fixed(out, torch.full((4, 512), float(r)))
    ok = ok and torch.equal(out[4:], torch.ones(4, 512)) and torch.equal(out[:4], torch.zeros(4, 512))
    lo, hi = vdist.shard_range(20, r, w)
    q.put((r, bool(ok), lo, hi))
    dist.barrier()
    dist.destroy_process_group()


def test_all_gather_embeddings_world_size_2_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29000 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
    assert res == [(0, True, 0, 10), (1, True, 10, 20)]


class _StubTicket:
    def __init__(self, counts, boxes, emb):
        self._r = (counts, boxes, emb, None, None)

    def result(self):
        return self._r


class _StubPipe:
    """FacePipeline stand-in on the CPU: frame i (every pixel == i % 251) "contains" i % 3 faces whose embedding's first
    component is the frame number and whose box encodes (frame, face); the classifier returns that number as the class."""
    label2name = {"label": list(range(1000)), "name": ["n%d" % i for i in range(1000)]}
    threshold = 0.5

    class detector:
        @staticmethod
        def _to_device_frames(q):
            return torch.from_numpy(np.stack(q)), False

    class classifier:
        num_classes = 1000

        @staticmethod
        def classify(emb, want_logp=False):
            return None, emb[:, 0].to(torch.int32), torch.ones(emb.shape[0])

    def __init__(self):
        self.submitted = []

    def submit(self, frames_dev, classify=True):
        assert classify is False
        ids = [int(f[0, 0, 1]) * 251 + int(f[0, 0, 0]) for f in frames_dev]      # frame number stored in two channels
        self.submitted.append(ids)
        counts = [i % 3 for i in ids]
        n = sum(counts)
        emb = torch.zeros((n, 512))
        boxes = np.zeros((n, 4), np.float32)
        o = 0
        for i, c in zip(ids, counts):
            for k in range(c):
                emb[o, 0] = i
                boxes[o] = [i, k, i + 10, k + 20]
                o += 1
        return _StubTicket(counts, boxes, emb)

    def flush(self):
        pass


def _stream_worker(rank, world, port, q, n_total, n_frames, cap=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    sys.path.insert(0, REPO)
    import torch.distributed as dist
    from vn_celeb_face_recognition_amd import dist as vdist
    from vn_celeb_face_recognition_amd.video import FrameSource, run_stream
    vdist.init_from_env("gloo")
    frames = np.zeros((n_total, 4, 6, 3), np.uint8)
    for i in range(n_total):
        frames[i, :, :, 0], frames[i, :, :, 1] = (i + 1) % 251, (i + 1) // 251
    src = FrameSource(frames, 25.0)
    pipe = _StubPipe()
    rows, processed = run_stream(src, pipe, n_frames, rank, world, device="cpu", cap=cap)
    q.put((rank, processed, src.reads, pipe.submitted, "".join(rows[k] for k in sorted(rows)) if rank == 0 else ""))
    dist.barrier()
    dist.destroy_process_group()


def test_video_stream_control_flow_two_ranks_gloo():
    """demo_video's multi-rank path (video.run_stream): every rank submits only its own batches to the throughput
    pipeline, the per-round all-gather carries embeddings + boxes, rank 0 classifies and emits every frame's tracker
    row in frame order -- compared with the single-process run of the same stream."""
    import torch.multiprocessing as mp
    from vn_celeb_face_recognition_amd.video import FrameSource, run_stream
    n_total, n_frames = 23, 4                      # 6 batches: rank 0 gets 0,2,4 -- rank 1 gets 1,3,5 (the short one)
    frames = np.zeros((n_total, 4, 6, 3), np.uint8)
    for i in range(n_total):
        frames[i, :, :, 0], frames[i, :, :, 1] = (i + 1) % 251, (i + 1) // 251
    rows1, p1 = run_stream(FrameSource(frames, 25.0), _StubPipe(), n_frames, 0, 1, device="cpu")
    want = "".join(rows1[k] for k in sorted(rows1))
    assert p1 == n_total and sorted(rows1) == list(range(1, n_total + 1))
    assert rows1[5] == '0.2,"[\'n5\', \'n5\']",5,"[[%s, 0.0, %s, 5.0], [%s, 0.25, %s, 5.25]]"\n' % (5 / 6, 15 / 6, 5 / 6, 15 / 6)
    assert rows1[3] == '0.12,"[]",3,"[]"\n'
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31000 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_stream_worker, args=(r, 2, port, q, n_total, n_frames)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(60)
    (r0, p0, reads0, sub0, text0), (r1, pr1, reads1, sub1, _) = res
    assert (p0, pr1) == (12, 11) and (reads0, reads1) == (12, 11)            # each rank read only its own frames
    assert sub0 == [[1, 2, 3, 4], [9, 10, 11, 12], [17, 18, 19, 20]]
    assert sub1 == [[5, 6, 7, 8], [13, 14, 15, 16], [21, 22, 23]]
    assert text0 == want


def _frames_for(n_total):
    frames = np.zeros((n_total, 4, 6, 3), np.uint8)
    for i in range(n_total):
        frames[i, :, :, 0], frames[i, :, :, 1] = (i + 1) % 251, (i + 1) // 251
    return frames


@pytest.mark.parametrize("n_total,n_frames,cap", [(19, 4, None), (3, 4, None), (23, 4, 2), (40, 4, None)])
def test_video_stream_uneven_rounds_two_ranks_gloo(n_total, n_frames, cap):
    """Collective order when the ranks own different numbers of batches (5 batches: rank 0 has 3 rounds, rank 1 has 2;
    1 batch: rank 1 has none), more rounds than the retire lag (10 batches), and a fixed block too small for a batch
    (cap 2 < faces per batch: the follow-up gather): every rank must issue round 0..R-1's exchange in the same order.
    Rows must equal the single-process run's."""
    import torch.multiprocessing as mp
    from vn_celeb_face_recognition_amd.video import FrameSource, run_stream
    rows1, p1 = run_stream(FrameSource(_frames_for(n_total), 25.0), _StubPipe(), n_frames, 0, 1, device="cpu")
    want = "".join(rows1[k] for k in sorted(rows1))
    assert p1 == n_total and sorted(rows1) == list(range(1, n_total + 1))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33000 + (os.getpid() % 2000) + n_total
    procs = [ctx.Process(target=_stream_worker, args=(r, 2, port, q, n_total, n_frames, cap)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(60)
    assert res[0][1] + res[1][1] == n_total
    assert res[0][4] == want


def test_video_stream_iterator_source_single_process():
    """a decoder-like source (plain iterator): the frame total is only known once it is exhausted"""
    from vn_celeb_face_recognition_amd.video import FrameSource, run_stream
    frames = _frames_for(11)
    rows_a, _ = run_stream(FrameSource(frames, 25.0), _StubPipe(), 4, 0, 1, device="cpu")
    src = FrameSource(iter(list(frames)), 25.0)
    rows_b, pb = run_stream(src, _StubPipe(), 4, 0, 1, device="cpu")
    assert pb == 11 and src.total == 11 and rows_a == rows_b


def test_cli_helpers(tmp_path):
    sys.path.insert(0, REPO)
    import find_embedding as fe
    import demo_video as dv
    from vn_celeb_face_recognition_amd.cli_utils import convert_sec_to_max_time_quantity, read_label2name
    batches, n = fe.create_batch_images(["f%d.png" % i for i in range(20)], 4)
    assert n == 5 and len(batches) == 5 and all(len(b) == 4 for b in batches)   # no empty trailing batch
    batches, n = fe.create_batch_images(["a", "b", "c"], 2)
    assert batches == [["a", "b"], ["c"]]
    img = np.arange(181 * 181 * 3, dtype=np.uint32).reshape(181, 181, 3).astype(np.uint8)
    assert np.array_equal(fe._fit(img, 160), img[10:170, 10:170])
    small = np.full((127, 127, 3), 9, np.uint8)
    fit = fe._fit(small, 160)
    assert fit.shape == (160, 160, 3) and np.array_equal(fit[16:143, 16:143], small) and fit[0, 0, 0] == 0
    row = dv.tracker_row(0.04, 1, ["A", "Unknown"], [np.array([96.0, 54.0, 192.0, 108.0], np.float32)] * 2, (1080, 1920, 3))
    assert row.startswith('0.04,"[\'A\', \'Unknown\']",1,"[[') and row.endswith(']]"\n')
    assert dv.tracker_row(1.0, 25, [], [], (10, 10, 3)) == '1.0,"[]",25,"[]"\n'
    assert convert_sec_to_max_time_quantity(3725.5) == "1.0h:2.0m:5.50s"
    p = tmp_path / "l2n.csv"
    p.write_text("label,name\n233,Ho Ngoc Ha\n135,Hoai Linh\n")
    assert read_label2name(str(p)) == {"label": [233, 135], "name": ["Ho Ngoc Ha", "Hoai Linh"]}


def test_oracle_warp_identity_translation_and_border():
    from oracle import align
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, size=(50, 60, 3), dtype=np.uint8)
    ident = np.array([[1.0, 0, 0], [0, 1.0, 0]])
    assert np.array_equal(align.warp_affine_u8(img, ident, 60, 50), img)
    shift = np.array([[1.0, 0, 7], [0, 1.0, -3]])       # dst(x,y) = src(x-7, y+3)
    out = align.warp_affine_u8(img, shift, 60, 50)
    assert np.array_equal(out[:47, 7:], img[3:, :53]) and (out[:, :7] == 0).all() and (out[47:] == 0).all()
    half = np.array([[1.0, 0, 0.5], [0, 1.0, 0]])       # half-pixel shift: exact 50/50 blend, rounded half up
    out = align.warp_affine_u8(img, half, 60, 50)
    want = (img[:, :-1].astype(np.int64) * 16384 + img[:, 1:].astype(np.int64) * 16384 + 16384) >> 15
    assert np.array_equal(out[:, 1:], want.astype(np.uint8))
    T = align.umeyama(np.array([[0, 0], [1, 0], [0, 1], [1, 1], [0.5, 0.5]]) * 3 + 2, np.array([[0, 0], [1, 0], [0, 1], [1, 1], [0.5, 0.5]]))
    assert np.allclose(T, [[1 / 3, 0, -2 / 3], [0, 1 / 3, -2 / 3], [0, 0, 1]], atol=1e-12)


def test_mjpeg_avi_round_trip_and_export(tmp_path):
    """mjpeg_avi.py: the container the CLIs read / write without OpenCV (demo_video.py:25-43,78-110)."""
    from PIL import Image
    from vn_celeb_face_recognition_amd.cli_utils import export_video_face_recognition, open_frame_source
    from vn_celeb_face_recognition_amd.mjpeg_avi import read_mjpeg_avi, write_mjpeg_avi
    yy, xx = np.mgrid[0:96, 0:128]
    frames = [np.stack([(xx * 2 + 10 * i) % 256, (yy * 2) % 256, (xx + yy) % 256], axis=-1).astype(np.uint8) for i in range(7)]
    p = str(tmp_path / "a.avi")
    assert write_mjpeg_avi(p, frames, 29.97, quality=95) == 7
    fps, gen, n = read_mjpeg_avi(p)
    got = list(gen)
    assert abs(fps - 29.97) < 1e-9 and n == 7 and len(got) == 7 and got[0].shape == (96, 128, 3)
    assert max(np.abs(g.astype(int) - f.astype(int)).mean() for g, f in zip(got, frames)) < 3.0     # JPEG, quality 95
    src = open_frame_source(p)                       # the CLI path: FrameSource over the decoded stream
    assert abs(src.fps - 29.97) < 1e-9
    with pytest.raises(ValueError):
        write_mjpeg_avi(str(tmp_path / "b.avi"), [frames[0], frames[1][:50]], 25)
    bad = tmp_path / "c.avi"
    bad.write_bytes(b"RIFF\x04\x00\x00\x00WAVE")
    with pytest.raises(ValueError):
        read_mjpeg_avi(str(bad))
    with pytest.raises(RuntimeError, match="Motion-JPEG"):
        open_frame_source(str(tmp_path / "movie.mp4"))
    # export: frame_1.png .. frame_N.png -> video (demo_video.py:25-43)
    od = tmp_path / "of"; od.mkdir()
    for i, f in enumerate(frames[:4], start=1):
        Image.fromarray(f).save(od / ("frame_%d.png" % i))
    out = str(tmp_path / "out.avi")
    export_video_face_recognition(str(od), 12.5, out)
    fps2, gen2, n2 = read_mjpeg_avi(out)
    assert fps2 == 12.5 and n2 == 4
    with pytest.raises(RuntimeError, match=".avi"):
        export_video_face_recognition(str(od), 12.5, str(tmp_path / "out.mp4"))
