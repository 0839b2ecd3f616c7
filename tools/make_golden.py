#!/usr/bin/env python3
"""Generate tests/golden/* by RUNNING THE REFERENCE in the build container.

Usage (build container only; /root/reference does not exist on the GPU box):
    PYTHONDONTWRITEBYTECODE=1 python tools/make_golden.py

What it does
  * imports the reference's model modules from /root/reference WITHOUT executing
    models/__init__.py (that file imports torchvision eagerly) by registering an
    empty package object whose __path__ points at /root/reference/models;
  * installs an in-memory torchvision shim (our own pure-torch code) for the three symbols the
    hot path needs: ops.boxes.batched_nms, transforms.functional.to_tensor and
    models.utils.load_state_dict_from_url (never called);
  * loads this repo's deterministic generator weights into the reference modules and saves
    seeded inputs + reference outputs as small .npz fixtures;
  * runs the reference MTCNN (real vendored weights) on the reference's own pictures, one
    image per call (NumPy >= 1.24 ragged-array defect);
  * runs the reference RetinaFace (cfg_mnet) with the generator's synthetic weights (its checkpoint is a download);
    torchvision's IntermediateLayerGetter is replaced by a pure-torch stand-in of ours;
  * asks the container's scikit-image 0.18.3 (/opt/conda python3.9) for
    SimilarityTransform.estimate on seeded landmark sets.

Only DATA is written to the repo: arrays, and copies of the reference's picture files used as
detector inputs.  No reference source text is stored.
"""
import importlib
import json
import os
import shutil
import subprocess
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True

from vn_celeb_face_recognition_amd.weights import generate_state_dict  # noqa: E402


# ----------------------------------------------------------------------------- shim
def _nms(boxes, scores, thr):
    order = torch.sort(scores, descending=True, stable=True)[1]
    x1, y1, x2, y2 = boxes.unbind(1)
    areas = (x2 - x1) * (y2 - y1)
    n = boxes.shape[0]
    sup = torch.zeros(n, dtype=torch.bool)
    keep = []
    for _i in range(n):
        i = int(order[_i])
        if sup[i]:
            continue
        keep.append(i)
        rest = order[_i + 1:]
        xx1 = torch.maximum(x1[i], x1[rest]); yy1 = torch.maximum(y1[i], y1[rest])
        xx2 = torch.minimum(x2[i], x2[rest]); yy2 = torch.minimum(y2[i], y2[rest])
        w = (xx2 - xx1).clamp(min=0); h = (yy2 - yy1).clamp(min=0)
        inter = w * h
        ovr = inter / (areas[i] + areas[rest] - inter)
        sup[rest[ovr > thr]] = True
    return torch.as_tensor(keep, dtype=torch.int64)


def _batched_nms(boxes, scores, idxs, iou_threshold):
    # torchvision's coordinate-offset formulation (_batched_nms_coordinate_trick)
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)
    max_coordinate = boxes.max()
    offsets = idxs.to(boxes) * (max_coordinate + torch.tensor(1).to(boxes))
    return _nms(boxes + offsets[:, None], scores, iou_threshold)


class _IntermediateLayerGetter(torch.nn.ModuleDict):
    """Our stand-in for torchvision.models._utils.IntermediateLayerGetter (torchvision is not installed): keeps the
    model's children up to the last requested one and returns the requested activations under their new names."""

    def __init__(self, model, return_layers):
        layers, left = {}, dict(return_layers)
        for name, module in model.named_children():
            layers[name] = module
            left.pop(name, None)
            if not left:
                break
        super().__init__(layers)
        self.return_layers = dict(return_layers)

    def forward(self, x):
        from collections import OrderedDict
        out = OrderedDict()
        for name, module in self.items():
            x = module(x)
            if name in self.return_layers:
                out[self.return_layers[name]] = x
        return out


def install_shim():
    tv = types.ModuleType("torchvision")
    tv.transforms = types.ModuleType("torchvision.transforms")
    tv.transforms.functional = types.ModuleType("torchvision.transforms.functional")
    tv.transforms.functional.to_tensor = lambda a: torch.from_numpy(np.ascontiguousarray(np.transpose(a, (2, 0, 1))))
    tv.ops = types.ModuleType("torchvision.ops")
    tv.ops.boxes = types.ModuleType("torchvision.ops.boxes")
    tv.ops.boxes.batched_nms = _batched_nms
    tv.models = types.ModuleType("torchvision.models")
    tv.models.utils = types.ModuleType("torchvision.models.utils")
    tv.models.utils.load_state_dict_from_url = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("offline"))
    tv.models._utils = types.ModuleType("torchvision.models._utils")
    tv.models._utils.IntermediateLayerGetter = _IntermediateLayerGetter
    tv.models.detection = types.ModuleType("torchvision.models.detection")
    tv.models.detection.backbone_utils = types.ModuleType("torchvision.models.detection.backbone_utils")
    for name in ("torchvision.models._utils", "torchvision.models.detection", "torchvision.models.detection.backbone_utils"):
        mod = tv
        for part in name.split(".")[1:]:
            mod = getattr(mod, part)
        sys.modules[name] = mod
    for name, mod in [("torchvision", tv), ("torchvision.transforms", tv.transforms),
                      ("torchvision.transforms.functional", tv.transforms.functional),
                      ("torchvision.ops", tv.ops), ("torchvision.ops.boxes", tv.ops.boxes),
                      ("torchvision.models", tv.models), ("torchvision.models.utils", tv.models.utils)]:
        sys.modules[name] = mod
    pkg = types.ModuleType("models")
    pkg.__path__ = [os.path.join(REF, "models")]
    sys.modules["models"] = pkg


def ref(name):
    return importlib.import_module("models." + name)


# ----------------------------------------------------------------------------- fixtures
def seeded_normal(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g, dtype=torch.float32)


def golden_irv1():
    m = ref("inception_resnet_v1").InceptionResnetV1(pretrained=None).eval()
    sd = generate_state_dict("irv1", seed=0, as_torch=True)
    missing = m.load_state_dict(sd, strict=True)
    x = seeded_normal((6, 3, 160, 160), 1234)
    # two real face crops resized by plain slicing (no resampling library involved)
    from PIL import Image
    files = sorted(os.listdir(os.path.join(REF, "data")))[:2]
    for k, f in enumerate(files):
        a = np.asarray(Image.open(os.path.join(REF, "data", f)).convert("RGB"))[10:170, 10:170]
        x[4 + k] = torch.from_numpy(((np.float32(a) - 127.5) / 128).transpose(2, 0, 1))
    taps = {}
    hooks = []
    for name in ["conv2d_1a", "conv2d_2b", "conv2d_4b", "repeat_1", "mixed_6a", "repeat_2", "mixed_7a",
                 "repeat_3", "block8", "last_bn"]:
        hooks.append(getattr(m, name).register_forward_hook(
            lambda mod, i, o, name=name: taps.__setitem__(name, o.detach())))
    with torch.no_grad():
        y = m(x)
    for h in hooks:
        h.remove()
    stats = {k: np.array([v.mean().item(), v.abs().mean().item(), v.std().item(), v.abs().max().item()],
                         dtype=np.float64) for k, v in taps.items()}
    np.savez_compressed(os.path.join(OUT, "irv1_seed0.npz"), input_seed=1234, real_crops=np.array(files),
                        real_inputs=x[4:6].numpy().astype(np.float16),  # exact: k/128 grid fits fp16
                        embeddings=y.numpy(), last_bn=taps["last_bn"].numpy(),
                        block8_sample=taps["block8"][:, ::64].numpy(),
                        conv2d_4b_sample=taps["conv2d_4b"][:2, ::32, ::4, ::4].numpy(),
                        **{"stat_" + k: v for k, v in stats.items()})
    print("irv1:", y.shape, "norms", y.norm(dim=1)[:3].tolist(), "missing", missing)


def golden_mlp():
    m = ref("mlp_model").MLPModel(512, 1001).eval()
    m.load_state_dict(generate_state_dict("mlp", seed=0, as_torch=True))
    e = torch.nn.functional.normalize(seeded_normal((32, 512), 77), dim=1)
    with torch.no_grad():
        lp = m(e)
    np.savez_compressed(os.path.join(OUT, "mlp_seed0.npz"), input_seed=77, logp=lp.numpy())
    print("mlp:", lp.shape, "max prob", lp.exp().max(dim=1)[0][:6].tolist())


def golden_ir100():
    mod = ref("iresnet_encoder")
    m = mod.iresnet100(pretrained=False, freeze_weights=False).eval()
    m.load_state_dict(generate_state_dict("iresnet100", seed=0, as_torch=True), strict=True)
    x = seeded_normal((2, 3, 112, 112), 4321)
    with torch.no_grad():
        y = m(x)
    np.savez_compressed(os.path.join(OUT, "ir100_seed0.npz"), input_seed=4321, features=y.numpy())
    print("ir100:", y.shape, y.abs().mean().item())


def golden_mtcnn():
    from PIL import Image
    mt = ref("mtcnn")
    os.makedirs(os.path.join(OUT, "images"), exist_ok=True)
    pics = [("images/mrDam_HaHo_recog.jpg", 50), ("images/mrDam_HaHo_recog.jpg", 20),
            ("images/hoai_linh_4_recog.jpg", 50), ("images/dam_vinh_hung_2_recog.jpg", 40),
            ("images/QuangLe_PhuongMyChi_recog.png", 30)]
    pics += [("data/" + f, 20) for f in sorted(os.listdir(os.path.join(REF, "data")))[:3]]
    out = {}
    meta = []
    for rel, mfs in pics:
        src = os.path.join(REF, rel)
        dst = os.path.join(OUT, "images", os.path.basename(rel))
        if not os.path.exists(dst):
            shutil.copyfile(src, dst)
            os.chmod(dst, 0o644)
        img = np.asarray(Image.open(src).convert("RGB"))
        det = mt.MTCNN(image_size=160, keep_all=True, min_face_size=mfs, device="cpu").eval()
        boxes, probs, points = det.inference(img, landmark=True)
        key = "%s@%d" % (os.path.basename(rel), mfs)
        n = len(boxes)
        out[key + "/boxes"] = np.asarray(boxes, dtype=np.float32).reshape(n, 4)
        out[key + "/probs"] = np.asarray(probs, dtype=np.float32).reshape(n)
        out[key + "/points"] = np.asarray(points, dtype=np.float32).reshape(n, 5, 2)
        meta.append({"file": os.path.basename(rel), "min_face_size": mfs, "n": int(n), "shape": list(img.shape)})
        print("mtcnn:", key, img.shape, "->", n, "faces", np.asarray(probs)[:4])
    # one P-Net level map, to pin the pyramid + P-Net numerics (mrDam, scale index 3 @ minsize 50)
    img = np.asarray(Image.open(os.path.join(REF, "images/mrDam_HaHo_recog.jpg")).convert("RGB"))
    df = ref("mtcnn_utils.detect_face")
    det = mt.MTCNN(min_face_size=50, device="cpu").eval()
    x = torch.from_numpy(img.copy()).permute(2, 0, 1).unsqueeze(0).float()
    h, w = img.shape[:2]
    scale = (12.0 / 50) * 0.709 ** 3
    with torch.no_grad():
        lvl = df.imresample(x, (int(h * scale + 1), int(w * scale + 1)))
        reg, prob = det.pnet((lvl - 127.5) * 0.0078125)
    out["pnet_level/scale"] = np.float64(scale)
    out["pnet_level/level"] = lvl.numpy()
    out["pnet_level/reg"] = reg.numpy()
    out["pnet_level/prob"] = prob.numpy()
    np.savez_compressed(os.path.join(OUT, "mtcnn_ref.npz"), **out)
    with open(os.path.join(OUT, "mtcnn_ref.json"), "w") as f:
        json.dump(meta, f, indent=1)


def golden_retina():
    """RetinaFace (cfg_mnet) with the generator's synthetic weights: synthetic 360x640 frames (heads + detections) and one
    of the reference's pictures (detections).  cfg_mnet['pretrain'] is switched off in memory: its backbone tarball
    (/content/...) does not exist offline and the full state_dict is loaded afterwards anyway."""
    from PIL import Image
    from vn_celeb_face_recognition_amd import synth
    rf = ref("retina_face")
    cfgm = importlib.import_module("models.retina_face_utils.config")
    cfgm.cfg_mnet["pretrain"] = False
    det = rf.RetinaFace("cfg_mnet", phase="test", device="cpu")
    sd = generate_state_dict("retina", seed=0, as_torch=True)
    det.load_state_dict(sd, strict=True)
    det.eval()
    out, meta = {}, []
    frames, _ = synth.make_frames(n_frames=2, faces_per_frame=4, height=360, width=640, seed=1)
    x = torch.stack([torch.from_numpy((np.float32(f) - det.channels_subtract).transpose(2, 0, 1)) for f in frames]).float()
    with torch.no_grad():
        loc, conf, ldm = det.forward(x)
    out["synth/loc"], out["synth/conf"], out["synth/ldm"] = loc.numpy(), conf.numpy(), ldm.numpy()
    cases = [("synth", list(frames))]
    img = np.asarray(Image.open(os.path.join(REF, "images/hoai_linh_4_recog.jpg")).convert("RGB"))
    cases.append(("hoai_linh_4_recog.jpg", [img]))
    import hashlib
    for key, imgs in cases:
        dets, scores, lms = det.inference(imgs, landmark=True)
        for i, (d, sc, lm) in enumerate(zip(dets, scores, lms)):
            out["%s/%d/boxes" % (key, i)] = np.asarray(d, np.float32)
            out["%s/%d/scores" % (key, i)] = np.asarray(sc, np.float32)
            out["%s/%d/points" % (key, i)] = np.asarray(lm, np.float32)
            print("retina:", key, i, imgs[i].shape, "->", len(sc), "faces", sc[:4])
        meta.append({"case": key, "shape": list(imgs[0].shape), "n": [int(len(sc)) for sc in scores],
                     "sha1": hashlib.sha1(np.ascontiguousarray(np.stack(imgs)).tobytes()).hexdigest()})
    np.savez_compressed(os.path.join(OUT, "retina_ref.npz"), **out)
    with open(os.path.join(OUT, "retina_ref.json"), "w") as f:
        json.dump(meta, f, indent=1)


def golden_umeyama():
    rng = np.random.default_rng(5)
    tmpl = np.array([[54.706573, 73.85186], [105.045425, 73.573425], [80.036, 102.48086],
                     [59.356144, 131.95071], [101.04271, 131.72014]], dtype=np.float32)
    lms = []
    for k in range(24):
        ang = rng.uniform(-0.6, 0.6)
        s = rng.uniform(0.4, 3.0)
        R = np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]])
        lm = (tmpl @ R.T) * s + rng.uniform(-40, 200, size=2) + rng.normal(0, 2.0, size=(5, 2))
        if k == 23:
            lm[:, 0] = -lm[:, 0] + 300  # mirrored face: exercises the det(A) < 0 branch
        lms.append(lm.astype(np.float32))
    lms = np.stack(lms)
    np.save("/tmp/_lms.npy", lms)
    np.save("/tmp/_tmpl.npy", tmpl)
    code = ("import numpy as np\nfrom skimage import transform as trans\n"
            "l=np.load('/tmp/_lms.npy'); t=np.load('/tmp/_tmpl.npy'); out=[]\n"
            "for lm in l:\n tf=trans.SimilarityTransform(); tf.estimate(lm, t); out.append(tf.params)\n"
            "np.save('/tmp/_params.npy', np.stack(out))\n")
    subprocess.run(["/opt/conda/bin/python3.9", "-c", code], check=True)
    params = np.load("/tmp/_params.npy")
    np.savez_compressed(os.path.join(OUT, "align_umeyama.npz"), landmarks=lms, template=tmpl,
                        skimage_0_18_3_params=params)
    print("umeyama:", params.shape, params[0])


def golden_celeb_stat():
    """celeb_statistic.py's interval statistics, run from the reference's own function bodies:
    the module itself cannot be imported here (cv2 / pafy / face_alignment / imgaug are absent: ordinary
    ModuleNotFoundError), so the three statistics functions and the two helpers they call are compiled from their
    source text into a scratch namespace -- executed, not stored."""
    import ast as _ast
    import math as _math

    def grab(path, names):
        tree = _ast.parse(open(path).read())
        keep = [n for n in tree.body if isinstance(n, _ast.FunctionDef) and n.name in names]
        return _ast.Module(body=keep, type_ignores=[])

    ns = {"ast": _ast, "math": _math, "json": json}
    exec(compile(grab(os.path.join(REF, "utils", "utils.py"), {"convert_sec_to_max_time_quantity", "write_json"}),
                 "ref_utils", "exec"), ns)
    exec(compile(grab(os.path.join(REF, "celeb_statistic.py"),
                      {"find_celeb_infor_in_interval", "export_json_stat_dynamic_itv", "export_json_stat_fixed_itv"}),
                 "ref_celeb_statistic", "exec"), ns)
    import pandas as pd
    rng = np.random.default_rng(7)
    people = ["celeb_%d" % i for i in range(6)] + ["Unknown"]
    lines = ["Time,Names,Frame_idx,Bboxes,Emotion"]
    t = 0.0
    for r in range(83):
        t += float(rng.choice([0.04, 0.2, 0.25, 1.0]))
        k = int(rng.integers(0, 4))
        names = [people[int(j)] for j in rng.integers(0, len(people), size=k)]
        boxes = [[round(float(v), 6) for v in np.sort(rng.random(4))] for _ in range(k)]
        emos = [["happy", "neutral"][: int(rng.integers(0, 3))] for _ in range(k)]
        lines.append('%s,"%s",%d,"%s","%s"' % (repr(t), names, 1 + r * 5, boxes, emos))
    csv_path = os.path.join(OUT, "celeb_stat_tracker.csv")
    with open(csv_path, "w") as f:
        f.write("\n".join(lines) + "\n")
    df = pd.read_csv(csv_path)
    out = {}
    tmp = os.path.join(OUT, "_tmp_stat.json")
    for tag, fn, arg, nap in (("dynamic_5_4", "export_json_stat_dynamic_itv", 5, 4), ("dynamic_7_2", "export_json_stat_dynamic_itv", 7, 2),
                              ("fixed_16_3", "export_json_stat_fixed_itv", 16, 3), ("fixed_83_1", "export_json_stat_fixed_itv", 83, 1)):
        ns[fn](df, tmp, arg, nap, "Unknown")
        out[tag] = {"text": open(tmp).read()}
    os.remove(tmp)
    out["hms"] = {repr(v): ns["convert_sec_to_max_time_quantity"](v) for v in (0.0, 59.999, 61.5, 3600.0, 7325.25, 86399.99)}
    with open(os.path.join(OUT, "celeb_stat_ref.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("celeb_stat:", {k: len(v.get("text", "")) if isinstance(v, dict) and "text" in v else len(v) for k, v in out.items()})


def mlp_train_case():
    """The synthetic task shared by the golden generator and tests/test_gpu_train.py: 12 identities, unit embeddings =
    class centre + noise; written as the reference's own on-disk formats (<stem>.npz with arr_0, label json of
    class -> file names: find_embedding.py:38-41, data_loader/vn_celeb_dataset.py:39-47)."""
    rng = np.random.default_rng(11)
    n_cls, per_cls_train, per_cls_val = 12, 13, 4
    centres = rng.normal(size=(n_cls, 512)).astype(np.float32)
    files, train, val = {}, {}, {}
    for c in range(n_cls):
        for k in range(per_cls_train + per_cls_val):
            e = centres[c] + 1.6 * rng.normal(size=512).astype(np.float32)
            e = (e / np.linalg.norm(e)).astype(np.float32)
            name = "c%02d_%02d.png" % (c, k)
            files[name] = e
            (train if k < per_cls_train else val).setdefault(str(c), []).append(name)
    cfg = {
        "name": "mlp_train golden", "data_path": "data",
        "train_dataset": {"name": "VNCelebEmbDataset", "args": {"data_dir": "emb", "label_file": "train.json"}},
        "train_data_loader": {"name": "train", "args": {"batch_size": 64, "shuffle": True, "num_workers": 0}},
        "val_dataset": {"name": "VNCelebEmbDataset", "args": {"data_dir": "emb", "label_file": "val.json"}},
        "val_data_loader": {"name": "val", "args": {"batch_size": 32, "shuffle": False, "num_workers": 0}},
        "transforms": "none", "metrics": ["accuracy"], "loss": "neg_log_llhood",
        "model": {"name": "MLPModel", "args": {"input_dim": 512, "num_classes": n_cls}},
        "trainer": {"name": "ClassificationTrainer", "resume_path": "", "save_dir": "saved", "device": "CPU", "log_step": 30,
                    "do_validation": True, "validation_step": 1, "epochs": 6, "tracked_metric": ["val_neg_log_llhood", "min"],
                    "patience": 10, "save_period": 3, "save_result": False, "track4plot": True},
        "optimizer": {"name": "Adam", "args": {"lr": 0.002, "weight_decay": 1e-04}},
        "lr_scheduler": {"name": "ReduceLROnPlateau", "args": {"mode": "min", "threshold": 0.5, "factor": 0.5, "patience": 1,
                                                               "min_lr": 1e-05, "threshold_mode": "rel"}},
    }
    return files, train, val, cfg


def write_mlp_train_case(root):
    files, train, val, cfg = mlp_train_case()
    os.makedirs(os.path.join(root, "emb"), exist_ok=True)
    for name, e in files.items():
        np.savez_compressed(os.path.join(root, "emb", name.split(".")[0] + ".npz"), e)
    for fn, d in (("train.json", train), ("val.json", val)):
        with open(os.path.join(root, fn), "w") as f:
            json.dump(d, f)
    cfg = json.loads(json.dumps(cfg))
    for k in ("train_dataset", "val_dataset"):
        cfg[k]["args"]["data_dir"] = os.path.join(root, "emb")
        cfg[k]["args"]["label_file"] = os.path.join(root, cfg[k]["args"]["label_file"])
    cfg["trainer"]["save_dir"] = os.path.join(root, "saved")
    return cfg


def golden_mlp_train():
    """MLP training on embeddings by the reference's OWN trainer: trainer/base_trainer.py +
    trainer/classification_trainer.py, data_loader/vn_celeb_emb_dataset.py, losses, models/mlp_model.py, driven exactly
    as train.py:22-76 drives them (seed 123, Adam, ReduceLROnPlateau, DataLoader shuffle, dropout), on CPU."""
    import tempfile
    import torch.optim as otpm
    from torch.utils.data import DataLoader
    for name in ("matplotlib", "matplotlib.pyplot", "imgaug", "imgaug.augmenters"):      # imported, never used on this path
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["torchvision"].transforms.Compose = lambda *a, **k: None
    sys.modules["torchvision"].transforms.Lambda = lambda *a, **k: None
    sys.modules["torchvision"].transforms.ToTensor = lambda *a, **k: None
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    root = tempfile.mkdtemp(prefix="vnf_mlp_train_")
    cfg = write_mlp_train_case(root)
    os.chdir(root)                                    # the reference trainer logs relative to the working directory
    try:
        for pkg in ("trainer", "data_loader"):            # bypass the package __init__ files (they import torchvision models / imgaug pipelines)
            m = types.ModuleType(pkg)
            m.__path__ = [os.path.join(REF, pkg)]
            sys.modules[pkg] = m
        import losses as loss_md
        cls_tr = importlib.import_module("trainer.classification_trainer").ClassificationTrainer
        ds_cls = importlib.import_module("data_loader.vn_celeb_emb_dataset").VNCelebEmbDataset
        torch.manual_seed(123)                           # train.py:16-20
        np.random.seed(123)
        train_ds = ds_cls(**cfg["train_dataset"]["args"], transforms=None)
        train_loader = DataLoader(dataset=train_ds, **cfg["train_data_loader"]["args"])
        val_ds = ds_cls(**cfg["val_dataset"]["args"], transforms=None)
        val_loader = DataLoader(dataset=val_ds, **cfg["val_data_loader"]["args"])
        model = ref("mlp_model").MLPModel(**cfg["model"]["args"])
        init = {k: v.detach().clone().numpy() for k, v in model.state_dict().items()}
        criterion = getattr(loss_md, cfg["loss"])
        metrics = [getattr(loss_md, x) for x in cfg["metrics"]]
        optimizer = getattr(otpm, cfg["optimizer"]["name"])(model.parameters(), **cfg["optimizer"]["args"])
        sched = getattr(otpm.lr_scheduler, cfg["lr_scheduler"]["name"])(optimizer, **cfg["lr_scheduler"]["args"])
        tr = cls_tr(cfg, model, criterion, metrics, optimizer, sched)
        tr.setup_loader(train_loader, val_loader)
        # the per-epoch learning rate is not logged by the reference: wrap its epoch method to record it
        lrs, logs = [], []
        orig = tr._train_epoch

        def wrapped(epoch):
            r = orig(epoch)
            lrs.append(optimizer.param_groups[0]["lr"])
            logs.append({k: float(v) for k, v in r.items()})
            return r
        tr._train_epoch = wrapped
        tr.train(cfg["trainer"]["track4plot"])
        ck_dir = str(tr.save_dir)
        cks = sorted(os.listdir(ck_dir))
        cp = torch.load(os.path.join(ck_dir, "checkpoint-epoch6.pth"), weights_only=False)   # our own fresh file
        sd = {k: v.numpy() for k, v in cp["state_dict"].items()}
        track = open(os.path.join(str(tr.log_dir), "log_loss.txt")).read()
    finally:
        os.chdir(cwd)
    out = {
        "init_dense_1_weight_sample": init["dense_1.weight"].reshape(-1)[::4099], "init_dense_2_bias": init["dense_2.bias"],
        "final_dense_1_weight_sample": sd["dense_1.weight"].reshape(-1)[::4099], "final_dense_1_bias": sd["dense_1.bias"],
        "final_dense_2_weight_sample": sd["dense_2.weight"].reshape(-1)[::97], "final_dense_2_bias": sd["dense_2.bias"],
        "lr_after_epoch": np.array(lrs), "train_loss": np.array([l["neg_log_llhood"] for l in logs]),
        "train_acc": np.array([l["accuracy"] for l in logs]), "val_loss": np.array([l["val_neg_log_llhood"] for l in logs]),
        "val_acc": np.array([l["val_accuracy"] for l in logs]),
        "exp_avg_sq_dense_2_bias": cp["optimizer"]["state"][3]["exp_avg_sq"].numpy(),
        "adam_step": np.array(float(cp["optimizer"]["state"][0]["step"])),
    }
    np.savez_compressed(os.path.join(OUT, "mlp_train_ref.npz"), **out)
    meta = {"checkpoint_files": cks, "checkpoint_keys": sorted(cp.keys()), "arch": cp["arch"], "epoch": cp["epoch"],
            "monitor_best": float(cp["monitor_best"]), "optimizer_state_keys": sorted(cp["optimizer"]["state"][0].keys()),
            "param_group_keys": sorted(cp["optimizer"]["param_groups"][0].keys()), "log_loss_txt": track}
    with open(os.path.join(OUT, "mlp_train_ref.json"), "w") as f:
        json.dump(meta, f, indent=1)
    shutil.rmtree(root, ignore_errors=True)
    print("mlp_train:", out["train_loss"], out["val_loss"], out["lr_after_epoch"], cks)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    install_shim()
    torch.manual_seed(0)
    which = sys.argv[1:] or ["irv1", "mlp", "ir100", "mtcnn", "umeyama", "celeb_stat", "retina", "mlp_train"]
    for w in which:
        globals()["golden_" + w]()
