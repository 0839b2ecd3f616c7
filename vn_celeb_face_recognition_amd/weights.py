"""Deterministic synthetic checkpoints with the reference's exact state_dict keys and shapes.

No trained face-embedding or MLP checkpoint exists offline (the reference downloads them:
models/inception_resnet_v1.py:316-331, models/iresnet_encoder.py:8-12,167), so parity fixtures,
tests and bench.py all run on weights produced here.  The generator is pure numpy and keyed by
(seed, crc32(tensor name)), so the GPU box recreates bit-identical tensors without any file
travelling.  Key names follow:

  * InceptionResnetV1  -- models/inception_resnet_v1.py:12-33 (BasicConv2d: conv/bn),
                          36-181 (blocks), 219-258 (top level)
  * MLPModel           -- models/mlp_model.py:6-8 (dense_1, dense_2)
  * IResNet-100        -- models/iresnet_encoder.py:26-61 (IBasicBlock), 83-99, 117-137
  * MTCNN P/R/O-Net    -- models/mtcnn.py:19-28, 62-74, 112-128 (real weights ship as
                          weights_mtcnn/*.pt; the synthetic ones are for stress tests only)
  * RetinaFace (mnet)  -- models/retina_face.py:73-108, retina_face_utils/components.py (the reference downloads
                          its checkpoint: cfg/detection/retina_face.json points at /content/...)

Distributions are chosen so activations keep O(1) magnitude through >100 layers (He-style conv
init, BN statistics near identity), which keeps the 1e-4 parity gate meaningful.
"""
import zlib
from collections import OrderedDict

import numpy as np

__all__ = [
    "irv1_spec", "mlp_spec", "iresnet_spec", "mtcnn_spec", "retina_spec", "generate_state_dict",
    "IRV1_MACS_PER_IMAGE", "IR100_MACS_PER_IMAGE",
]

# exact algorithmic work per image.
IRV1_MACS_PER_IMAGE = 1_417_662_304
IR100_MACS_PER_IMAGE = 12_089_606_144


# synthetic RetinaFace class head (see _draw "cls_w" / "cls_b")
CLS_GAIN = 1.5
CLS_BIAS = 6.0


def _basic_conv(spec, prefix, cin, cout, k):
    kh, kw = (k, k) if isinstance(k, int) else k
    spec.append((prefix + ".conv.weight", (cout, cin, kh, kw), "conv"))
    _bn(spec, prefix + ".bn", cout)


def _bn(spec, prefix, c, res=False):
    spec.append((prefix + ".weight", (c,), "bn_w_res" if res else "bn_w"))
    spec.append((prefix + ".bias", (c,), "bn_b"))
    spec.append((prefix + ".running_mean", (c,), "bn_m"))
    spec.append((prefix + ".running_var", (c,), "bn_v"))
    spec.append((prefix + ".num_batches_tracked", (), "nbt"))


def irv1_spec():
    """Ordered (name, shape, kind) list == InceptionResnetV1(pretrained=None).state_dict()."""
    s = []
    _basic_conv(s, "conv2d_1a", 3, 32, 3)
    _basic_conv(s, "conv2d_2a", 32, 32, 3)
    _basic_conv(s, "conv2d_2b", 32, 64, 3)
    _basic_conv(s, "conv2d_3b", 64, 80, 1)
    _basic_conv(s, "conv2d_4a", 80, 192, 3)
    _basic_conv(s, "conv2d_4b", 192, 256, 3)
    for i in range(5):
        p = "repeat_1.%d" % i
        _basic_conv(s, p + ".branch0", 256, 32, 1)
        _basic_conv(s, p + ".branch1.0", 256, 32, 1)
        _basic_conv(s, p + ".branch1.1", 32, 32, 3)
        _basic_conv(s, p + ".branch2.0", 256, 32, 1)
        _basic_conv(s, p + ".branch2.1", 32, 32, 3)
        _basic_conv(s, p + ".branch2.2", 32, 32, 3)
        s.append((p + ".conv2d.weight", (256, 96, 1, 1), "conv_res"))
        s.append((p + ".conv2d.bias", (256,), "bias"))
    _basic_conv(s, "mixed_6a.branch0", 256, 384, 3)
    _basic_conv(s, "mixed_6a.branch1.0", 256, 192, 1)
    _basic_conv(s, "mixed_6a.branch1.1", 192, 192, 3)
    _basic_conv(s, "mixed_6a.branch1.2", 192, 256, 3)
    for i in range(10):
        p = "repeat_2.%d" % i
        _basic_conv(s, p + ".branch0", 896, 128, 1)
        _basic_conv(s, p + ".branch1.0", 896, 128, 1)
        _basic_conv(s, p + ".branch1.1", 128, 128, (1, 7))
        _basic_conv(s, p + ".branch1.2", 128, 128, (7, 1))
        s.append((p + ".conv2d.weight", (896, 256, 1, 1), "conv_res"))
        s.append((p + ".conv2d.bias", (896,), "bias"))
    _basic_conv(s, "mixed_7a.branch0.0", 896, 256, 1)
    _basic_conv(s, "mixed_7a.branch0.1", 256, 384, 3)
    _basic_conv(s, "mixed_7a.branch1.0", 896, 256, 1)
    _basic_conv(s, "mixed_7a.branch1.1", 256, 256, 3)
    _basic_conv(s, "mixed_7a.branch2.0", 896, 256, 1)
    _basic_conv(s, "mixed_7a.branch2.1", 256, 256, 3)
    _basic_conv(s, "mixed_7a.branch2.2", 256, 256, 3)
    for p in ["repeat_3.%d" % i for i in range(5)] + ["block8"]:
        _basic_conv(s, p + ".branch0", 1792, 192, 1)
        _basic_conv(s, p + ".branch1.0", 1792, 192, 1)
        _basic_conv(s, p + ".branch1.1", 192, 192, (1, 3))
        _basic_conv(s, p + ".branch1.2", 192, 192, (3, 1))
        s.append((p + ".conv2d.weight", (1792, 384, 1, 1), "conv_res"))
        s.append((p + ".conv2d.bias", (1792,), "bias"))
    s.append(("last_linear.weight", (512, 1792), "linear"))
    _bn(s, "last_bn", 512)
    return s


def mlp_spec(input_dim=512, num_classes=1001):
    return [
        ("dense_1.weight", (2048, input_dim), "linear_unit"),
        ("dense_1.bias", (2048,), "bias"),
        ("dense_2.weight", (num_classes, 2048), "linear_wide"),
        ("dense_2.bias", (num_classes,), "bias"),
    ]


def iresnet_spec(layers=(3, 13, 30, 3), num_features=512):
    """IResNet (default layers == iresnet100) state_dict layout."""
    s = [("conv1.weight", (64, 3, 3, 3), "conv")]
    _bn(s, "bn1", 64)
    s.append(("prelu.weight", (64,), "prelu"))
    inplanes = 64
    for li, (planes, nblk) in enumerate(zip((64, 128, 256, 512), layers), start=1):
        for b in range(nblk):
            p = "layer%d.%d" % (li, b)
            cin = inplanes if b == 0 else planes
            _bn(s, p + ".bn1", cin)
            s.append((p + ".conv1.weight", (planes, cin, 3, 3), "conv_lin"))
            _bn(s, p + ".bn2", planes)
            s.append((p + ".prelu.weight", (planes,), "prelu"))
            s.append((p + ".conv2.weight", (planes, planes, 3, 3), "conv_res"))
            _bn(s, p + ".bn3", planes, res=True)
            if b == 0:
                s.append((p + ".downsample.0.weight", (planes, cin, 1, 1), "conv_lin"))
                _bn(s, p + ".downsample.1", planes)
        inplanes = planes
    _bn(s, "bn2", 512)
    s.append(("fc.weight", (num_features, 512 * 49), "linear"))
    s.append(("fc.bias", (num_features,), "bias"))
    _bn(s, "features", num_features)
    return s


def mtcnn_spec(net):
    if net == "pnet":
        return [
            ("conv1.weight", (10, 3, 3, 3), "conv"), ("conv1.bias", (10,), "bias"),
            ("prelu1.weight", (10,), "prelu"),
            ("conv2.weight", (16, 10, 3, 3), "conv"), ("conv2.bias", (16,), "bias"),
            ("prelu2.weight", (16,), "prelu"),
            ("conv3.weight", (32, 16, 3, 3), "conv"), ("conv3.bias", (32,), "bias"),
            ("prelu3.weight", (32,), "prelu"),
            ("conv4_1.weight", (2, 32, 1, 1), "conv"), ("conv4_1.bias", (2,), "bias"),
            ("conv4_2.weight", (4, 32, 1, 1), "conv_res"), ("conv4_2.bias", (4,), "bias"),
        ]
    if net == "rnet":
        return [
            ("conv1.weight", (28, 3, 3, 3), "conv"), ("conv1.bias", (28,), "bias"),
            ("prelu1.weight", (28,), "prelu"),
            ("conv2.weight", (48, 28, 3, 3), "conv"), ("conv2.bias", (48,), "bias"),
            ("prelu2.weight", (48,), "prelu"),
            ("conv3.weight", (64, 48, 2, 2), "conv"), ("conv3.bias", (64,), "bias"),
            ("prelu3.weight", (64,), "prelu"),
            ("dense4.weight", (128, 576), "linear_he"), ("dense4.bias", (128,), "bias"),
            ("prelu4.weight", (128,), "prelu"),
            ("dense5_1.weight", (2, 128), "linear_he"), ("dense5_1.bias", (2,), "bias"),
            ("dense5_2.weight", (4, 128), "linear"), ("dense5_2.bias", (4,), "bias"),
        ]
    if net == "onet":
        return [
            ("conv1.weight", (32, 3, 3, 3), "conv"), ("conv1.bias", (32,), "bias"),
            ("prelu1.weight", (32,), "prelu"),
            ("conv2.weight", (64, 32, 3, 3), "conv"), ("conv2.bias", (64,), "bias"),
            ("prelu2.weight", (64,), "prelu"),
            ("conv3.weight", (64, 64, 3, 3), "conv"), ("conv3.bias", (64,), "bias"),
            ("prelu3.weight", (64,), "prelu"),
            ("conv4.weight", (128, 64, 2, 2), "conv"), ("conv4.bias", (128,), "bias"),
            ("prelu4.weight", (128,), "prelu"),
            ("dense5.weight", (256, 1152), "linear_he"), ("dense5.bias", (256,), "bias"),
            ("prelu5.weight", (256,), "prelu"),
            ("dense6_1.weight", (2, 256), "linear_he"), ("dense6_1.bias", (2,), "bias"),
            ("dense6_2.weight", (4, 256), "linear"), ("dense6_2.bias", (4,), "bias"),
            ("dense6_3.weight", (10, 256), "linear"), ("dense6_3.bias", (10,), "bias"),
        ]
    raise ValueError(net)


def retina_spec():
    """RetinaFace(cfg_mnet).state_dict(): models/retina_face.py:73-108, retina_face_utils/components.py:9-40,100-121
    (the MobileNetV1 avg / fc layers are dropped by IntermediateLayerGetter, retina_face.py:89)."""
    s = []

    def conv_bn(p, cin, cout, k=3, first=False):
        s.append((p + ".0.weight", (cout, cin, k, k), "conv"))
        _bn(s, p + ".1", cout)
        if first:   # raw pixels minus the channel means come in at ~70 rms: the stem's BN brings them to O(1)
            s[-2] = (p + ".1.running_var", (cout,), "bn_v_pix")

    def conv_dw(p, cin, cout):
        s.append((p + ".0.weight", (cin, 1, 3, 3), "conv"))
        _bn(s, p + ".1", cin)
        s.append((p + ".3.weight", (cout, cin, 1, 1), "conv"))
        _bn(s, p + ".4", cout)

    conv_bn("body.stage1.0", 3, 8, first=True)
    for i, (a, b) in enumerate([(8, 16), (16, 32), (32, 32), (32, 64), (64, 64)], start=1):
        conv_dw("body.stage1.%d" % i, a, b)
    conv_dw("body.stage2.0", 64, 128)
    for i in range(1, 6):
        conv_dw("body.stage2.%d" % i, 128, 128)
    conv_dw("body.stage3.0", 128, 256)
    conv_dw("body.stage3.1", 256, 256)
    for i, c in enumerate((64, 128, 256), start=1):
        conv_bn("fpn.output%d" % i, c, 64, k=1)
    conv_bn("fpn.merge1", 64, 64)
    conv_bn("fpn.merge2", 64, 64)
    for i in (1, 2, 3):
        p = "ssh%d" % i
        conv_bn(p + ".conv3X3", 64, 32)
        conv_bn(p + ".conv5X5_1", 64, 16)
        conv_bn(p + ".conv5X5_2", 16, 16)
        conv_bn(p + ".conv7X7_2", 16, 16)
        conv_bn(p + ".conv7x7_3", 16, 16)
    for name, width, kind in (("ClassHead", 2, "cls_w"), ("BboxHead", 4, "conv_res"), ("LandmarkHead", 10, "conv_res")):
        for i in range(3):
            s.append(("%s.%d.conv1x1.weight" % (name, i), (2 * width, 64, 1, 1), kind))
            s.append(("%s.%d.conv1x1.bias" % (name, i), (2 * width,), "cls_b" if kind == "cls_w" else "bias"))
    return s


def _draw(rng, shape, kind):
    if kind == "nbt":
        return np.array(0, dtype=np.int64)
    fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else 1
    if kind in ("conv", "linear_he"):
        a = rng.standard_normal(shape) * np.sqrt(2.0 / fan_in)
    elif kind in ("conv_res", "conv_lin", "linear"):
        a = rng.standard_normal(shape) * np.sqrt(1.0 / fan_in)
    elif kind == "linear_unit":
        # consumes an L2-normalised embedding (element variance 1/fan_in): unit-variance
        # weights give O(1) pre-activations
        a = rng.standard_normal(shape)
    elif kind == "linear_wide":
        # classifier head: logit std ~5 so the winning softmax probability is spread over
        # (0,1) and straddles recognition thresholds (demo_image.py:132-137) in the tests
        a = rng.standard_normal(shape) * np.sqrt(50.0 / fan_in)
    elif kind == "bn_w_res":
        # last BN of a residual branch: small gain keeps the 49-block IR-100 trunk O(1)
        a = rng.uniform(0.15, 0.3, shape)
    elif kind == "bias":
        a = rng.standard_normal(shape) * 0.05
    elif kind == "bn_w":
        a = rng.uniform(0.8, 1.2, shape)
    elif kind == "bn_b":
        a = rng.standard_normal(shape) * 0.1
    elif kind == "bn_m":
        a = rng.standard_normal(shape) * 0.1
    elif kind == "bn_v":
        a = rng.uniform(0.6, 1.4, shape)
    elif kind == "prelu":
        a = rng.uniform(0.1, 0.3, shape)
    elif kind == "bn_v_pix":
        a = rng.uniform(3000.0, 7000.0, shape)
    elif kind == "cls_w":
        # face / background logits: wide enough that a few anchors in a thousand clear vis_thres (0.6) while ~98 %
        # stay under conf_thres (0.02), like a trained detector on a frame with a handful of faces
        a = rng.standard_normal(shape) * np.sqrt(CLS_GAIN / fan_in)
    elif kind == "cls_b":
        a = np.where(np.arange(shape[0]) % 2 == 0, CLS_BIAS, -CLS_BIAS) + rng.standard_normal(shape) * 0.05
    else:
        raise ValueError(kind)
    return a.astype(np.float32)


_SPECS = {
    "irv1": irv1_spec,
    "mlp": mlp_spec,
    "iresnet100": iresnet_spec,
    "pnet": lambda: mtcnn_spec("pnet"),
    "rnet": lambda: mtcnn_spec("rnet"),
    "onet": lambda: mtcnn_spec("onet"),
    "retina": retina_spec,
}


def generate_state_dict(arch, seed=0, as_torch=False, **spec_kwargs):
    """OrderedDict name -> ndarray (fp32; int64 scalar for num_batches_tracked)."""
    spec = _SPECS[arch](**spec_kwargs)
    out = OrderedDict()
    for name, shape, kind in spec:
        rng = np.random.default_rng([int(seed), zlib.crc32((arch + "/" + name).encode())])
        out[name] = _draw(rng, shape, kind)
    if as_torch:
        import torch
        out = OrderedDict((k, torch.from_numpy(np.ascontiguousarray(v))) for k, v in out.items())
    return out
