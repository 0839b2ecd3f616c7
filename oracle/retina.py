"""Oracle: RetinaFace (mobilenet0.25 configuration), restated from the reference.  Test infrastructure only.

Follows /root/reference/models/retina_face.py (forward 133-154, inference 156-232),
/root/reference/models/retina_face_utils/components.py (conv_bn / conv_dw 9-40, SSH 42-64, FPN 66-97, MobileNetV1 100-121),
retina_face_utils/config.py cfg_mnet (1-19), retina_face_utils/prior_box.py (PriorBox.forward 20-34),
retina_face_utils/box_utils.py (decode 209-227, decode_landm 229-247) and retina_face_utils/nms/py_cpu_nms.py (10-37).

The network uses torch-CPU fp32 functional ops exactly as the reference modules do (eval-mode BatchNorm, eps 1e-5).

Score ties: the reference orders with `scores.argsort()[::-1]` (NumPy default = unstable introsort) once for the top-K cut
and once inside py_cpu_nms.  ties="numpy" does the same calls; ties="table" pins a stable sort (kind="stable") in both
places, which is the rule the HIP kernels implement.  The two agree whenever scores are distinct.
"""
from itertools import product
from math import ceil

import numpy as np
import torch
import torch.nn.functional as F

CFG_MNET = {"min_sizes": [[16, 32], [64, 128], [256, 512]], "steps": [8, 16, 32], "variance": [0.1, 0.2], "clip": False}
CHANNELS_SUBTRACT = (104, 117, 123)


def _t(sd, k):
    v = sd[k]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v))


def _bn(sd, p, x):
    return F.batch_norm(x, _t(sd, p + ".running_mean"), _t(sd, p + ".running_var"), _t(sd, p + ".weight"), _t(sd, p + ".bias"),
                        False, 0.1, 1e-5)


def conv_bn(sd, p, x, stride=1, leaky=None, k=3):
    """components.py:9-28: conv(k, stride, pad k//2, no bias) + BN (+ LeakyReLU when leaky is not None)."""
    x = _bn(sd, p + ".1", F.conv2d(x, _t(sd, p + ".0.weight"), None, stride, k // 2))
    return x if leaky is None else F.leaky_relu(x, leaky)


def conv_dw(sd, p, x, stride, leaky=0.1):
    """components.py:30-40."""
    w = _t(sd, p + ".0.weight")
    x = F.leaky_relu(_bn(sd, p + ".1", F.conv2d(x, w, None, stride, 1, 1, w.shape[0])), leaky)
    return F.leaky_relu(_bn(sd, p + ".4", F.conv2d(x, _t(sd, p + ".3.weight"))), leaky)


def body(sd, x):
    """MobileNetV1 stages 1..3 through IntermediateLayerGetter (components.py:100-121, retina_face.py:89)."""
    x = conv_bn(sd, "body.stage1.0", x, 2, 0.1)
    for i, s in ((1, 1), (2, 2), (3, 1), (4, 2), (5, 1)):
        x = conv_dw(sd, "body.stage1.%d" % i, x, s)
    c1 = x
    for i in range(6):
        x = conv_dw(sd, "body.stage2.%d" % i, x, 2 if i == 0 else 1)
    c2 = x
    x = conv_dw(sd, "body.stage3.0", x, 2)
    c3 = conv_dw(sd, "body.stage3.1", x, 1)
    return c1, c2, c3


def fpn(sd, c1, c2, c3):
    """components.py:66-97 (out_channels 64 -> leaky 0.1)."""
    o1 = conv_bn(sd, "fpn.output1", c1, 1, 0.1, k=1)
    o2 = conv_bn(sd, "fpn.output2", c2, 1, 0.1, k=1)
    o3 = conv_bn(sd, "fpn.output3", c3, 1, 0.1, k=1)
    o2 = conv_bn(sd, "fpn.merge2", o2 + F.interpolate(o3, size=[o2.size(2), o2.size(3)], mode="nearest"), 1, 0.1)
    o1 = conv_bn(sd, "fpn.merge1", o1 + F.interpolate(o2, size=[o1.size(2), o1.size(3)], mode="nearest"), 1, 0.1)
    return o1, o2, o3


def ssh(sd, p, x):
    """components.py:42-64 (out_channel 64 -> leaky 0.1)."""
    a = conv_bn(sd, p + ".conv3X3", x)
    t = conv_bn(sd, p + ".conv5X5_1", x, 1, 0.1)
    b = conv_bn(sd, p + ".conv5X5_2", t)
    c = conv_bn(sd, p + ".conv7x7_3", conv_bn(sd, p + ".conv7X7_2", t, 1, 0.1))
    return F.relu(torch.cat([a, b, c], dim=1))


def _head(sd, name, i, x, width):
    """retina_face.py:20-54: 1x1 conv, NHWC, view(N, -1, width)."""
    y = F.conv2d(x, _t(sd, "%s.%d.conv1x1.weight" % (name, i)), _t(sd, "%s.%d.conv1x1.bias" % (name, i)))
    return y.permute(0, 2, 3, 1).contiguous().view(y.shape[0], -1, width)


def forward(sd, x, logits=False):
    """retina_face.py:133-154, phase 'test': (bbox (N,A,4), softmax conf (N,A,2), landmarks (N,A,10))."""
    with torch.no_grad():
        feats = [ssh(sd, "ssh%d" % (i + 1), f) for i, f in enumerate(fpn(sd, *body(sd, x)))]
        bbox = torch.cat([_head(sd, "BboxHead", i, f, 4) for i, f in enumerate(feats)], dim=1)
        cls = torch.cat([_head(sd, "ClassHead", i, f, 2) for i, f in enumerate(feats)], dim=1)
        ldm = torch.cat([_head(sd, "LandmarkHead", i, f, 10) for i, f in enumerate(feats)], dim=1)
    return bbox, (cls if logits else F.softmax(cls, dim=-1)), ldm


def prior_box(height, width):
    """prior_box.py:9-34 (clip False): (A,4) fp32 [cx, cy, s_kx, s_ky]."""
    anchors = []
    for k, step in enumerate(CFG_MNET["steps"]):
        fh, fw = ceil(height / step), ceil(width / step)
        for i, j in product(range(fh), range(fw)):
            for min_size in CFG_MNET["min_sizes"][k]:
                s_kx = min_size / width
                s_ky = min_size / height
                anchors += [(j + 0.5) * step / width, (i + 0.5) * step / height, s_kx, s_ky]
    return torch.Tensor(anchors).view(-1, 4)


def decode(loc, priors, variances):
    """box_utils.py:209-227."""
    boxes = torch.cat((priors[:, :2] + loc[:, :2] * variances[0] * priors[:, 2:],
                       priors[:, 2:] * torch.exp(loc[:, 2:] * variances[1])), 1)
    boxes[:, :2] -= boxes[:, 2:] / 2
    boxes[:, 2:] += boxes[:, :2]
    return boxes


def decode_landm(pre, priors, variances):
    """box_utils.py:229-247."""
    return torch.cat(tuple(priors[:, :2] + pre[:, 2 * j:2 * j + 2] * variances[0] * priors[:, 2:] for j in range(5)), dim=1)


def _argsort_desc(scores, ties):
    return scores.argsort(kind="stable" if ties == "table" else None)[::-1]


def py_cpu_nms(dets, thresh, ties="numpy"):
    """nms/py_cpu_nms.py:10-37."""
    x1, y1, x2, y2, scores = dets[:, 0], dets[:, 1], dets[:, 2], dets[:, 3], dets[:, 4]
    areas = (x2 - x1 + 1) * (y2 - y1 + 1)
    order = _argsort_desc(scores, ties)
    keep = []
    while order.size > 0:
        i = order[0]
        keep.append(i)
        xx1 = np.maximum(x1[i], x1[order[1:]])
        yy1 = np.maximum(y1[i], y1[order[1:]])
        xx2 = np.minimum(x2[i], x2[order[1:]])
        yy2 = np.minimum(y2[i], y2[order[1:]])
        w = np.maximum(0.0, xx2 - xx1 + 1)
        h = np.maximum(0.0, yy2 - yy1 + 1)
        inter = w * h
        ovr = inter / (areas[i] + areas[order[1:]] - inter)
        inds = np.where(ovr <= thresh)[0]
        order = order[inds + 1]
    return keep


def postprocess(loc, conf, landms, height, width, conf_thres=0.02, topk_bf_nms=5000, keep_top_k=750, nms_thres=0.4,
                vis_thres=0.6, ties="numpy"):
    """retina_face.py:176-221 for one image: loc (A,4), conf (A,2) softmax, landms (A,10) torch fp32 ->
    (dets (k,4), scores (k,), landmarks (k,5,2))."""
    priors = prior_box(height, width)
    var = CFG_MNET["variance"]
    boxes = (decode(loc, priors, var) * torch.Tensor([width, height, width, height])).numpy()
    scores = conf.numpy()[:, 1]
    lm = (decode_landm(landms, priors, var) * torch.Tensor([width, height] * 5)).numpy()
    inds = np.where(scores > conf_thres)[0]
    boxes, lm, scores = boxes[inds], lm[inds], scores[inds]
    order = _argsort_desc(scores, ties)[:topk_bf_nms]
    boxes, lm, scores = boxes[order], lm[order], scores[order]
    dets = np.hstack((boxes, scores[:, np.newaxis])).astype(np.float32, copy=False)
    keep = py_cpu_nms(dets, nms_thres, ties)
    dets, lm = dets[keep, :][:keep_top_k, :], lm[keep][:keep_top_k, :]
    chosen = dets[:, 4] >= vis_thres
    dets, lm = dets[chosen, :], lm[chosen, :]
    return dets[:, :4], dets[:, 4], lm.reshape(-1, 5, 2)


def inference(sd, rgb_images, landmark=True, ties="numpy", **thresholds):
    """retina_face.py:156-232: lists (one entry per image) of boxes (k,4), scores (k,), landmarks (k,5,2)."""
    arr = [np.float32(im) - CHANNELS_SUBTRACT for im in rgb_images]
    h, w, _ = arr[0].shape
    x = torch.stack([torch.from_numpy(a.transpose(2, 0, 1)) for a in arr], dim=0).to(torch.float)
    loc, conf, landms = forward(sd, x)
    out = [postprocess(loc[i], conf[i], landms[i], h, w, ties=ties, **thresholds) for i in range(loc.shape[0])]
    dets, scores, lms = [o[0] for o in out], [o[1] for o in out], [o[2] for o in out]
    return (dets, scores, lms) if landmark else (dets, scores)
