"""Oracle: the MTCNN P/R/O-Net cascade, restated from the reference.

Follows /root/reference/models/mtcnn.py (PNet.forward 38-49, RNet.forward 84-99,
ONet.forward 138-157, MTCNN.detect 318-361, MTCNN.inference 511-513) and
/root/reference/models/mtcnn_utils/detect_face.py (detect_face 25-185, bbreg 188-200,
generateBoundingBox 203-218, nms_numpy 221-257, batched_nms_numpy 260-274, pad 277-289,
rerec 292-301, imresample 304-306).  Test infrastructure only.

Third-party arithmetic restated here because torchvision is absent offline ("parity unpinned"
at this boundary): torchvision.ops.batched_nms (detect_face.py:79,93,128) ==
greedy NMS per image, candidates visited in stable score-descending order, area =
(x2-x1)*(y2-y1) (no +1), a later box is dropped when inter/(a_i+a_j-inter) > thr, kept rows
returned in score-descending order.  torchvision has two implementations (coordinate-offset
trick below 4000 elements on CPU, per-class loop above); both compute this per-image result,
the trick differing only by fp32 rounding of offset coordinates for image index > 0.  The
oracle (and the HIP kernels) use the per-image form.

The nets use torch-CPU fp32 functional ops exactly as the reference modules do.
"""
import numpy as np
import torch
import torch.nn.functional as F


def _t(sd, k):
    v = sd[k]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(v)


# ----------------------------------------------------------------------------- nets
def pnet_forward(sd, x):
    """mtcnn.py:38-49 -> (reg (B,4,oh,ow), prob (B,2,oh,ow))."""
    x = F.prelu(F.conv2d(x, _t(sd, "conv1.weight"), _t(sd, "conv1.bias")), _t(sd, "prelu1.weight"))
    x = F.max_pool2d(x, 2, 2, ceil_mode=True)
    x = F.prelu(F.conv2d(x, _t(sd, "conv2.weight"), _t(sd, "conv2.bias")), _t(sd, "prelu2.weight"))
    x = F.prelu(F.conv2d(x, _t(sd, "conv3.weight"), _t(sd, "conv3.bias")), _t(sd, "prelu3.weight"))
    a = F.softmax(F.conv2d(x, _t(sd, "conv4_1.weight"), _t(sd, "conv4_1.bias")), dim=1)
    b = F.conv2d(x, _t(sd, "conv4_2.weight"), _t(sd, "conv4_2.bias"))
    return b, a


def rnet_forward(sd, x):
    """mtcnn.py:84-99 -> (reg (N,4), prob (N,2))."""
    x = F.prelu(F.conv2d(x, _t(sd, "conv1.weight"), _t(sd, "conv1.bias")), _t(sd, "prelu1.weight"))
    x = F.max_pool2d(x, 3, 2, ceil_mode=True)
    x = F.prelu(F.conv2d(x, _t(sd, "conv2.weight"), _t(sd, "conv2.bias")), _t(sd, "prelu2.weight"))
    x = F.max_pool2d(x, 3, 2, ceil_mode=True)
    x = F.prelu(F.conv2d(x, _t(sd, "conv3.weight"), _t(sd, "conv3.bias")), _t(sd, "prelu3.weight"))
    x = x.permute(0, 3, 2, 1).contiguous()  # flatten order (W,H,C): mtcnn.py:93-94
    x = F.prelu(F.linear(x.view(x.shape[0], -1), _t(sd, "dense4.weight"), _t(sd, "dense4.bias")),
                _t(sd, "prelu4.weight"))
    a = F.softmax(F.linear(x, _t(sd, "dense5_1.weight"), _t(sd, "dense5_1.bias")), dim=1)
    b = F.linear(x, _t(sd, "dense5_2.weight"), _t(sd, "dense5_2.bias"))
    return b, a


def onet_forward(sd, x):
    """mtcnn.py:138-157 -> (reg (N,4), landmarks (N,10), prob (N,2))."""
    x = F.prelu(F.conv2d(x, _t(sd, "conv1.weight"), _t(sd, "conv1.bias")), _t(sd, "prelu1.weight"))
    x = F.max_pool2d(x, 3, 2, ceil_mode=True)
    x = F.prelu(F.conv2d(x, _t(sd, "conv2.weight"), _t(sd, "conv2.bias")), _t(sd, "prelu2.weight"))
    x = F.max_pool2d(x, 3, 2, ceil_mode=True)
    x = F.prelu(F.conv2d(x, _t(sd, "conv3.weight"), _t(sd, "conv3.bias")), _t(sd, "prelu3.weight"))
    x = F.max_pool2d(x, 2, 2, ceil_mode=True)
    x = F.prelu(F.conv2d(x, _t(sd, "conv4.weight"), _t(sd, "conv4.bias")), _t(sd, "prelu4.weight"))
    x = x.permute(0, 3, 2, 1).contiguous()
    x = F.prelu(F.linear(x.view(x.shape[0], -1), _t(sd, "dense5.weight"), _t(sd, "dense5.bias")),
                _t(sd, "prelu5.weight"))
    a = F.softmax(F.linear(x, _t(sd, "dense6_1.weight"), _t(sd, "dense6_1.bias")), dim=1)
    b = F.linear(x, _t(sd, "dense6_2.weight"), _t(sd, "dense6_2.bias"))
    c = F.linear(x, _t(sd, "dense6_3.weight"), _t(sd, "dense6_3.bias"))
    return b, c, a


def _chunked(fn, sd, x, chunk=512):
    """detect_face.py:16-23 fixed_batch_process."""
    outs = [fn(sd, x[i:i + chunk]) for i in range(0, len(x), chunk)]
    return tuple(torch.cat(v, dim=0) for v in zip(*outs))


# ----------------------------------------------------------------------------- geometry
def scale_pyramid(h, w, minsize, factor):
    """detect_face.py:50-60 (python doubles)."""
    m = 12.0 / minsize
    minl = min(h, w) * m
    scale_i = m
    scales = []
    while minl >= 12:
        scales.append(scale_i)
        scale_i = scale_i * factor
        minl = minl * factor
    return scales


def level_size(h, w, scale):
    """detect_face.py:71."""
    return int(h * scale + 1), int(w * scale + 1)


def area_resample(img, oh, ow):
    """detect_face.py:304-306: interpolate(mode='area') == adaptive average pooling with bins
    [floor(i*H/oh), ceil((i+1)*H/oh)).  Explicit fp32 restatement (row-major running sum,
    then divided by the bin height and then by the bin width -- two roundings, as ATen does) used to pin the HIP kernels bit for bit; it is
    checked against torch's own kernel in tests/test_oracle_mtcnn.py."""
    img = np.asarray(img, dtype=np.float32)
    C, H, W = img.shape[-3:]
    lead = img.shape[:-3]
    x = img.reshape((-1, H, W))
    out = np.empty((x.shape[0], oh, ow), dtype=np.float32)
    for i in range(oh):
        h0 = (i * H) // oh
        h1 = -((-(i + 1) * H) // oh)
        for j in range(ow):
            w0 = (j * W) // ow
            w1 = -((-(j + 1) * W) // ow)
            acc = np.zeros(x.shape[0], dtype=np.float32)
            for ih in range(h0, h1):
                for iw in range(w0, w1):
                    acc = (acc + x[:, ih, iw]).astype(np.float32)
            out[:, i, j] = (acc / np.float32(h1 - h0)) / np.float32(w1 - w0)
    return out.reshape(lead + (C, oh, ow))


def imresample(img, sz):
    return F.interpolate(img, size=sz, mode="area")


def generate_bounding_box(reg, probs, scale, thresh):
    """detect_face.py:203-218; reg (B,4,oh,ow), probs (B,oh,ow) torch fp32."""
    stride, cellsize = 2, 12
    reg = reg.permute(1, 0, 2, 3)
    mask = probs >= thresh
    mask_inds = mask.nonzero()
    image_inds = mask_inds[:, 0]
    score = probs[mask]
    reg = reg[:, mask].permute(1, 0)
    bb = mask_inds[:, 1:].type(reg.dtype).flip(1)
    q1 = ((stride * bb + 1) / scale).floor()
    q2 = ((stride * bb + cellsize - 1 + 1) / scale).floor()
    return torch.cat([q1, q2, score.unsqueeze(1), reg], dim=1).numpy(), image_inds.numpy()


def nms_iou(boxes, scores, thr):
    """torchvision.ops.nms semantics (see module header).  Returns kept indices, score-desc."""
    n = boxes.shape[0]
    if n == 0:
        return np.zeros((0,), dtype=np.int64)
    b = boxes.astype(np.float32)
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    areas = ((x2 - x1) * (y2 - y1)).astype(np.float32)
    order = np.argsort(-scores.astype(np.float32), kind="stable")
    suppressed = np.zeros(n, dtype=bool)
    keep = []
    thr = np.float32(thr)
    for _i in range(n):
        i = order[_i]
        if suppressed[i]:
            continue
        keep.append(i)
        rest = order[_i + 1:]
        xx1 = np.maximum(x1[i], x1[rest])
        yy1 = np.maximum(y1[i], y1[rest])
        xx2 = np.minimum(x2[i], x2[rest])
        yy2 = np.minimum(y2[i], y2[rest])
        w = np.maximum(np.float32(0), xx2 - xx1)
        h = np.maximum(np.float32(0), yy2 - yy1)
        inter = (w * h).astype(np.float32)
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = inter / (areas[i] + areas[rest] - inter)
        suppressed[rest[ovr > thr]] = True
    return np.asarray(keep, dtype=np.int64)


def batched_nms(boxes, scores, idxs, thr):
    """Per-image NMS; kept rows returned in global score-descending order (stable)."""
    if boxes.shape[0] == 0:
        return np.zeros((0,), dtype=np.int64)
    keep = []
    for c in np.unique(idxs):
        rows = np.nonzero(idxs == c)[0]
        keep.append(rows[nms_iou(boxes[rows], scores[rows], thr)])
    keep = np.concatenate(keep)
    return keep[np.argsort(-scores[keep].astype(np.float32), kind="stable")]


def nms_min(boxes, scores, thr, ties="numpy"):
    """detect_face.py:221-257 with method 'Min': areas with +1, ascending argsort, visit from the
    back, overlap = inter / min(area_i, area_j), survivors are those with o <= thr.

    ties: the reference calls np.argsort(s) with NumPy's default UNSTABLE sort, so the visiting
    order of EQUAL scores is implementation defined (it changes with array length, NumPy version
    and the CPU's SIMD dispatch) -- and equal scores do occur: softmax saturates to exactly 1.0f
    on clear faces.  "numpy" keeps the verbatim call (what the goldens captured in the build
    container); "table" pins equal scores to table order (lower row first), the deterministic
    rule the HIP kernel implements."""
    if boxes.size == 0:
        return np.zeros((0,), dtype=np.int64)
    x1, y1, x2, y2 = (boxes[:, k].copy() for k in range(4))
    area = (x2 - x1 + 1) * (y2 - y1 + 1)
    I = np.argsort(scores) if ties == "numpy" else np.lexsort((-np.arange(len(scores)), scores))
    pick = []
    while I.size > 0:
        i = I[-1]
        pick.append(i)
        idx = I[:-1]
        xx1 = np.maximum(x1[i], x1[idx])
        yy1 = np.maximum(y1[i], y1[idx])
        xx2 = np.minimum(x2[i], x2[idx])
        yy2 = np.minimum(y2[i], y2[idx])
        w = np.maximum(0.0, xx2 - xx1 + 1)
        h = np.maximum(0.0, yy2 - yy1 + 1)
        inter = w * h
        o = inter / np.minimum(area[i], area[idx])
        I = I[np.where(o <= thr)]
    return np.asarray(pick, dtype=np.int64)


def batched_nms_min(boxes, scores, idxs, thr, ties="numpy"):
    """detect_face.py:260-274 per image (offset trick == per-image result; header)."""
    if boxes.shape[0] == 0:
        return np.zeros((0,), dtype=np.int64)
    picks = []
    for c in np.unique(idxs):
        rows = np.nonzero(idxs == c)[0]
        picks.append(rows[nms_min(boxes[rows], scores[rows], thr, ties)])
    keep = np.concatenate(picks)
    # the reference runs ONE nms over offset boxes: picks come out in global score-descending order
    order = np.argsort(-scores[keep], kind="stable")
    return keep[order]


def bbreg(box, reg):
    """detect_face.py:188-200 (w,h WITH +1)."""
    box = box.copy()
    w = box[:, 2] - box[:, 0] + np.float32(1)
    h = box[:, 3] - box[:, 1] + np.float32(1)
    b1 = box[:, 0] + reg[:, 0] * w
    b2 = box[:, 1] + reg[:, 1] * h
    b3 = box[:, 2] + reg[:, 2] * w
    b4 = box[:, 3] + reg[:, 3] * h
    box[:, :4] = np.stack([b1, b2, b3, b4], axis=1)
    return box


def rerec(box):
    """detect_face.py:292-301."""
    box = box.copy()
    h = box[:, 3] - box[:, 1]
    w = box[:, 2] - box[:, 0]
    l = np.maximum(w, h)
    box[:, 0] = box[:, 0] + w * np.float32(0.5) - l * np.float32(0.5)
    box[:, 1] = box[:, 1] + h * np.float32(0.5) - l * np.float32(0.5)
    box[:, 2] = box[:, 0] + l
    box[:, 3] = box[:, 1] + l
    return box


def pad(boxes, w, h):
    """detect_face.py:277-289: trunc toward zero, clamp; returns y, ey, x, ex int32."""
    b = np.trunc(boxes).astype(np.int32)
    x, y, ex, ey = b[:, 0].copy(), b[:, 1].copy(), b[:, 2].copy(), b[:, 3].copy()
    x[x < 1] = 1
    y[y < 1] = 1
    ex[ex > w] = w
    ey[ey > h] = h
    return y, ey, x, ex


def _crops(imgs, image_inds, y, ey, x, ex, size):
    """detect_face.py:108-114 / 137-143."""
    out = []
    for k in range(len(y)):
        if ey[k] > (y[k] - 1) and ex[k] > (x[k] - 1):
            img_k = imgs[int(image_inds[k]), :, (y[k] - 1):ey[k], (x[k] - 1):ex[k]].unsqueeze(0)
            out.append(imresample(img_k, (size, size)))
    out = torch.cat(out, dim=0)
    return (out - 127.5) * 0.0078125


# ----------------------------------------------------------------------------- cascade
def detect_face(imgs, minsize, pnet_sd, rnet_sd, onet_sd, threshold, factor, stages=None, ties="numpy"):
    """detect_face.py:25-185.  imgs: (B,H,W,3) uint8 ndarray (or a list of equal-size HWC
    arrays).  Returns (list of (n,5) boxes, list of (n,5,2) points) per image.
    `stages`, if a dict, receives the intermediate tables used by the staged parity tests."""
    if isinstance(imgs, (list, tuple)):
        if any(np.asarray(i).shape != np.asarray(imgs[0]).shape for i in imgs):
            raise Exception("MTCNN batch processing only compatible with equal-dimension images.")
        imgs = np.stack([np.uint8(i) for i in imgs])
    imgs = np.asarray(imgs)
    if imgs.ndim == 3:
        imgs = imgs[None]
    with torch.no_grad():
        x = torch.from_numpy(imgs.copy()).permute(0, 3, 1, 2).float()
        B, _, h, w = x.shape
        scales = scale_pyramid(h, w, minsize, factor)

        boxes, image_inds, scale_picks = [], [], []
        offset = 0
        for si, scale in enumerate(scales):
            im_data = imresample(x, level_size(h, w, scale))
            im_data = (im_data - 127.5) * 0.0078125
            reg, probs = pnet_forward(pnet_sd, im_data)
            if stages is not None and si in stages.get("want_pnet_levels", ()):
                stages["pnet_level_%d" % si] = (im_data.numpy(), reg.numpy(), probs.numpy())
            b_s, ii_s = generate_bounding_box(reg, probs[:, 1], scale, threshold[0])
            boxes.append(b_s)
            image_inds.append(ii_s)
            pick = batched_nms(b_s[:, :4], b_s[:, 4], ii_s, 0.5)
            scale_picks.append(pick + offset)
            offset += b_s.shape[0]
        boxes = np.concatenate(boxes, axis=0) if boxes else np.zeros((0, 9), np.float32)
        image_inds = np.concatenate(image_inds, axis=0) if image_inds else np.zeros((0,), np.int64)
        scale_picks = np.concatenate(scale_picks, axis=0) if scale_picks else np.zeros((0,), np.int64)
        if stages is not None:
            stages["scales"] = list(scales)
            stages["n_stage1_raw"] = int(boxes.shape[0])
        boxes, image_inds = boxes[scale_picks], image_inds[scale_picks]
        if stages is not None:
            stages["stage1_scale_nms"] = (boxes.copy(), image_inds.copy())

        pick = batched_nms(boxes[:, :4], boxes[:, 4], image_inds, 0.7)
        boxes, image_inds = boxes[pick], image_inds[pick]

        regw = boxes[:, 2] - boxes[:, 0]
        regh = boxes[:, 3] - boxes[:, 1]
        qq1 = boxes[:, 0] + boxes[:, 5] * regw
        qq2 = boxes[:, 1] + boxes[:, 6] * regh
        qq3 = boxes[:, 2] + boxes[:, 7] * regw
        qq4 = boxes[:, 3] + boxes[:, 8] * regh
        boxes = np.stack([qq1, qq2, qq3, qq4, boxes[:, 4]], axis=1).astype(np.float32)
        boxes = rerec(boxes)
        y, ey, xx, ex = pad(boxes, w, h)
        if stages is not None:
            stages["stage1"] = (boxes.copy(), image_inds.copy())

        if len(boxes) > 0:
            im_data = _crops(x, image_inds, y, ey, xx, ex, 24)
            out0, out1 = _chunked(rnet_forward, rnet_sd, im_data)
            out0, out1 = out0.numpy(), out1.numpy()
            score = out1[:, 1]
            ipass = score > np.float32(threshold[1])
            boxes = np.concatenate([boxes[ipass, :4], score[ipass, None]], axis=1)
            image_inds = image_inds[ipass]
            mv = out0[ipass]
            pick = batched_nms(boxes[:, :4], boxes[:, 4], image_inds, 0.7)
            boxes, image_inds, mv = boxes[pick], image_inds[pick], mv[pick]
            boxes = bbreg(boxes, mv)
            boxes = rerec(boxes)
        if stages is not None:
            stages["stage2"] = (boxes.copy(), image_inds.copy())

        points = np.zeros((0, 5, 2), dtype=np.float32)
        if len(boxes) > 0:
            y, ey, xx, ex = pad(boxes, w, h)
            im_data = _crops(x, image_inds, y, ey, xx, ex, 48)
            out0, out1, out2 = _chunked(onet_forward, onet_sd, im_data)
            out0, out1, out2 = out0.numpy(), out1.numpy(), out2.numpy()
            score = out2[:, 1]
            ipass = score > np.float32(threshold[2])
            pts = out1[ipass]
            boxes = np.concatenate([boxes[ipass, :4], score[ipass, None]], axis=1)
            image_inds = image_inds[ipass]
            mv = out0[ipass]
            w_i = boxes[:, 2] - boxes[:, 0] + np.float32(1)
            h_i = boxes[:, 3] - boxes[:, 1] + np.float32(1)
            px = w_i[:, None] * pts[:, 0:5] + boxes[:, 0:1] - np.float32(1)
            py = h_i[:, None] * pts[:, 5:10] + boxes[:, 1:2] - np.float32(1)
            points = np.stack([px, py], axis=2).astype(np.float32)
            boxes = bbreg(boxes, mv)
            if stages is not None:
                stages["stage3_pre_nms"] = (boxes.copy(), image_inds.copy(), points.copy())
            pick = batched_nms_min(boxes[:, :4], boxes[:, 4], image_inds, 0.7, ties)
            boxes, image_inds, points = boxes[pick], image_inds[pick], points[pick]

        batch_boxes, batch_points = [], []
        for b_i in range(B):
            sel = np.where(image_inds == b_i)[0]
            batch_boxes.append(boxes[sel].copy())
            batch_points.append(points[sel].copy())
        return batch_boxes, batch_points


def mtcnn_detect(imgs, pnet_sd, rnet_sd, onet_sd, min_face_size=20, thresholds=(0.6, 0.7, 0.7),
                 factor=0.709, select_largest=True, landmarks=True, stages=None, ties="numpy"):
    """mtcnn.py:318-361 + inference 511-513.  Returns per-image lists (ragged-safe: the
    reference's np.array() of ragged lists raises on NumPy >= 1.24)."""
    single = not isinstance(imgs, (list, tuple)) and np.asarray(imgs).ndim == 3
    bb, pp = detect_face(imgs, min_face_size, pnet_sd, rnet_sd, onet_sd, list(thresholds), factor,
                         stages=stages, ties=ties)
    boxes, probs, points = [], [], []
    for box, point in zip(bb, pp):
        if len(box) == 0:
            boxes.append([]); probs.append([]); points.append([])
            continue
        if select_largest:
            order = np.argsort((box[:, 2] - box[:, 0]) * (box[:, 3] - box[:, 1]))[::-1]
            box, point = box[order], point[order]
        boxes.append(box[:, :4]); probs.append(box[:, 4]); points.append(point)
    if single:
        boxes, probs, points = boxes[0], probs[0], points[0]
    return (boxes, probs, points) if landmarks else (boxes, probs)
