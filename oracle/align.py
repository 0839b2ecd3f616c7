"""Oracle: face crop, 5-point similarity alignment and input normalisation.

Restates:
  /root/reference/demo_image.py:174-199 get_face_from_boxes, 236-239 move_landmark_to_box,
      273-306 parallel_detect_and_align (the RGB->BGR->RGB round trip at 291-294 is the
      identity for a per-channel warp and is not restated)
  /root/reference/align_face.py:12-48 center_point_dict, 51-57 alignment
  /root/reference/data_loader/__init__.py:27-34,52-56 transforms_default
Test infrastructure only.

Third-party arithmetic (absent offline; versions unpinned by the reference => "parity
unpinned" at these two boundaries):
  * skimage.transform.SimilarityTransform.estimate == Umeyama's closed form (scale, rotation,
    translation).  `umeyama` below follows the published algorithm in float64; it is checked
    against scikit-image 0.18.3 (the /opt/conda interpreter of the build container) by
    tools/make_golden.py, vectors in tests/golden/align_umeyama.npz.
  * cv2.warpAffine(img, M, (w,h), borderValue=0) for 8-bit images, INTER_LINEAR,
    BORDER_CONSTANT: the classic OpenCV (3.x - 4.10) fixed-point kernel -- M inverted in
    double, source coordinates in 1/1024 px rounded to 1/32 px (INTER_BITS=5), bilinear
    weights as 15-bit integers, result (sum + 2^14) >> 15.  No OpenCV is installed here, so
    this restatement cannot be run against the library; it is what the HIP kernel matches
    bit for bit.
"""
import numpy as np

# align_face.py:12-48 (ArcFace 5-point templates, values are data)
CENTER_POINTS = {
    "(96, 112)": np.array([[30.2946, 51.6963], [65.5318, 51.5014], [48.0252, 71.7366],
                           [33.5493, 92.3655], [62.7299, 92.2041]], dtype=np.float32),
    "(112, 112)": np.array([[38.2946, 51.6963], [73.5318, 51.5014], [56.0252, 71.7366],
                            [41.5493, 92.3655], [70.7299, 92.2041]], dtype=np.float32),
    "(150, 150)": np.array([[51.287415, 69.23612], [98.48009, 68.97509], [75.03375, 96.075806],
                            [55.646385, 123.7038], [94.72754, 123.48763]], dtype=np.float32),
    "(160, 160)": np.array([[54.706573, 73.85186], [105.045425, 73.573425], [80.036, 102.48086],
                            [59.356144, 131.95071], [101.04271, 131.72014]], dtype=np.float32),
    "(224, 224)": np.array([[76.589195, 103.3926], [147.0636, 103.0028], [112.0504, 143.4732],
                            [83.098595, 184.731], [141.4598, 184.4082]], dtype=np.float32),
}


def umeyama(src, dst):
    """Least-squares similarity (Umeyama 1991) mapping src -> dst; returns 3x3 float64."""
    src = np.asarray(src, dtype=np.float64)
    dst = np.asarray(dst, dtype=np.float64)
    num, dim = src.shape
    src_mean, dst_mean = src.mean(axis=0), dst.mean(axis=0)
    sd, dd = src - src_mean, dst - dst_mean
    A = dd.T @ sd / num
    d = np.ones((dim,), dtype=np.float64)
    if np.linalg.det(A) < 0:
        d[dim - 1] = -1
    T = np.eye(dim + 1, dtype=np.float64)
    U, S, V = np.linalg.svd(A)
    rank = np.linalg.matrix_rank(A)
    if rank == 0:
        return np.nan * T
    if rank == dim - 1:
        if np.linalg.det(U) * np.linalg.det(V) > 0:
            T[:dim, :dim] = U @ V
        else:
            s = d[dim - 1]
            d[dim - 1] = -1
            T[:dim, :dim] = U @ np.diag(d) @ V
            d[dim - 1] = s
    else:
        T[:dim, :dim] = U @ np.diag(d) @ V
    scale = 1.0 / sd.var(axis=0).sum() * (S @ d)
    T[:dim, dim] = dst_mean - scale * (T[:dim, :dim] @ src_mean.T)
    T[:dim, :dim] *= scale
    return T


def invert_affine(M):
    """cv::warpAffine's in-place inversion of the 2x3 forward matrix (double)."""
    M = np.asarray(M, dtype=np.float64).copy()
    D = M[0, 0] * M[1, 1] - M[0, 1] * M[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[1, 1] * D, M[0, 0] * D
    M[0, 0] = A11
    M[0, 1] *= -D
    M[1, 0] *= -D
    M[1, 1] = A22
    b1 = -M[0, 0] * M[0, 2] - M[0, 1] * M[1, 2]
    b2 = -M[1, 0] * M[0, 2] - M[1, 1] * M[1, 2]
    M[0, 2], M[1, 2] = b1, b2
    return M


def _sat_int(v):
    return np.clip(np.rint(v), -2147483648.0, 2147483647.0).astype(np.int64)


def warp_affine_u8(img, M, dst_w, dst_h):
    """cv2.warpAffine(img, M, (dst_w,dst_h), borderValue=0.0) for HxWxC uint8 (see header)."""
    img = np.asarray(img)
    assert img.dtype == np.uint8 and img.ndim == 3
    H, W, C = img.shape
    Mi = invert_affine(M)
    AB_BITS, INTER_BITS = 10, 5
    AB_SCALE = 1 << AB_BITS
    TAB = 1 << INTER_BITS
    round_delta = AB_SCALE // TAB // 2
    xs = np.arange(dst_w, dtype=np.float64)
    adelta = _sat_int(Mi[0, 0] * xs * AB_SCALE)
    bdelta = _sat_int(Mi[1, 0] * xs * AB_SCALE)
    ys = np.arange(dst_h, dtype=np.float64)
    X0 = _sat_int((Mi[0, 1] * ys + Mi[0, 2]) * AB_SCALE) + round_delta
    Y0 = _sat_int((Mi[1, 1] * ys + Mi[1, 2]) * AB_SCALE) + round_delta
    X = (X0[:, None] + adelta[None, :]) >> (AB_BITS - INTER_BITS)
    Y = (Y0[:, None] + bdelta[None, :]) >> (AB_BITS - INTER_BITS)
    sx = np.clip(X >> INTER_BITS, -32768, 32767)
    sy = np.clip(Y >> INTER_BITS, -32768, 32767)
    fx = X & (TAB - 1)
    fy = Y & (TAB - 1)
    # 15-bit weights: (32-fy)(32-fx)*32 etc. are exact integers, sum == 32768
    w00 = (TAB - fy) * (TAB - fx) * 32
    w01 = (TAB - fy) * fx * 32
    w10 = fy * (TAB - fx) * 32
    w11 = fy * fx * 32

    def fetch(yy, xx):
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        v = img[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)].astype(np.int64)
        return np.where(ok[..., None], v, 0)

    acc = (fetch(sy, sx) * w00[..., None] + fetch(sy, sx + 1) * w01[..., None]
           + fetch(sy + 1, sx) * w10[..., None] + fetch(sy + 1, sx + 1) * w11[..., None])
    out = (acc + (1 << 14)) >> 15
    return np.clip(out, 0, 255).astype(np.uint8)


def alignment(img, template, landmarks, dst_w, dst_h):
    """align_face.py:51-57: estimate(landmarks -> template), warp."""
    T = umeyama(landmarks, template)
    return warp_affine_u8(img, T[0:2, :], dst_w, dst_h)


def crop_box(box, ori_h, ori_w):
    """demo_image.py:179-182 integer crop rectangle (x1,y1,x2,y2), python int() truncation."""
    x1 = max(int(box[0]), 0)
    y1 = max(int(box[1]), 0)
    x2 = min(int(box[2] + 1), ori_w)
    y2 = min(int(box[3] + 1), ori_h)
    return x1, y1, x2, y2


def detect_align_faces(rgb_image, boxes, landmarks, template, target_w, target_h):
    """demo_image.py:283-295 for one image: returns list of (target_h,target_w,3) uint8 faces."""
    H, W = rgb_image.shape[:2]
    faces = []
    for box, lm in zip(boxes, landmarks):
        x1, y1, x2, y2 = crop_box(box, H, W)
        face = rgb_image[y1:y2, x1:x2, :]
        moved = np.asarray(lm, dtype=np.float32) - np.asarray(box[:2], dtype=np.float32)
        faces.append(alignment(face, template, moved, target_w, target_h))
    return faces


def transforms_default(face_u8):
    """data_loader/__init__.py:27-34,52-56: float32 -> (x-127.5)/128 -> CHW."""
    x = (np.float32(face_u8) - 127.5) / 128
    return np.transpose(x, (2, 0, 1)).astype(np.float32)
