"""Host-side mirror of the reference's MTCNN detector plugin, backed by libvnface.so.

  MTCNN.__init__   <- /root/reference/models/mtcnn.py:200-227 (same kwargs; cfg/detection/mtcnn.json)
  MTCNN.detect     <- /root/reference/models/mtcnn.py:278-361
  MTCNN.inference  <- /root/reference/models/mtcnn.py:511-513
  input handling   <- /root/reference/models/mtcnn_utils/detect_face.py:26-46

The cascade itself (pyramid, P/R/O-Net, NMS, crop/resize, box arithmetic) runs in HIP kernels
behind vnf_mtcnn_detect; frames are uploaded once and stay resident for the alignment warp
(`last_frames_device`).  Results are returned per image as lists of arrays -- the reference's
np.array() of ragged per-image lists raises on NumPy >= 1.24.
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib

_WEIGHTS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "weights_mtcnn")


def _load_net(name):
    # models/mtcnn.py:32-36 (state_dict files vendored from facenet-pytorch, MIT)
    return torch.load(os.path.join(_WEIGHTS, name + ".pt"), map_location="cpu", weights_only=True)


class MTCNN:
    def __init__(self, image_size=160, margin=0, min_face_size=20, thresholds=[0.6, 0.7, 0.7], factor=0.709,
                 post_process=True, select_largest=True, selection_method=None, keep_all=False, device=None,
                 max_batch=16, max_height=1080, max_width=1920, state_dicts=None, max_candidates=0):
        self.image_size = image_size
        self.margin = margin
        self.min_face_size = int(min_face_size)
        self.thresholds = [float(t) for t in thresholds]
        self.factor = float(factor)
        self.post_process = post_process
        self.select_largest = select_largest
        self.keep_all = keep_all
        self.selection_method = selection_method or ('largest' if select_largest else 'probability')
        self.training = False
        self._sd = state_dicts or tuple(_load_net(n) for n in ("pnet", "rnet", "onet"))
        self._cap = [int(max_batch), int(max_height), int(max_width)]
        self._max_candidates = int(max_candidates)    # vnf_mtcnn_cfg.max_candidates: rows per frame of the stage tables (0: 2048); grows on overflow
        self._handle = None
        self._handle_key = None
        self._frames = None
        self.device = torch.device('cpu')
        if device is not None:
            self.to(device)

    def eval(self):
        return self

    def to(self, device):
        self.device = torch.device(device)
        return self

    def __del__(self):
        try:
            self._drop()
        except Exception:
            pass

    def _drop(self):
        if self._handle is not None:
            _lib.load().vnf_destroy(self._handle)
            self._handle = None

    def _ensure(self, b, h, w):
        if self.device.type != "cuda":
            raise RuntimeError("MTCNN runs on MI355X only: construct it with device='cuda:0' (there is no CPU path)")
        self._cap = [max(self._cap[0], b), max(self._cap[1], h), max(self._cap[2], w)]
        dev = self.device.index if self.device.index is not None else torch.cuda.current_device()
        key = (dev, tuple(self._cap))
        if self._handle is not None and self._handle_key == key:
            return self._handle
        self._drop()
        lib = _lib.load()
        with torch.cuda.device(dev):
            _lib.check(lib.vnf_init(dev))
            cfg = _lib.MtcnnCfg()
            cfg.min_face_size = self.min_face_size
            for i in range(3):
                cfg.thresholds[i] = self.thresholds[i]
            cfg.factor = self.factor
            cfg.select_largest = 1 if self.select_largest else 0
            cfg.max_batch, cfg.max_height, cfg.max_width = self._cap
            cfg.max_candidates = self._max_candidates
            (dp, np_, kp), (dr, nr, kr), (do, no, ko) = (_lib.make_descs(sd) for sd in self._sd)
            h_ = ctypes.c_void_p()
            _lib.check(lib.vnf_mtcnn_create(dp, np_, dr, nr, do, no, ctypes.byref(cfg), ctypes.byref(h_)))
            del kp, kr, ko
        self._handle, self._handle_key = h_, key
        return h_

    # ---- input handling (detect_face.py:26-46)
    def _to_device_frames(self, img):
        single = False
        if isinstance(img, torch.Tensor):
            t = img
            if t.dim() == 3:
                t, single = t.unsqueeze(0), True
        elif isinstance(img, np.ndarray):
            a = img
            if a.ndim == 3:
                a, single = a[None], True
            t = torch.from_numpy(np.ascontiguousarray(a))
        else:
            if not isinstance(img, (list, tuple)):
                img, single = [img], True
            arrs = [np.asarray(i) for i in img]
            if any(a.shape != arrs[0].shape for a in arrs):
                raise Exception("MTCNN batch processing only compatible with equal-dimension images.")
            t = torch.from_numpy(np.stack([np.uint8(a) for a in arrs]))
        if t.dim() != 4 or t.shape[3] != 3:
            raise ValueError("expected HWC RGB images, got shape %s" % (tuple(t.shape),))
        if t.dtype != torch.uint8:
            t = t.to(torch.uint8)
        return t.to(self.device, non_blocking=False).contiguous(), single

    def last_frames_device(self):
        """(B,H,W,3) uint8 cuda tensor of the frames of the last detect() call (kept for the warp)."""
        return self._frames

    def detect_device(self, frames):
        """frames: (B,H,W,3) u8 cuda.  Returns (counts list, boxes (n,4), probs (n,), points (n,5,2)) on host."""
        b, h, w, _ = frames.shape
        hd = self._ensure(b, h, w)
        lib = _lib.load()
        cap = 256
        while True:
            counts = np.zeros(b, dtype=np.int32)
            boxes = np.empty((cap, 4), dtype=np.float32)
            probs = np.empty((cap,), dtype=np.float32)
            points = np.empty((cap, 10), dtype=np.float32)
            n_out = ctypes.c_int32(0)
            with torch.cuda.device(frames.device):
                rc = lib.vnf_mtcnn_detect(hd, ctypes.c_void_p(frames.data_ptr()), b, h, w, counts.ctypes.data,
                                          boxes.ctypes.data, probs.ctypes.data, points.ctypes.data, cap,
                                          ctypes.byref(n_out), _lib.current_stream_ptr())
            if rc == -4 and n_out.value > cap:
                cap = int(n_out.value)
                continue
            if rc == -4 and b"candidate table overflow" in lib.vnf_last_error() and self._max_candidates < (1 << 20):
                # a frame with more stage-1 survivors than the stage tables have rows (the reference has no cap,
                # detect_face.py:79-93): grow the tables and run the batch again -- like a vector, never a truncation
                self._max_candidates = max(4096, 2 * max(self._max_candidates, 2048))
                self._handle_key = None
                hd = self._ensure(b, h, w)
                continue
            _lib.check(rc)
            n = n_out.value
            return counts.tolist(), boxes[:n], probs[:n], points[:n].reshape(n, 5, 2)

    def stage_times(self, frames, reps=5):
        """Per-stage device time of one detection (vnf_mtcnn_stage_times: HIP events between the stages on the
        current stream).  frames: (B,H,W,3) u8 cuda.  Returns {stage: {"ms": median over reps, "bytes": algorithmic
        bytes of the launch}}; kernels stages carry their kernel's name (pyramid, pnet_conv1_pool, ...)."""
        b, h, w, _ = frames.shape
        hd = self._ensure(b, h, w)
        lib = _lib.load()
        acc = {}
        for _ in range(reps):
            buf = ctypes.create_string_buffer(1 << 14)
            with torch.cuda.device(frames.device):
                _lib.check(lib.vnf_mtcnn_stage_times(hd, ctypes.c_void_p(frames.data_ptr()), b, h, w, buf, len(buf),
                                                     _lib.current_stream_ptr()))
            seen = {}
            for line in buf.value.decode().splitlines():
                name, ms, nbytes = line.split()
                e = seen.setdefault(name, [0.0, 0.0])     # chunked stages repeat: sum within one call
                e[0] += float(ms)
                e[1] += float(nbytes)
            for name, (ms, nbytes) in seen.items():
                acc.setdefault(name, {"ms": [], "bytes": nbytes})["ms"].append(ms)
        return {k: {"ms": float(np.median(v["ms"])), "bytes": v["bytes"]} for k, v in acc.items()}

    def debug_stage3(self, boxes, onet_out):
        """O-stage decode alone (vnf_mtcnn_debug_stage3) on a caller-made single-frame candidate table: boxes (n,4)
        before bbreg, onet_out (n,15) [prob, reg x4, landmark x x5, landmark y x5] -> (boxes (k,4), probs (k,),
        points (k,5,2)) after threshold, bbreg, "Min" NMS and the area ordering.  Test hook for tied scores."""
        boxes = np.ascontiguousarray(boxes, dtype=np.float32).reshape(-1, 4)
        oo = np.ascontiguousarray(onet_out, dtype=np.float32).reshape(-1, 15)
        n = boxes.shape[0]
        hd = self._ensure(1, 64, 64)
        fin = np.empty((max(n, 1), 15), dtype=np.float32)
        n_out = ctypes.c_int32(0)
        dev = self.device.index if self.device.index is not None else torch.cuda.current_device()
        with torch.cuda.device(dev):
            _lib.check(_lib.load().vnf_mtcnn_debug_stage3(hd, boxes.ctypes.data, oo.ctypes.data, n, fin.ctypes.data,
                                                          fin.shape[0], ctypes.byref(n_out), _lib.current_stream_ptr()))
        fin = fin[:n_out.value]
        return fin[:, :4].copy(), fin[:, 4].copy(), fin[:, 5:].reshape(-1, 5, 2).copy()

    def _require_handle(self):
        if self._handle is None:
            raise RuntimeError("results_device(): no detection has run on this detector yet")
        return self._handle

    def results_device(self, n, device=None):
        """Device-resident copy of the last detect_device() on this handle: (frame_idx (n,) int32, boxes (n,4),
        probs (n,), points (n,10)) cuda tensors filled on the current stream -- the inputs of vnf_align, without
        the host round trip."""
        dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        fidx = torch.empty((n,), dtype=torch.int32, device=dev)
        boxes = torch.empty((n, 4), dtype=torch.float32, device=dev)
        probs = torch.empty((n,), dtype=torch.float32, device=dev)
        points = torch.empty((n, 10), dtype=torch.float32, device=dev)
        if n:
            with torch.cuda.device(dev):
                _lib.check(_lib.load().vnf_mtcnn_results_device(
                    self._require_handle(), ctypes.c_void_p(fidx.data_ptr()), ctypes.c_void_p(boxes.data_ptr()),
                    ctypes.c_void_p(probs.data_ptr()), ctypes.c_void_p(points.data_ptr()), n, _lib.current_stream_ptr()))
        return fidx, boxes, probs, points

    def detect(self, img, landmarks=False):
        frames, single = self._to_device_frames(img)
        self._frames = frames
        counts, bx, pr, pt = self.detect_device(frames)
        boxes, probs, points = [], [], []
        o = 0
        for c in counts:
            if c == 0:
                boxes.append([]); probs.append([]); points.append([])   # mtcnn.py:330-333
            else:
                boxes.append(bx[o:o + c].copy()); probs.append(pr[o:o + c].copy()); points.append(pt[o:o + c].copy())
            o += c
        if single:
            boxes, probs, points = boxes[0], probs[0], points[0]
        if landmarks:
            return boxes, probs, points
        return boxes, probs

    def inference(self, rgb_image, landmark=True):
        return self.detect(rgb_image, landmark)

    def forward(self, *a, **k):
        raise NotImplementedError("MTCNN.forward (crop extraction, mtcnn.py:229-276) is not used by the demos' hot path; "
                                  "use inference() + the alignment warp")

    __call__ = forward

    def debug_pnet_level(self, img, level):
        """Dense pyramid level, P-Net prob and reg maps of one level for one image (staged parity tests)."""
        frames, _ = self._to_device_frames(img)
        b, h, w, _ = frames.shape
        hd = self._ensure(1, h, w)
        lib = _lib.load()
        big = h * w + 16
        lvl = np.empty(3 * big, np.float32); prob = np.empty(big, np.float32); reg = np.empty(4 * big, np.float32)
        dims = (ctypes.c_int32 * 4)()
        with torch.cuda.device(frames.device):
            _lib.check(lib.vnf_mtcnn_debug_pnet(hd, ctypes.c_void_p(frames.data_ptr()), h, w, level, lvl.ctypes.data,
                                                prob.ctypes.data, reg.ctypes.data, dims, _lib.current_stream_ptr()))
        hs, ws, oh, ow = (int(d) for d in dims)
        return (lvl[:3 * hs * ws].reshape(3, hs, ws), prob[:oh * ow].reshape(oh, ow), reg[:4 * oh * ow].reshape(4, oh, ow))
