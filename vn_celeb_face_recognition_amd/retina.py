"""Host-side mirror of the reference's RetinaFace detector plugin, backed by libvnface.so.

  RetinaFace.__init__   <- /root/reference/models/retina_face.py:55-108 (same kwargs; cfg/detection/retina_face.json)
  RetinaFace.inference  <- /root/reference/models/retina_face.py:156-232
  RetinaFace.load_model <- /root/reference/models/retina_face.py:233-266 ('module.' prefix, optional 'state_dict' level)

The network (MobileNetV1-0.25 body, FPN, SSH, heads), the prior-box decode, top-K, py_cpu_nms and the keep / visibility
cuts all run in HIP kernels behind vnf_retina_detect; frames are uploaded once and stay resident for the alignment warp.
Only the mobilenet0.25 configuration (cfg_mnet, the one cfg/detection/retina_face.json selects) is built.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from .detector import MTCNN


def remove_prefix(state_dict, prefix):
    return {(k.split(prefix, 1)[-1] if k.startswith(prefix) else k): v for k, v in state_dict.items()}


class RetinaFace:
    """RetinaFace (mobilenet0.25) detector plugin (/root/reference/models/retina_face.py:56-232).

    Limits of the device tables (reference-legal inputs beyond them fail loudly, never truncate): at most 16 384 anchors
    per frame above `conf_thres` (VNF_E_CAPACITY beyond) and `keep_top_k` <= 768 (rejected at create).
    Without `checkpoint_path` / `state_dict` the detector is built from the generator's SYNTHETIC weights (no trained
    checkpoint exists offline; the reference's cfg points at /content/...): boxes are then meaningless -- a
    RuntimeWarning says so; pass `synthetic=True` (benchmarks, tests) to acknowledge it."""
    channels_subtract = (104, 117, 123)

    def __init__(self, backbone_cfg="cfg_mnet", phase="test", backbone_path=None, device="cuda:0", conf_thres=0.02,
                 topk_bf_nms=5000, keep_top_k=750, nms_thres=0.4, vis_thres=0.6, checkpoint_path=None, state_dict=None,
                 seed=0, max_batch=1, compute_dtype="f32", synthetic=False):
        if backbone_cfg != "cfg_mnet":
            raise NotImplementedError("RetinaFace: only backbone_cfg='cfg_mnet' (mobilenet0.25) is built for MI355X; %r "
                                      "needs the torchvision ResNet-50 download (retina_face.py:84-86)" % (backbone_cfg,))
        if phase != "test":
            raise NotImplementedError("RetinaFace: inference only (phase='test'); training the detector is out of scope")
        self.phase = phase
        self.device = torch.device(device)
        self.conf_thres = float(conf_thres)
        self.topk_bf_nms = int(topk_bf_nms)
        self.keep_top_k = int(keep_top_k)
        self.nms_thres = float(nms_thres)
        self.vis_thres = float(vis_thres)
        self._max_batch = int(max_batch)
        if compute_dtype not in ("f32", "f16x2"):
            raise ValueError("RetinaFace compute_dtype: 'f32' (exact, default) or 'f16x2' (split-f16, ~15 %% faster), got %r" % (compute_dtype,))
        self.compute_dtype = compute_dtype
        self._handle = None
        self._handle_key = None
        self._frames = None
        if state_dict is not None:
            self._sd = remove_prefix(dict(state_dict), "module.")
        elif checkpoint_path is not None:
            self.load_model(checkpoint_path)
        else:
            # no trained checkpoint exists offline (the reference's lives under /content/...): deterministic synthetic weights
            from .weights import generate_state_dict
            if not synthetic:
                import warnings
                warnings.warn("RetinaFace: no checkpoint_path / state_dict given -- running on SYNTHETIC random weights "
                              "(seed %d); detections are meaningless.  Pass checkpoint_path=... (cfg/detection/"
                              "retina_face.json) for a trained model, or synthetic=True to silence this." % seed,
                              RuntimeWarning, stacklevel=2)
            self._sd = generate_state_dict("retina", seed=seed)

    def load_model(self, pretrained_path):
        d = torch.load(pretrained_path, map_location="cpu", weights_only=True)
        if "state_dict" in d.keys():
            d = d["state_dict"]
        self._sd = remove_prefix(d, "module.")
        self._drop()

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
            raise RuntimeError("RetinaFace runs on MI355X only: construct it with device='cuda:0' (there is no CPU path)")
        self._max_batch = max(self._max_batch, b)
        dev = self.device.index if self.device.index is not None else torch.cuda.current_device()
        key = (dev, self._max_batch, h, w)
        if self._handle is not None and self._handle_key == key:
            return self._handle
        self._drop()
        lib = _lib.load()
        with torch.cuda.device(dev):
            _lib.check(lib.vnf_init(dev))
            cfg = _lib.RetinaCfg()
            cfg.height, cfg.width, cfg.max_batch = h, w, self._max_batch
            cfg.conf_thres, cfg.topk_bf_nms, cfg.nms_thres = self.conf_thres, self.topk_bf_nms, self.nms_thres
            cfg.keep_top_k, cfg.vis_thres = self.keep_top_k, self.vis_thres
            cfg.compute_dtype = 5 if self.compute_dtype == "f16x2" else 0     # VNF_F16X2 / VNF_F32
            descs, n, keep = _lib.make_descs(self._sd)
            h_ = ctypes.c_void_p()
            _lib.check(lib.vnf_retina_create(descs, n, ctypes.byref(cfg), ctypes.byref(h_)))
            del keep
        self._handle, self._handle_key = h_, key
        return h_

    _to_device_frames = MTCNN._to_device_frames   # same input forms (list of HWC arrays, (B,H,W,3) array / tensor)

    def last_frames_device(self):
        return self._frames

    def detect_device(self, frames):
        """frames: (B,H,W,3) u8 cuda.  Returns (counts list, boxes (n,4), scores (n,), points (n,5,2)) on host."""
        b, h, w, _ = frames.shape
        hd = self._ensure(b, h, w)
        lib = _lib.load()
        cap = b * self.keep_top_k       # a frame never returns more rows than keep_top_k: one call, no capacity retry
        while True:
            counts = np.zeros(b, dtype=np.int32)
            boxes = np.empty((cap, 4), dtype=np.float32)
            probs = np.empty((cap,), dtype=np.float32)
            points = np.empty((cap, 10), dtype=np.float32)
            n_out = ctypes.c_int32(0)
            with torch.cuda.device(frames.device):
                rc = lib.vnf_retina_detect(hd, ctypes.c_void_p(frames.data_ptr()), b, h, w, counts.ctypes.data,
                                           boxes.ctypes.data, probs.ctypes.data, points.ctypes.data, cap,
                                           ctypes.byref(n_out), _lib.current_stream_ptr())
            if rc == -4 and n_out.value > cap:
                cap = int(n_out.value)
                continue
            _lib.check(rc)
            n = n_out.value
            return counts.tolist(), boxes[:n], probs[:n], points[:n].reshape(n, 5, 2)

    def results_device(self, n, device=None):
        """Device-resident copy of the last detect_device(): (frame_idx, boxes, scores, points (n,10)) cuda tensors."""
        dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        fidx = torch.empty((n,), dtype=torch.int32, device=dev)
        boxes = torch.empty((n, 4), dtype=torch.float32, device=dev)
        probs = torch.empty((n,), dtype=torch.float32, device=dev)
        points = torch.empty((n, 10), dtype=torch.float32, device=dev)
        if n:
            if self._handle is None:
                raise RuntimeError("results_device(): no detection has run on this detector yet")
            with torch.cuda.device(dev):
                _lib.check(_lib.load().vnf_retina_results_device(
                    self._handle, ctypes.c_void_p(fidx.data_ptr()), ctypes.c_void_p(boxes.data_ptr()),
                    ctypes.c_void_p(probs.data_ptr()), ctypes.c_void_p(points.data_ptr()), n, _lib.current_stream_ptr()))
        return fidx, boxes, probs, points

    def inference(self, rgb_images, landmark=True):
        """retina_face.py:156-232: rgb_images = an iterable of equal-size HWC RGB images (a single image is NOT
        accepted, like the reference, which iterates its argument).  Returns lists with one entry per image:
        boxes (k,4), scores (k,), and landmarks (k,5,2) when landmark is true."""
        frames, _ = self._to_device_frames(list(rgb_images) if not isinstance(rgb_images, (np.ndarray, torch.Tensor))
                                           else rgb_images)
        if frames.dim() != 4:
            raise ValueError("RetinaFace.inference expects a batch (list) of images")
        self._frames = frames
        counts, bx, pr, pt = self.detect_device(frames)
        dets, scores, lms = [], [], []
        o = 0
        for c in counts:
            dets.append(bx[o:o + c].copy()); scores.append(pr[o:o + c].copy()); lms.append(pt[o:o + c].copy())
            o += c
        if landmark:
            return dets, scores, lms
        return dets, scores

    def debug_heads(self, b):
        """Raw head maps of the last detection: list over the 3 pyramid levels of (b, fh, fw, 32) arrays."""
        out = []
        lib = _lib.load()
        for lvl in range(3):
            dims = (ctypes.c_int32 * 2)()
            h, w = self._handle_key[2], self._handle_key[3]
            cap = b * ((h + 7) // 8 + 1) * ((w + 7) // 8 + 1) * 32
            buf = np.empty(cap, np.float32)
            _lib.check(lib.vnf_retina_debug_heads(self._handle, lvl, b, buf.ctypes.data, cap, dims))
            out.append(buf[:b * dims[0] * dims[1] * 32].reshape(b, dims[0], dims[1], 32).copy())
        return out

    def forward(self, *a, **k):
        raise NotImplementedError("RetinaFace.forward returns raw head tensors in the reference (retina_face.py:133-154); "
                                  "use inference(), or debug_heads() for the staged parity maps")

    __call__ = forward
