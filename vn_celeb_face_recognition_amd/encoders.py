"""Host-side mirrors of the reference's encoder plugins, backed by libvnface.so.

  InceptionResnetV1  <-  /root/reference/models/inception_resnet_v1.py:184-303
  iresnet100         <-  /root/reference/models/iresnet_encoder.py:64-196

Same constructor kwargs, `.to()`, `.eval()`, `load_state_dict()`, and `__call__((N,3,S,S)) ->
(N,512)` on the same device (demo_image.py:30-34, find_embedding.py:58).  All arithmetic runs in
hand-written HIP kernels; torch only owns the input/output device memory and the stream.
There is no CPU path: calling a model that is not on a CUDA(ROCm) device raises.
"""
import ctypes
import os
from collections import OrderedDict

import numpy as np
import torch

from . import _lib
from .weights import generate_state_dict

_DTYPES = {"bf16": _lib.VNF_BF16, "f16": _lib.VNF_F16, "fp16": _lib.VNF_F16, "f16x2": _lib.VNF_F16X2, "f32": _lib.VNF_F32,
           "fp32": _lib.VNF_F32, torch.bfloat16: _lib.VNF_BF16, torch.float16: _lib.VNF_F16,
           torch.float32: _lib.VNF_F32}


def _torch_home():
    # models/inception_resnet_v1.py:334-341
    return os.path.expanduser(os.getenv("TORCH_HOME", os.path.join(os.getenv("XDG_CACHE_HOME", "~/.cache"), "torch")))


def _load_checkpoint_file(path):
    """Flat state_dict or {'state_dict': ...}, loaded without executing pickled code."""
    cp = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(cp, dict) and "state_dict" in cp and isinstance(cp["state_dict"], dict):
        cp = cp["state_dict"]
    return cp


class _Encoder:
    """nn.Module-shaped wrapper around a vnf encoder handle."""
    _arch = None
    _arch_name = None
    input_size = None

    def __init__(self, device=None, compute_dtype="f16x2", max_batch=256):
        self._sd = None
        self._handle = None
        self._handle_key = None
        self.compute_dtype = compute_dtype
        self.max_batch = int(max_batch)
        self.training = False
        self.device = torch.device("cpu")
        if device is not None:
            self.to(device)

    # ---- nn.Module surface used by the reference's callers
    def eval(self):
        self.training = False
        return self

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("inference-only encoder (training is out of scope, SURVEY.md 8)")
        return self

    def to(self, device):
        self.device = torch.device(device)
        return self

    def cuda(self, device=None):
        return self.to("cuda" if device is None else device)

    def parameters(self):
        return iter(())

    def state_dict(self):
        return OrderedDict((k, torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v)
                           for k, v in self._sd.items())

    def load_state_dict(self, state_dict, strict=True):
        expected = [k for k in self._expected_keys()]
        missing = [k for k in expected if k not in state_dict]
        if strict and missing:
            raise RuntimeError("Missing key(s) in state_dict: %s" % ", ".join(missing[:8]))
        sd = OrderedDict(self._sd) if (self._sd is not None and not strict) else OrderedDict()
        for k, v in state_dict.items():
            sd[k] = v
        self._sd = sd
        self._drop_handle()
        return self

    def __del__(self):
        try:
            self._drop_handle()
        except Exception:
            pass

    def _drop_handle(self):
        if self._handle is not None:
            _lib.load().vnf_destroy(self._handle)
            self._handle = None

    def _expected_keys(self):
        return [n for n, _, kind in self._spec() if kind != "nbt"]

    def _ensure_handle(self):
        if self.device.type != "cuda":
            raise RuntimeError("%s runs on MI355X only: move it to a cuda device (there is no CPU path)"
                               % type(self).__name__)
        key = (self.device.index or 0, self.compute_dtype, self.max_batch)
        if self._handle is not None and self._handle_key == key:
            return self._handle
        self._drop_handle()
        lib = _lib.load()
        dev = self.device.index if self.device.index is not None else torch.cuda.current_device()
        with torch.cuda.device(dev):
            _lib.check(lib.vnf_init(dev))
            descs, n, keep = _lib.make_descs(self._sd)
            h = ctypes.c_void_p()
            _lib.check(lib.vnf_encoder_create(self._arch, descs, n, _DTYPES[self.compute_dtype], self.max_batch,
                                              ctypes.byref(h)))
            del keep
        self._handle, self._handle_key = h, key
        return h

    def __call__(self, x):
        return self.forward(x)

    def forward(self, x):
        h = self._ensure_handle()
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] != self.input_size or x.shape[3] != self.input_size:
            raise ValueError("expected (N,3,%d,%d) input, got %s" % (self.input_size, self.input_size, tuple(x.shape)))
        if x.device.type != "cuda":
            raise RuntimeError("input tensor must live on the encoder's cuda device")
        x = x.contiguous()
        n = x.shape[0]
        out = torch.empty((n, 512), dtype=torch.float32, device=x.device)
        lib = _lib.load()
        with torch.cuda.device(x.device):
            for n0 in range(0, n, self.max_batch):
                nn = min(self.max_batch, n - n0)
                xs = x[n0:n0 + nn]
                _lib.check(lib.vnf_embed(h, ctypes.c_void_p(xs.data_ptr()), nn, _lib.torch_dtype_code(x.dtype),
                                         ctypes.c_void_p(out[n0:].data_ptr()), _lib.current_stream_ptr()))
        return out

    def set_streams(self, max_streams):
        """Cap the encoder's internal batch split (vnf_encoder_set_streams): 1 when other work shares the GPU."""
        _lib.check(_lib.load().vnf_encoder_set_streams(self._ensure_handle(), int(max_streams)))

    def set_contexts(self, n):
        """Rotate consecutive calls over n private activation-buffer sets (vnf_encoder_set_contexts), so calls issued
        on different streams overlap on the GPU."""
        _lib.check(_lib.load().vnf_encoder_set_contexts(self._ensure_handle(), int(n)))

    def embed_stream(self, batches, lanes=3):
        """Throughput mode over an iterable of independent (N,3,S,S) batches (host or cuda tensors): batch i runs on
        stream i % lanes over the encoder's activation contexts, so up to `lanes` batches are in flight on the GPU;
        yields (index, embeddings (N,512) cuda, ready_event) in order -- wait on / synchronise the event before
        reading.  What find_embedding.py's loop over a directory becomes (find_embedding.py:44-59)."""
        dev = self.device if isinstance(self.device, torch.device) else torch.device(self.device)
        lanes = max(1, min(4, int(lanes)))
        if lanes > 1:
            self.set_streams(1)
            self.set_contexts(lanes)
        from .streams import side_streams
        streams = side_streams(dev, lanes)      # process-wide stream objects (streams.py: hardware-queue binding)
        start = torch.cuda.current_stream(dev).record_event()
        for s_ in streams:
            s_.wait_event(start)
        inflight = []
        for i, x in enumerate(batches):
            s_ = streams[i % lanes]
            with torch.cuda.stream(s_):
                x = x.to(dev, non_blocking=True)
                emb = self(x)
                ev = s_.record_event()
            x.record_stream(s_)
            inflight.append((i, emb, ev))
            if len(inflight) >= lanes:
                yield inflight.pop(0)
        for item in inflight:
            yield item

    # ---- extras used by tests / bench
    def tap(self, name, n):
        """Copy an internal activation of the last forward (first n images) to a (n,C,H,W) fp32 array."""
        h = self._ensure_handle()
        lib = _lib.load()
        shape = (ctypes.c_int64 * 4)()
        lib.vnf_encoder_tap(h, name.encode(), n, None, 0, shape)  # query shape (returns capacity error)
        total = int(shape[0] * shape[1] * shape[2] * shape[3])
        if total <= 0:
            _lib.check(lib.vnf_encoder_tap(h, name.encode(), n, None, 0, shape))
        buf = np.empty(total, dtype=np.float32)
        torch.cuda.synchronize()
        _lib.check(lib.vnf_encoder_tap(h, name.encode(), n, buf.ctypes.data, total, shape))
        return buf.reshape(tuple(int(s) for s in shape))

    def profile(self, x):
        """Per-launch device-time table of one forward (text)."""
        h = self._ensure_handle()
        x = x.contiguous()
        out = torch.empty((x.shape[0], 512), dtype=torch.float32, device=x.device)
        buf = ctypes.create_string_buffer(1 << 16)
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().vnf_encoder_profile(h, ctypes.c_void_p(x.data_ptr()), x.shape[0],
                                                       _lib.torch_dtype_code(x.dtype), ctypes.c_void_p(out.data_ptr()),
                                                       _lib.current_stream_ptr(), buf, len(buf)))
        return buf.value.decode()

    def flops_per_image(self):
        h = self._ensure_handle()
        a, e = ctypes.c_double(), ctypes.c_double()
        _lib.check(_lib.load().vnf_encoder_flops(h, ctypes.byref(a), ctypes.byref(e)))
        return a.value, e.value


class InceptionResnetV1(_Encoder):
    """Drop-in for models.InceptionResnetV1 (inception_resnet_v1.py:202).

    compute_dtype (build extension; may also be given in the -eargs JSON): "f16x2" (default) is the parity path --
    split-f16 operands on the 16-bit MFMA, <= 1e-4 embedding L2 against the reference (measured ~1.5e-6), the same
    default in every CLI so embeddings the MLP is trained on and embeddings it classifies come from one arithmetic;
    "bf16" / "f16" trade accuracy (5e-3 / 6e-4) for 2.8x the throughput; "f32" is the exact fp32 fma chain.

    pretrained: None -> deterministic generator weights (seed 0; the reference would leave
    torch's random init); 'vggface2' / 'casia-webface' -> the file the reference caches under
    $TORCH_HOME/checkpoints (never downloaded here); or a path to a local state_dict file
    (build extension).
    """
    _arch = _lib.VNF_ARCH_IRV1
    input_size = 160
    _FILES = {"vggface2": "20180402-114759-vggface2.pt", "casia-webface": "20180408-102900-casia-webface.pt"}

    def __init__(self, pretrained=None, classify=False, num_classes=None, dropout_prob=0.6, device=None,
                 compute_dtype="f16x2", max_batch=256, seed=0):
        if classify:
            raise NotImplementedError("classify=True (logits head) is not on the inference hot path")
        self.pretrained = pretrained
        self.classify = classify
        self.num_classes = num_classes
        super().__init__(device=None, compute_dtype=compute_dtype, max_batch=max_batch)
        if pretrained is None:
            self._sd = generate_state_dict("irv1", seed)
        else:
            path = pretrained
            if pretrained in self._FILES:
                path = os.path.join(_torch_home(), "checkpoints", self._FILES[pretrained])
            if not os.path.exists(path):
                raise FileNotFoundError(
                    "pretrained weights %r not found at %s (no network: place the file there or pass a local path)"
                    % (pretrained, path))
            self.load_state_dict(_load_checkpoint_file(path), strict=False)
        if device is not None:
            self.to(device)

    def _spec(self):
        from .weights import irv1_spec
        return irv1_spec()


class _IResNet100(_Encoder):
    _arch = _lib.VNF_ARCH_IR100
    input_size = 112

    def _spec(self):
        from .weights import iresnet_spec
        return iresnet_spec()


def iresnet100(pretrained=False, progress=True, freeze_weights=False, checkpoint_path="", compute_dtype="f16x2",
               max_batch=256, seed=0, **kwargs):
    """Drop-in for models.iresnet100 (iresnet_encoder.py:162-181,194-196); kwargs of
    cfg/embedding/iresnet100_enc.json.  pretrained=True needs checkpoint_path (a file holding
    {'state_dict': ...}); the URL branch of the reference cannot run offline."""
    if kwargs:
        raise TypeError("unexpected keyword arguments: %s" % sorted(kwargs))
    m = _IResNet100(compute_dtype=compute_dtype, max_batch=max_batch)
    m._sd = generate_state_dict("iresnet100", seed)
    if pretrained:
        if not checkpoint_path:
            raise FileNotFoundError("iresnet100(pretrained=True) needs checkpoint_path: no network access")
        print("Loaded encoder state dict from checkpoint path {}".format(checkpoint_path))
        m.load_state_dict(_load_checkpoint_file(checkpoint_path), strict=False)
    return m
