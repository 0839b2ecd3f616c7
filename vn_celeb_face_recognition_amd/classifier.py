"""Host-side mirror of the reference's MLP identity classifier, backed by libvnface.so.

  MLPModel            <-  /root/reference/models/mlp_model.py:4-15
  load_model_classify <-  /root/reference/demo_image.py:16-21   (checkpoint dict of
                          trainer/base_trainer.py:91-98: needs 'epoch' and 'state_dict')
"""
import ctypes
from collections import OrderedDict

import numpy as np
import torch

from . import _lib
from .weights import generate_state_dict


class MLPModel:
    """MLPModel(input_dim, num_classes): __call__((F,input_dim)) -> (F,num_classes) log-probs."""

    def __init__(self, input_dim, num_classes, max_batch=1024, seed=0):
        self.input_dim = int(input_dim)
        self.num_classes = int(num_classes)
        self.max_batch = int(max_batch)
        self.training = False
        self.device = torch.device("cpu")
        self._handle = None
        self._handle_dev = None
        # the reference starts from torch's random init; a deterministic generator stands in
        self._sd = generate_state_dict("mlp", seed, input_dim=self.input_dim, num_classes=self.num_classes)

    def eval(self):
        self.training = False
        return self

    def to(self, device):
        self.device = torch.device(device)
        return self

    def state_dict(self):
        return OrderedDict((k, torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v)
                           for k, v in self._sd.items())

    def load_state_dict(self, state_dict, strict=True):
        want = {"dense_1.weight": (2048, self.input_dim), "dense_1.bias": (2048,),
                "dense_2.weight": (self.num_classes, 2048), "dense_2.bias": (self.num_classes,)}
        for k, shp in want.items():
            if k not in state_dict:
                raise RuntimeError("Missing key(s) in state_dict: %s" % k)
            if tuple(state_dict[k].shape) != shp:
                raise RuntimeError("size mismatch for %s: %s vs %s" % (k, tuple(state_dict[k].shape), shp))
        if strict and set(state_dict) - set(want):
            raise RuntimeError("Unexpected key(s) in state_dict: %s" % sorted(set(state_dict) - set(want)))
        self._sd = OrderedDict((k, state_dict[k]) for k in want)
        self._drop()
        return self

    def _drop(self):
        if self._handle is not None:
            _lib.load().vnf_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._drop()
        except Exception:
            pass

    def _ensure(self):
        if self.device.type != "cuda":
            raise RuntimeError("MLPModel runs on MI355X only: move it to a cuda device (there is no CPU path)")
        dev = self.device.index if self.device.index is not None else torch.cuda.current_device()
        if self._handle is not None and self._handle_dev == dev:
            return self._handle
        self._drop()
        lib = _lib.load()
        with torch.cuda.device(dev):
            _lib.check(lib.vnf_init(dev))
            descs, n, keep = _lib.make_descs(self._sd)
            h = ctypes.c_void_p()
            _lib.check(lib.vnf_mlp_create(descs, n, self.input_dim, self.num_classes, self.max_batch, ctypes.byref(h)))
            del keep
        self._handle, self._handle_dev = h, dev
        return h

    def classify(self, emb, want_logp=True):
        """(F,input_dim) fp32 cuda -> (logp (F,C) or None, argmax (F,) int32, prob (F,) fp32)."""
        h = self._ensure()
        if emb.device.type != "cuda":
            raise RuntimeError("embeddings must live on the classifier's cuda device")
        emb = emb.float().contiguous()
        f = emb.shape[0]
        logp = torch.empty((f, self.num_classes), dtype=torch.float32, device=emb.device) if want_logp else None
        amax = torch.empty((f,), dtype=torch.int32, device=emb.device)
        prob = torch.empty((f,), dtype=torch.float32, device=emb.device)
        lib = _lib.load()
        with torch.cuda.device(emb.device):
            for f0 in range(0, f, self.max_batch):
                nn = min(self.max_batch, f - f0)
                _lib.check(lib.vnf_classify(h, ctypes.c_void_p(emb[f0:].data_ptr()), nn,
                                            ctypes.c_void_p(logp[f0:].data_ptr()) if want_logp else None,
                                            ctypes.c_void_p(amax[f0:].data_ptr()), ctypes.c_void_p(prob[f0:].data_ptr()),
                                            _lib.current_stream_ptr()))
        return logp, amax, prob

    def forward(self, emb):
        return self.classify(emb, want_logp=True)[0]

    __call__ = forward


def load_model_classify(checkpoint_path, model):
    """demo_image.py:16-21.  The checkpoint is a plain dict (epoch, state_dict, optimizer state,
    config); weights_only loading refuses anything that would execute code."""
    cp = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
    print("Loading checkpoint: {} ... after training for {} epochs.".format(checkpoint_path, cp['epoch']))
    model.load_state_dict(cp['state_dict'])
    return model
