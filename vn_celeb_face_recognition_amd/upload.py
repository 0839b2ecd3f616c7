"""Asynchronous frame upload for real streams: decode -> BGR->RGB -> device upload (/root/reference/demo_video.py:86-118,
models/mtcnn_utils/detect_face.py:26-46).  The reference stacks the decoded frames and `.to(device)`s them inline;
here a batch goes host -> HBM behind the GPU work of the batches before it:

  * a ring of PINNED host buffers, one (B,H,W,3) u8 slot per batch in flight.  A decoder that can write into a caller
    buffer fills `slot()` directly (`FrameUploader.slot` / `commit`); frames that already exist as arrays are copied in
    by a few host threads (one memcpy per frame: it replaces the reference's np.stack, it is not an extra pass);
  * one copy stream per device (streams.side_stream role 6): `hipMemcpyAsync` pinned -> device, then an event.  The
    detector waits for that event on ITS stream (`FacePipeline.submit(frames, ready=event)`); the host never blocks on
    the transfer, only -- when the ring wraps -- on the copy that used the slot `depth` batches ago;
  * a ring of DEVICE buffers as well: a batch is 100 MB at 16 x 1080p, and a fresh allocation per batch that several
    streams still hold (record_stream) keeps the caching allocator from recycling -- it falls back to hipMalloc, which
    synchronises the device.  The caller hands a slot back with `release(slot, event)` (event = its last consumer, e.g.
    the ticket's embedding event); the next copy into that slot waits for the event ON THE COPY STREAM.  A slot that
    was never released is simply not reused (that upload gets a fresh tensor, slot -1).
"""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

UPLOAD_STREAM_ROLE = 6     # streams.py roles: 0 detection, 1.. embedding lanes / extra detectors, 6 upload, 7 collective


class FrameUploader:
    def __init__(self, device, depth=3, threads=4):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("FrameUploader needs a cuda device (there is no CPU path)")
        from .streams import side_stream
        self.stream = side_stream(self.device, UPLOAD_STREAM_ROLE)
        self.depth = max(2, int(depth))
        self._ring = [None] * self.depth       # pinned (B,H,W,3) u8 tensors
        self._busy = [None] * self.depth       # event of the last H2D copy out of each slot
        self._next = 0
        self._dev = [None] * self.depth        # device (B,H,W,3) u8 buffers (flat), reused
        self._dev_state = [None] * self.depth  # None: free | "out": handed out, not released | event: released behind it
        self._dnext = 0
        self._pool = ThreadPoolExecutor(max_workers=max(1, int(threads))) if threads > 1 else None
        self.bytes = 0                         # uploaded so far (bench / tests)
        self.last_slot = -1                    # device slot of the last upload (for release())

    def slot(self, shape):
        """The next pinned (B,H,W,3) u8 staging buffer (numpy view + ring index), free to be written: waits only for
        the copy that read it `depth` uploads ago."""
        k = self._next
        self._next = (k + 1) % self.depth
        if self._busy[k] is not None:
            self._busy[k].synchronize()
            self._busy[k] = None
        shape = tuple(int(s) for s in shape)
        n = int(np.prod(shape))
        buf = self._ring[k]
        if buf is None or buf.numel() < n:
            buf = self._ring[k] = torch.empty((n,), dtype=torch.uint8).pin_memory()
        return buf[:n].view(shape), k

    def _device_slot(self, shape):
        """(device tensor, slot) for the next upload: a ring slot when it is free or was released, else a fresh tensor."""
        n = int(np.prod(shape))
        k = self._dnext
        st = self._dev_state[k]
        if st == "out":                        # still with its consumer and never released: do not touch it
            with torch.cuda.stream(self.stream):
                return torch.empty(shape, dtype=torch.uint8, device=self.device), -1
        self._dnext = (k + 1) % self.depth
        if st is not None:
            self.stream.wait_event(st)         # the previous consumer of this slot, on the copy stream (no host wait)
        buf = self._dev[k]
        if buf is None or buf.numel() < n:
            with torch.cuda.stream(self.stream):
                buf = self._dev[k] = torch.empty((n,), dtype=torch.uint8, device=self.device)
        self._dev_state[k] = "out"
        return buf[:n].view(shape), k

    def release(self, slot, event):
        """The consumer of device slot `slot` is done once `event` has passed (None: it is done now)."""
        if slot is not None and slot >= 0:
            self._dev_state[slot] = event

    def commit(self, staged, k):
        """Enqueue pinned -> device for staging slot k; returns ((B,H,W,3) u8 cuda tensor, event that marks it resident,
        device slot for release())."""
        dev, slot = self._device_slot(tuple(staged.shape))
        with torch.cuda.stream(self.stream):
            dev.copy_(staged, non_blocking=True)
            ev = self.stream.record_event()
        if k is not None:
            self._busy[k] = ev
        self.bytes += staged.numel()
        self.last_slot = slot
        return dev, ev

    def upload(self, frames):
        """frames: list of equal-size HWC u8 arrays, or a (B,H,W,3) u8 array / host tensor (pinned tensors skip the
        staging copy).  Returns (frames_dev, ready_event); `last_slot` is the device slot to release() once the batch's
        last consumer is enqueued."""
        if isinstance(frames, torch.Tensor):
            if frames.is_cuda:
                return frames, torch.cuda.current_stream(self.device).record_event()
            if frames.is_pinned() and frames.dtype == torch.uint8 and frames.is_contiguous():
                return self.commit(frames, None)      # already in a pinned buffer (a decoder's own ring): no staging copy
            frames = frames.numpy()
        if isinstance(frames, np.ndarray):
            if frames.ndim == 3:
                frames = frames[None]
            arrs = list(frames)
        else:
            arrs = [np.asarray(f) for f in frames]
        if not arrs:
            raise ValueError("upload(): empty batch")
        if any(a.shape != arrs[0].shape for a in arrs):
            raise Exception("MTCNN batch processing only compatible with equal-dimension images.")   # detect_face.py:33-34
        if arrs[0].ndim != 3 or arrs[0].shape[2] != 3:
            raise ValueError("expected HWC RGB images, got shape %s" % (arrs[0].shape,))
        staged, k = self.slot((len(arrs),) + arrs[0].shape)
        view = staged.numpy()

        def put(i):
            np.copyto(view[i], arrs[i], casting="unsafe")      # np.uint8(a) of detect_face.py:36

        if self._pool is not None and len(arrs) > 1 and arrs[0].size >= (1 << 18):
            list(self._pool.map(put, range(len(arrs))))
        else:
            for i in range(len(arrs)):
                put(i)
        return self.commit(staged, k)

    def close(self):
        if self._pool is not None:
            self._pool.shutdown(wait=True)
            self._pool = None
