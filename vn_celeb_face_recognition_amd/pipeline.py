"""Host-side mirror of the reference's pipeline glue (same names, argument meaning and error
behaviour), with the arithmetic in libvnface.so:

  center_point_dict, alignment      <- /root/reference/align_face.py:12-48, 51-57
  transforms_default                <- /root/reference/data_loader/__init__.py:27-34,52-56
  find_embedding                    <- /root/reference/demo_image.py:30-34
  recognize_celeb                   <- /root/reference/demo_image.py:50-76
  identify_person                   <- /root/reference/demo_image.py:113-147
  get_face_from_boxes               <- /root/reference/demo_image.py:174-199
  move_landmark_to_box              <- /root/reference/demo_image.py:236-239
  parallel_detect_and_align         <- /root/reference/demo_image.py:273-306

`FacePipeline` is the build's resident fast path: frames are uploaded once, boxes, landmarks,
aligned faces and embeddings stay in HBM (the reference round-trips through the host between
detection, OpenCV alignment and embedding: SURVEY.md 3.1), only names / boxes come back.
"""
import ctypes
import threading

import numpy as np
import torch

from . import _lib

# align_face.py:12-48 (ArcFace 5-point templates; values are data)
center_point_dict = {
    '(96, 112)': np.array([[30.2946, 51.6963], [65.5318, 51.5014], [48.0252, 71.7366],
                           [33.5493, 92.3655], [62.7299, 92.2041]], dtype=np.float32),
    '(112, 112)': np.array([[38.2946, 51.6963], [73.5318, 51.5014], [56.0252, 71.7366],
                            [41.5493, 92.3655], [70.7299, 92.2041]], dtype=np.float32),
    '(150, 150)': np.array([[51.287415, 69.23612], [98.48009, 68.97509], [75.03375, 96.075806],
                            [55.646385, 123.7038], [94.72754, 123.48763]], dtype=np.float32),
    '(160, 160)': np.array([[54.706573, 73.85186], [105.045425, 73.573425], [80.036, 102.48086],
                            [59.356144, 131.95071], [101.04271, 131.72014]], dtype=np.float32),
    '(224, 224)': np.array([[76.589195, 103.3926], [147.0636, 103.0028], [112.0504, 143.4732],
                            [83.098595, 184.731], [141.4598, 184.4082]], dtype=np.float32),
}


def _dev(device=None):
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("the alignment / recognition path runs on MI355X only (no CPU path)")
    return device


def align_faces_device(frames_dev, frame_idx, boxes, points, template, size, want_u8=True, norm_dtype=None,
                       out_norm=None):
    """vnf_align on resident frames.  frames_dev: (B,H,W,3) u8 cuda; frame_idx (n,) int32, boxes (n,4),
    points (n,5,2) fp32 (cuda or host).  Returns (faces_u8 (n,S,S,3) or None, faces_norm (n,3,S,S) or None)."""
    dev = frames_dev.device
    n = int(boxes.shape[0])
    B, H, W, _ = frames_dev.shape
    fi = torch.as_tensor(frame_idx, dtype=torch.int32).to(dev).contiguous()
    bx = torch.as_tensor(boxes, dtype=torch.float32).to(dev).contiguous().view(n, 4)
    pt = torch.as_tensor(points, dtype=torch.float32).to(dev).contiguous().view(n, 10)
    tm = np.ascontiguousarray(template, dtype=np.float32).reshape(10)
    u8 = torch.empty((n, size, size, 3), dtype=torch.uint8, device=dev) if want_u8 else None
    if out_norm is not None:     # caller-owned (n,3,S,S) destination, e.g. a slice of an accumulation buffer
        nm, norm_dtype = out_norm, out_norm.dtype
    else:
        nm = torch.empty((n, 3, size, size), dtype=norm_dtype, device=dev) if norm_dtype is not None else None
    if n:
        with torch.cuda.device(dev):
            _lib.check(_lib.load().vnf_align(
                ctypes.c_void_p(frames_dev.data_ptr()), B, H, W, ctypes.c_void_p(fi.data_ptr()),
                ctypes.c_void_p(bx.data_ptr()), ctypes.c_void_p(pt.data_ptr()), n, tm.ctypes.data, size,
                ctypes.c_void_p(u8.data_ptr()) if want_u8 else None,
                ctypes.c_void_p(nm.data_ptr()) if nm is not None else None,
                _lib.torch_dtype_code(norm_dtype) if nm is not None else 0, _lib.current_stream_ptr()))
    return u8, nm


def alignment(cv_img, src, dst, dst_w, dst_h, device=None):
    """align_face.py:51-57: warp `cv_img` (HxWx3 u8) so landmarks `dst` land on template `src`."""
    if dst_w != dst_h:
        raise NotImplementedError("square targets only (every template of center_point_dict the demos use)")
    dev = _dev(device)
    img = torch.from_numpy(np.ascontiguousarray(cv_img)).to(dev).unsqueeze(0)
    h, w = cv_img.shape[:2]
    # the whole image is the crop: box (0,0,w-1,h-1) -> crop [0,w) x [0,h), landmark shift 0
    box = np.array([[0.0, 0.0, w - 1, h - 1]], dtype=np.float32)
    u8, _ = align_faces_device(img, np.zeros(1, np.int32), box, np.asarray(dst, np.float32).reshape(1, 5, 2), src, dst_w)
    return u8[0].cpu().numpy()


def transforms_default(face_u8):
    """data_loader/__init__.py:27-34,52-56 (host tensor, as the reference returns)."""
    x = (np.float32(face_u8) - 127.5) / 128
    return torch.from_numpy(np.ascontiguousarray(np.transpose(x, (2, 0, 1))))


def find_embedding(image_tensor, embedding_model):
    embedding_model.eval()
    with torch.no_grad():
        embeddings = embedding_model(image_tensor)
    return embeddings.detach()


def identify_names(preds, probs, n_classes, name_df, threshold):
    """demo_image.py:119-147 after the argmax: threshold (float, or dict str(class) -> float as in
    celeb_statistic.py:128-136), then the first label2name row of each kept label, else 'Unknown'."""
    preds = preds.cpu().numpy() if hasattr(preds, "cpu") else np.asarray(preds)
    probs = probs.cpu().numpy() if hasattr(probs, "cpu") else np.asarray(probs)
    filtered = []
    for p, pr in zip(preds, probs):
        thr = threshold if type(threshold) is float else threshold[str(int(p))]
        filtered.append(int(p) if pr >= thr else n_classes)
    first = {}
    for l, nm in zip(list(name_df['label']), list(name_df['name'])):
        first.setdefault(int(l), nm)
    return [first.get(p, 'Unknown') for p in filtered]


def identify_person(embeddings, classify_model, name_df, threshold):
    """demo_image.py:113-147.  name_df: anything with ['label'] / ['name'] columns (pandas
    DataFrame or dict of sequences)."""
    classify_model.eval()
    logp, amax, prob = classify_model.classify(embeddings, want_logp=False)
    return identify_names(amax, prob, classify_model.num_classes, name_df, threshold)


def recognize_celeb(bth_alg_face_list, device, emb_model, classify_model, transforms, label2name_df, threshold):
    """demo_image.py:50-76."""
    alg_face_list = []
    for x in bth_alg_face_list:
        alg_face_list += x
    tf_list = [transforms(face) for face in alg_face_list]
    if len(tf_list) > 0:
        aligned_faces_tf = torch.stack(tf_list, dim=0)
        embeddings = find_embedding(aligned_faces_tf.to(device), emb_model)
        names = identify_person(embeddings, classify_model, label2name_df, threshold)
        bth_names, counter = [], 0
        for n_face in [len(x) for x in bth_alg_face_list]:
            bth_names.append(names[counter: counter + n_face])
            counter += n_face
    else:
        bth_names = [[] for _ in range(len(bth_alg_face_list))]
    return bth_names


class Ticket:
    """One submitted frame batch (FacePipeline.submit).  `done` is set once detection has finished on the host and
    the embedding work is enqueued; the embeddings / classes are produced on the pipeline's embedding stream."""
    __slots__ = ("counts", "boxes", "probs", "points", "emb", "amax", "prob", "event", "done", "error", "_slice", "_pipe", "upload_slot")

    def __init__(self):
        self.counts = self.boxes = self.probs = self.points = None
        self.emb = self.amax = self.prob = self.event = self.error = self._slice = self._pipe = None
        self.upload_slot = -1
        self.done = threading.Event()

    def wait_host(self):
        self.done.wait()
        if self.error is not None:
            raise self.error
        return self

    @property
    def n_faces(self):
        return int(sum(self.wait_host().counts))

    def result(self):
        """Wait for the detection, order the caller's current stream after the embedding work; returns
        (counts, boxes, emb, amax, prob)."""
        self.wait_host()
        if self.event is None and self._pipe is not None:
            self._pipe.flush()      # its faces were still waiting for a full embed batch
        if self.event is not None:
            torch.cuda.current_stream(self.emb.device).wait_event(self.event)
        return self.counts, self.boxes, self.emb, self.amax, self.prob


class FacePipeline:
    """detect -> align -> embed -> classify with every intermediate resident in HBM.

    One call per frame batch (demo_video.py:86-129 without the host round trips): frames are
    uploaded once, vnf_mtcnn_detect leaves boxes/landmarks for vnf_align, the warp writes the
    normalised NCHW batch straight in the encoder's input dtype, and only names, boxes and
    (optionally) embeddings come back."""

    def __init__(self, detector, encoder, classifier, label2name, target_size, threshold=0.0, embed_batch=0, embed_lanes=1):
        self.detectors = list(detector) if isinstance(detector, (list, tuple)) else [detector]
        self.detector, self.encoder, self.classifier = self.detectors[0], encoder, classifier
        self.label2name = label2name
        self.size = int(target_size)
        self.template = center_point_dict[str((self.size, self.size))]
        self.threshold = threshold
        self.in_dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "fp16": torch.float16}.get(
            getattr(encoder, "compute_dtype", "f32"), torch.float32)
        self._det_streams = self._emb_stream = self._threads = self._queues = None
        self._enc_lock = threading.Lock()
        self._next = 0
        # throughput mode only: faces of consecutive submits are embedded together once `embed_batch` of them are
        # waiting (0: every batch at once).  The encoder's launches have a fixed latency cost, so 256 faces cost
        # 1.4x what 128 do, not 2x.
        self.embed_batch = int(embed_batch)
        # embedding streams that consecutive embed launches rotate over (with as many encoder activation contexts),
        # so one group's latency-bound tail runs under the next group's stem
        self.embed_lanes = max(1, int(embed_lanes))
        self._lane = 0
        self._acc = None
        self._acc_n = 0
        self._pending = []
        self._classify = True

    def embed_frames(self, frames_dev):
        """frames_dev (B,H,W,3) u8 cuda -> (counts, boxes (n,4) host, embeddings (n,512) cuda)."""
        counts, boxes, probs, points = self.detector.detect_device(frames_dev)
        n = int(boxes.shape[0])
        if n == 0:
            return counts, boxes, torch.empty((0, 512), dtype=torch.float32, device=frames_dev.device)
        fidx_d, boxes_d, _, points_d = self.detector.results_device(n, frames_dev.device)   # no host round trip
        _, faces = align_faces_device(frames_dev, fidx_d, boxes_d, points_d, self.template, self.size, want_u8=False,
                                      norm_dtype=self.in_dtype)
        return counts, boxes, self.encoder(faces)

    def _detect_embed(self, k, frames_dev, ready, t, classify):
        """detection on detector k's stream, then (under the encoder lock) alignment + embedding + classification
        on the shared embedding stream"""
        dev = frames_dev.device
        det_s = self._det_streams[k]
        det_s.wait_event(ready)
        with torch.cuda.stream(det_s):
            t.counts, t.boxes, t.probs, t.points = self.detectors[k].detect_device(frames_dev)
            n = len(t.boxes)
            if n:
                # the detections stay on the device for the warp (no host round trip, no pageable H2D copy whose
                # implicit stream synchronisation would stall the host behind the previous batch's embedding)
                fidx_d, boxes_d, _, points_d = self.detectors[k].results_device(n, dev)
                found = det_s.record_event()
        frames_dev.record_stream(det_s)
        if n:
            with self._enc_lock:
                emb_s = self._emb_stream = self._emb_streams[self._lane]
                emb_s.wait_event(ready)
                emb_s.wait_event(found)
                with torch.cuda.stream(emb_s):
                    if self.embed_batch <= 0:
                        _, faces = align_faces_device(frames_dev, fidx_d, boxes_d, points_d, self.template, self.size,
                                                      want_u8=False, norm_dtype=self.in_dtype)
                        t.emb = self.encoder(faces)
                        if classify:
                            _, t.amax, t.prob = self.classifier.classify(t.emb, want_logp=False)
                        t.event = emb_s.record_event()
                        self._lane = (self._lane + 1) % self.embed_lanes
                    else:
                        cap = max(self.embed_batch, n)
                        if self._acc_n + n > cap:
                            self._flush_locked()      # moves on to the next lane
                            emb_s = self._emb_stream
                            emb_s.wait_event(ready)
                            emb_s.wait_event(found)
                        with torch.cuda.stream(emb_s):
                            acc = self._acc[self._lane]
                            if acc is None or acc.shape[0] < cap:
                                acc = self._acc[self._lane] = torch.empty((cap, 3, self.size, self.size),
                                                                             dtype=self.in_dtype, device=dev)
                            align_faces_device(frames_dev, fidx_d, boxes_d, points_d, self.template, self.size,
                                               want_u8=False, out_norm=acc[self._acc_n:self._acc_n + n])
                        t._slice, t._pipe = (self._acc_n, n), self
                        self._pending.append(t)
                        self._acc_n += n
                        self._classify = classify
                        frames_dev.record_stream(emb_s)
                        if self._acc_n >= self.embed_batch:
                            self._flush_locked()
                for x in (frames_dev, fidx_d, boxes_d, points_d):
                    x.record_stream(emb_s)
        else:
            t.emb = torch.empty((0, 512), dtype=torch.float32, device=dev)

    def _flush_locked(self):
        """embed (+ classify) the waiting faces on the current lane's stream, then move to the next lane; caller holds
        the encoder lock"""
        if not self._pending:
            return
        emb_s = self._emb_streams[self._lane]
        with torch.cuda.stream(emb_s):
            emb = self.encoder(self._acc[self._lane][:self._acc_n])
            amax = prob = None
            if self._classify:
                _, amax, prob = self.classifier.classify(emb, want_logp=False)
            ev = emb_s.record_event()
        for t in self._pending:
            o, k = t._slice
            t.emb = emb[o:o + k]
            if amax is not None:
                t.amax, t.prob = amax[o:o + k], prob[o:o + k]
            t.event = ev
        self._pending, self._acc_n = [], 0
        self._lane = (self._lane + 1) % self.embed_lanes
        self._emb_stream = self._emb_streams[self._lane]

    def flush(self):
        """Embed whatever faces are still waiting for a full embed batch (end of stream)."""
        with self._enc_lock:
            self._flush_locked()

    def _worker(self, k, dev):
        torch.cuda.set_device(dev)
        while True:
            job = self._queues[k].get()
            if job is None:
                return
            frames_dev, ready, t, classify = job
            try:
                self._detect_embed(k, frames_dev, ready, t, classify)
            except BaseException as e:   # surfaced by ticket.result()
                t.error = e
            t.done.set()

    def submit(self, frames_dev, classify=True, ready=None):
        """Throughput mode: enqueue one frame batch and return a ticket without waiting for it.  `ready`: event after
        which frames_dev is resident (upload.FrameUploader's copy stream); default: the caller's current stream.

        Detection runs on a detection stream (it synchronises with the host three times to size the candidate
        tables: detect_face.py's own stage boundaries); alignment + embedding (+ classification) run on the
        embedding stream, so the embedding of batch i overlaps the detection of batch i+1.  With several
        detector handles (`FacePipeline(detector=[d0, d1], ...)`) each gets a host thread and a stream, so
        detections also overlap each other across their host synchronisations; batches go round-robin and
        `ticket.result()` returns them in submission order regardless of completion order."""
        dev = frames_dev.device
        if self._det_streams is None:
            # process-wide stream objects, roles in a fixed order (streams.py: a fresh pair per pipeline can end up on
            # one hardware queue and serialise detection with embedding)
            from .streams import side_streams
            nd = len(self.detectors)
            self._det_streams = [side_streams(dev, 1, 0)[0]] + side_streams(dev, nd - 1, 1 + self.embed_lanes)
            self._emb_streams = side_streams(dev, self.embed_lanes, 1)
            self._emb_stream = self._emb_streams[0]
            self._acc = [None] * self.embed_lanes
            if hasattr(self.encoder, "set_streams"):
                self.encoder.set_streams(1)   # the detection stream fills the gaps the encoder's own forks would
                if self.embed_lanes > 1:
                    self.encoder.set_contexts(self.embed_lanes)
            if len(self.detectors) > 1:
                import queue
                self._queues = [queue.Queue() for _ in self.detectors]
                self._threads = [threading.Thread(target=self._worker, args=(k, dev), daemon=True)
                                 for k in range(len(self.detectors))]
                for th in self._threads:
                    th.start()
        if ready is None:
            ready = torch.cuda.current_stream(dev).record_event()
        t = Ticket()
        if len(self.detectors) == 1:
            self._detect_embed(0, frames_dev, ready, t, classify)
            t.done.set()
        else:
            self._queues[self._next % len(self.detectors)].put((frames_dev, ready, t, classify))
            self._next += 1
        return t

    def close(self):
        """Stop the detection threads (idempotent)."""
        if self._threads:
            for q in self._queues:
                q.put(None)
            for th in self._threads:
                th.join()
            self._threads = None

    @property
    def embed_stream(self):
        return self._emb_stream

    def recognize_frames(self, rgb_images):
        """list of equal-size HWC u8 RGB frames (or a (B,H,W,3) array / cuda tensor) ->
        (per-frame name lists, per-frame box lists, embeddings (n,512) cuda)."""
        frames, _ = self.detector._to_device_frames(rgb_images)
        counts, boxes, emb = self.embed_frames(frames)
        names = identify_person(emb, self.classifier, self.label2name, self.threshold) if emb.shape[0] else []
        bth_names, bth_boxes, o = [], [], 0
        for c in counts:
            bth_names.append(names[o:o + c])
            bth_boxes.append([boxes[k] for k in range(o, o + c)])
            o += c
        return bth_names, bth_boxes, emb


def get_face_from_boxes(image, boxes, box_requirements=None):
    """demo_image.py:174-199 (host views, no arithmetic)."""
    list_faces, face_idx = [], []
    ori_h, ori_w = image.shape[:2]
    for idx, box in enumerate(boxes):
        x1 = max(int(box[0]), 0)
        y1 = max(int(box[1]), 0)
        x2 = min(int(box[2] + 1), ori_w)
        y2 = min(int(box[3] + 1), ori_h)
        w, h = x2 - x1, y2 - y1
        max_dim, min_dim = max(w, h), min(w, h)
        chosen = box_requirements is None or (min_dim > box_requirements['min_dim'] and
                                              (max_dim / min_dim < box_requirements['box_ratio']))
        if chosen:
            list_faces.append(image[y1:y2, x1:x2, :])
            face_idx.append(idx)
    return list_faces, face_idx


def move_landmark_to_box(box, landmark):
    return landmark - box[:2]


def parallel_detect_and_align(rgb_images, detection_md, center_point, target_fs, log=False):
    """demo_image.py:273-306: returns (per-image lists of aligned (S,S,3) u8 faces, chosen boxes).
    Frames are uploaded once; detection and the warp both read them from HBM."""
    bth_boxes, _, bth_landmarks = detection_md.inference(rgb_images, landmark=True)
    frames = detection_md.last_frames_device()
    fidx, boxes, points = [], [], []
    for i, (bx, lm) in enumerate(zip(bth_boxes, bth_landmarks)):
        for b, l in zip(bx, lm):
            fidx.append(i); boxes.append(b); points.append(l)
    n = len(boxes)
    bth_aligned_faces = [[] for _ in rgb_images]
    bth_chosen_bb = [[] for _ in rgb_images]
    if n:
        u8, _ = align_faces_device(frames, np.asarray(fidx, np.int32), np.asarray(boxes, np.float32),
                                   np.asarray(points, np.float32), center_point, target_fs[0])
        u8 = u8.cpu().numpy()
        for k in range(n):
            bth_aligned_faces[fidx[k]].append(u8[k])
            bth_chosen_bb[fidx[k]].append(boxes[k])
    elif log:
        print('Face not found in this image !')
    return bth_aligned_faces, bth_chosen_bb
