"""Frame-stream control flow of demo_video.py / celeb_statistic.py (/root/reference/demo_video.py:46-199), the same
on one GPU and on N:

  * frames are cut into batches of --n_frames; batch b belongs to rank b % world (frames are independent units,
    demo_video.py:186-188).  A rank only READS its own batches (random-access sources skip the others' decode);
  * every rank pushes its batches through `FacePipeline.submit` (detection stream + embedding stream, faces of
    consecutive batches embedded together) and retires them two batches late, so the GPU always has work queued;
  * retiring round r (batches r*world .. r*world + world-1) is the ONE exchange step of the path: an
    all-gather(v) of each rank's (faces, 512 + 4 + 1) fp32 rows -- embedding, box, frame slot -- issued on a side
    stream behind the batch's own event, i.e. overlapped with the batches already in flight;
  * rank 0 classifies the gathered embeddings (one vnf_classify per round) and collates tracker rows in frame order.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import dist as vdist
from .pipeline import identify_names


def tracker_row(time_in_video, frame_idx, names, bboxes, frame_shape):
    """demo_video.py:155-168 (one CSV row)."""
    row = [str(time_in_video), '"' + str(names) + '"', str(frame_idx)]
    if len(bboxes) == 0:
        scaled_bboxes = []
    else:
        h, w, _ = frame_shape
        scale = np.array([w, h, w, h])
        # plain python floats: NumPy >= 2 would print np.float64(...) into the CSV, the reference's files hold bare numbers
        scaled_bboxes = [[float(v) for v in x / scale] for x in bboxes]
    row.append('"' + str(scaled_bboxes) + '"')
    return ','.join(row) + '\n'


class FrameSource:
    """Batches of a frame stream for one rank.  `frames` is either a random-access sequence (len + __getitem__: a list
    of image paths is decoded with `load`, an array is indexed) or a plain iterator (a decoder: the other ranks' frames
    are pulled and dropped).  Frame numbers start at 1 and time = number / fps, as demo_video.py:84-90 counts them."""

    def __init__(self, frames, fps, load=None):
        self.frames, self.fps, self.load = frames, float(fps), load
        self.random_access = hasattr(frames, "__getitem__") and hasattr(frames, "__len__")
        self.reads = 0   # frames this rank actually fetched (tests)

    def _get(self, i):
        self.reads += 1
        f = self.frames[i]
        return self.load(f) if self.load is not None else np.asarray(f)

    def rank_batches(self, n_frames, rank=0, world=1):
        """yields (batch_index, [frames], [[time, frame_number], ...]) for batches with batch_index % world == rank"""
        if self.random_access:
            total = len(self.frames)
            for b in range(rank, (total + n_frames - 1) // n_frames, world):
                idx = range(b * n_frames, min(total, (b + 1) * n_frames))
                yield b, [self._get(i) for i in idx], [[(i + 1) / self.fps, i + 1] for i in idx]
            return
        b, q, inf, count = 0, [], [], 0
        for frame in self.frames:
            count += 1
            if b % world == rank:
                self.reads += 1
                q.append(frame)
                inf.append([count / self.fps, count])
            if count % n_frames == 0:
                if q:
                    yield b, q, inf
                b, q, inf = b + 1, [], []
        if q:
            yield b, q, inf

    def __iter__(self):
        """every frame in order (single-process callers that sample frames themselves: celeb_statistic.py)"""
        if self.random_access:
            return (self._get(i) for i in range(len(self.frames)))
        return iter(self.frames)


def run_stream(source, pipe, n_frames, rank=0, world=1, device=None, on_frame=None, log=None, lag=2):
    """Push this rank's batches through `pipe.submit`, exchange per round, collate on rank 0.

    pipe: FacePipeline-like -- .detector._to_device_frames(list) -> (frames_dev, _), .submit(frames_dev, classify=False)
    -> ticket with .result() -> (counts, boxes (n,4) host, emb (n,512) device, _, _), .flush(),
    .classifier.classify(emb, want_logp=False) -> (_, amax, prob), .classifier.num_classes, .label2name, .threshold.
    on_frame(frame_rgb, frame_number, names, boxes): called on the rank that owns the frame (annotated-frame writer);
    requesting it makes every rank classify the gathered embeddings (the names are needed where the pixels are).
    Returns (rows: {frame_number: csv row}, complete on rank 0; frames processed by this rank)."""
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    if dev.type == "cuda":
        from .streams import side_stream
        comm = side_stream(dev, 7)          # the collective's stream (roles 0..6 belong to the pipeline, streams.py)
    else:
        comm = None
    classify_here = rank == 0 or on_frame is not None
    rows, inflight = {}, []        # inflight: (round, ticket or None, frames, info)
    state = {"processed": 0, "shape": None}

    def exchange(payload):
        """all-gather(v) of the round's rows + classification of the gathered embeddings; on the side stream when on a
        GPU, so it only waits for this batch's event, not for the batches submitted after it"""
        parts, _ = vdist.all_gather_embeddings(payload)
        amax = prob = None
        if classify_here:
            allp = torch.cat(parts) if len(parts) > 1 else parts[0]
            if allp.shape[0]:
                _, amax, prob = pipe.classifier.classify(allp[:, :512].contiguous(), want_logp=False)
                amax, prob = amax.cpu(), prob.cpu()
        return [p.cpu() for p in parts], amax, prob

    def retire(item):
        rnd, t, q, inf = item
        payload = torch.empty((0, 517), dtype=torch.float32, device=dev)   # a round without a batch of mine
        if t is not None:
            counts, boxes, emb, _, _ = t.result()
            n = int(sum(counts))
            if n:
                slot = np.repeat(np.arange(len(counts)), counts).astype(np.float32)   # frame slot inside the batch
                extra = np.concatenate([np.asarray(boxes, np.float32).reshape(n, 4), slot[:, None]], axis=1)
                payload = torch.cat([emb.to(dev).float(), torch.from_numpy(extra).to(dev)], dim=1)
        if comm is not None:
            comm.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(comm):
                parts, amax, prob = exchange(payload)
        else:
            parts, amax, prob = exchange(payload)
        if not classify_here:
            return
        names = identify_names(amax, prob, pipe.classifier.num_classes, pipe.label2name, pipe.threshold) if amax is not None else []
        o = 0
        for r, part in enumerate(parts):                      # rank r ran batch rnd * world + r
            k = part.shape[0]
            nm, bx, sl = names[o:o + k], part[:, 512:516].numpy(), part[:, 516].numpy().astype(np.int64)
            o += k
            if r == rank and t is not None:                   # my own frames: pixels, times and numbers are here
                for i, (tm, num) in enumerate(inf):
                    sel = np.nonzero(sl == i)[0]
                    f_names, f_boxes = [nm[j] for j in sel], [bx[j] for j in sel]
                    if on_frame is not None:
                        on_frame(q[i], num, f_names, f_boxes)
                    if rank == 0:
                        rows[num] = tracker_row(tm, num, f_names, f_boxes, q[i].shape)
            elif rank == 0:                                   # another rank's frames: number and time follow from the batch index
                for i in np.unique(sl):
                    sel = np.nonzero(sl == i)[0]
                    num = (rnd * world + r) * n_frames + int(i) + 1
                    rows[num] = tracker_row(num / source.fps, num, [nm[j] for j in sel], [bx[j] for j in sel], state["shape"])

    rounds = 0
    for b, q, inf in source.rank_batches(n_frames, rank, world):
        if state["shape"] is None:
            state["shape"] = q[0].shape                       # frames of one stream share a shape
        frames_dev, _ = pipe.detector._to_device_frames(q)
        inflight.append((b // world, pipe.submit(frames_dev, classify=False), q, inf))
        rounds = b // world + 1
        state["processed"] += len(q)
        if log is not None:
            log(state["processed"], inf)
        while len(inflight) > lag:
            retire(inflight.pop(0))
    # every rank joins every round's exchange: the last round may hold batches for the low ranks only
    total_rounds, total_frames = rounds, state["processed"]
    if world > 1:
        t = torch.tensor([rounds, -state["processed"]], dtype=torch.int64, device=dev)
        mx = t.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(t)
        total_rounds, total_frames = int(mx[0].item()), -int(t[1].item())
    if hasattr(pipe, "flush"):
        pipe.flush()
    while inflight:
        retire(inflight.pop(0))
    for rnd in range(rounds, total_rounds):
        retire((rnd, None, None, None))
    if rank == 0:
        # a frame without faces sent nothing through the exchange: its (empty) row follows from the frame count
        for num in range(1, total_frames + 1):
            if num not in rows:
                rows[num] = tracker_row(num / source.fps, num, [], [], state["shape"] or (1, 1, 3))
    return rows, state["processed"]
