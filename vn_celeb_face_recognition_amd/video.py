"""Frame-stream control flow of demo_video.py / celeb_statistic.py (/root/reference/demo_video.py:46-199), the same
on one GPU and on N:

  * frames are cut into batches of --n_frames; batch b belongs to rank b % world (frames are independent units,
    demo_video.py:186-188).  A rank only READS its own batches (random-access sources skip the others' decode);
  * every rank uploads its batches through a pinned staging ring on a copy stream (upload.FrameUploader) and pushes
    them through `FacePipeline.submit` (detection stream + embedding stream, faces of consecutive batches embedded
    together); batches are retired two late, so the GPU always has work queued;
  * retiring round r (batches r*world .. r*world + world-1) is the ONE exchange step of the path: ONE
    fixed-size all-gather of each rank's (1 + cap, 512 + 4 + 1) fp32 block -- a header row with the face count, then
    embedding, box, frame slot per face -- issued on a side stream behind the batch's own event.  Its result is read
    on the host one round LATER (pinned D2H + event), so no rank ever blocks on a collective it has just issued; a
    batch with more than `cap` faces is completed by one exactly-sized follow-up gather that every rank derives from
    the same header rows;
  * every rank issues the same collectives in the same order: rounds 0 .. ceil(batches / world) - 1, each exactly once
    (a rank without a batch in the last round sends an empty block), and nothing else in between;
  * rank 0 classifies the gathered embeddings (one vnf_classify per round) and collates tracker rows in frame order.
"""
import numpy as np
import torch
import torch.distributed as dist

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
        self.total = len(frames) if self.random_access else None   # frames in the stream (iterators: known once exhausted)

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
        self.total = count
        if q:
            yield b, q, inf

    def __iter__(self):
        """every frame in order (single-process callers that sample frames themselves: celeb_statistic.py)"""
        if self.random_access:
            return (self._get(i) for i in range(len(self.frames)))
        return iter(self.frames)


def run_stream(source, pipe, n_frames, rank=0, world=1, device=None, on_frame=None, log=None, lag=2, cap=None):
    """Push this rank's batches through `pipe.submit`, exchange per round, collate on rank 0.

    pipe: FacePipeline-like -- .submit(frames_dev, classify=False[, ready=event]) -> ticket with .result() ->
    (counts, boxes (n,4) host, emb (n,512) device, _, _), .flush(), .detector._to_device_frames(list) (CPU stand-ins
    only; on a GPU the frames go through upload.FrameUploader), .classifier.classify(emb, want_logp=False) ->
    (_, amax, prob), .classifier.num_classes, .label2name, .threshold.
    on_frame(frame_rgb, frame_number, names, boxes): called on the rank that owns the frame (annotated-frame writer);
    requesting it makes every rank classify the gathered embeddings (the names are needed where the pixels are).
    cap: faces per rank and round carried by the fixed-size exchange (default max(256, 16 * n_frames)).
    Returns (rows: {frame_number: csv row}, complete on rank 0; frames processed by this rank)."""
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    on_gpu = dev.type == "cuda"
    comm = uploader = None
    if on_gpu:
        from .streams import side_stream
        from .upload import FrameUploader
        comm = side_stream(dev, 7)          # the collective's stream (roles 0..6 belong to the pipeline, streams.py)
        uploader = FrameUploader(dev, depth=lag + 3)
    cap = int(cap) if cap else max(256, 16 * int(n_frames))
    WIDTH = 517                             # 512 embedding + 4 box + 1 frame slot
    classify_here = rank == 0 or on_frame is not None
    rows, inflight, pending = {}, [], []    # inflight: (round, ticket or None, frames, info); pending: issued exchanges
    state = {"processed": 0, "shape": None}

    def to_host(x):
        if x is None or not on_gpu:
            return x
        h = torch.empty(x.shape, dtype=x.dtype, pin_memory=True)
        h.copy_(x, non_blocking=True)
        return h

    def classify(rows_dev):
        if not classify_here or rows_dev.shape[0] == 0:
            return None, None
        _, amax, prob = pipe.classifier.classify(rows_dev[:, :512].contiguous(), want_logp=False)
        return amax, prob

    def issue(item):
        """enqueue round `rnd`'s exchange (+ classification of the gathered embeddings) on the side stream, behind
        this batch's own event only; nothing here waits on the host"""
        rnd, t, q, inf = item
        n, payload = 0, None
        if t is not None:
            counts, boxes, emb, _, _ = t.result()
            if uploader is not None:
                # the batch's frames were last read by its warp, which precedes its embedding event (a batch without
                # faces: by its detection, whose read-back the host has already waited for)
                uploader.release(getattr(t, "upload_slot", -1), getattr(t, "event", None))
            n = int(sum(counts))
            if n:
                slot = np.repeat(np.arange(len(counts)), counts).astype(np.float32)   # frame slot inside the batch
                extra = np.concatenate([np.asarray(boxes, np.float32).reshape(n, 4), slot[:, None]], axis=1)
                payload = torch.cat([emb.to(dev).float(), torch.from_numpy(extra).to(dev, non_blocking=True)], dim=1)
        rec = {"rnd": rnd, "q": q, "inf": inf, "own": t is not None, "n": n, "spill": None, "hdr": None}
        if comm is not None:
            comm.wait_stream(torch.cuda.current_stream(dev))
        with (torch.cuda.stream(comm) if comm is not None else _null()):
            if world == 1:
                gathered = payload if n else torch.empty((0, WIDTH), dtype=torch.float32, device=dev)
            else:
                blk = torch.zeros((cap + 1, WIDTH), dtype=torch.float32, device=dev)
                blk[0, 0] = float(n)
                if n:
                    blk[1:1 + min(n, cap)] = payload[:cap]
                    if n > cap:
                        rec["spill"] = payload[cap:]
                out = torch.empty((world * (cap + 1), WIDTH), dtype=torch.float32, device=dev)
                dist.all_gather_into_tensor(out, blk)
                g = out.view(world, cap + 1, WIDTH)
                rec["hdr"] = to_host(g[:, 0, 0].contiguous())
                gathered = g[:, 1:, :].reshape(world * cap, WIDTH)
            amax, prob = classify(gathered)
            if classify_here:
                rec["extra"], rec["amax"], rec["prob"] = to_host(gathered[:, 512:].contiguous()), to_host(amax), to_host(prob)
            if payload is not None and comm is not None:
                payload.record_stream(comm)
            rec["event"] = comm.record_event() if comm is not None else None
        return rec

    def consume(rec):
        """host side of a round, one round after its exchange was issued: per-rank counts from the header rows, the
        follow-up gather for a batch that did not fit the fixed block, names and tracker rows"""
        if rec["event"] is not None:
            rec["event"].synchronize()
        counts = [rec["n"]] if world == 1 else [int(round(float(c))) for c in rec["hdr"].tolist()]
        parts = None
        if classify_here:
            ex = rec["extra"].numpy()
            am = rec["amax"].numpy() if rec["amax"] is not None else np.zeros((0,), np.int32)
            pr = rec["prob"].numpy() if rec["prob"] is not None else np.zeros((0,), np.float32)
            stride = cap if world > 1 else 0
            parts = [[ex[r * stride: r * stride + min(c, cap if world > 1 else c)],
                      am[r * stride: r * stride + min(c, cap if world > 1 else c)],
                      pr[r * stride: r * stride + min(c, cap if world > 1 else c)]] for r, c in enumerate(counts)]
        if world > 1 and max(counts) > cap:
            # rare: some rank's batch held more faces than the fixed block; every rank sees the same headers, so every
            # rank issues this same exactly-sized follow-up here
            m = max(counts) - cap
            with (torch.cuda.stream(comm) if comm is not None else _null()):
                blk = torch.zeros((m, WIDTH), dtype=torch.float32, device=dev)
                if rec["spill"] is not None:
                    blk[:rec["spill"].shape[0]] = rec["spill"]
                out = torch.empty((world * m, WIDTH), dtype=torch.float32, device=dev)
                dist.all_gather_into_tensor(out, blk)
                amax, prob = classify(out)
                if classify_here:
                    ex2, am2, pr2 = out[:, 512:].cpu().numpy(), amax.cpu().numpy(), prob.cpu().numpy()
                    for r, c in enumerate(counts):
                        k = max(0, c - cap)
                        parts[r] = [np.concatenate([parts[r][0], ex2[r * m: r * m + k]]),
                                    np.concatenate([parts[r][1], am2[r * m: r * m + k]]),
                                    np.concatenate([parts[r][2], pr2[r * m: r * m + k]])]
                elif comm is not None:
                    comm.synchronize()
        if not classify_here:
            return
        rnd, q, inf = rec["rnd"], rec["q"], rec["inf"]
        amax_all = np.concatenate([p[1] for p in parts]) if parts else np.zeros((0,), np.int32)
        names = identify_names(amax_all, np.concatenate([p[2] for p in parts]), pipe.classifier.num_classes,
                               pipe.label2name, pipe.threshold) if amax_all.shape[0] else []
        o = 0
        for r, (ex, _, _) in enumerate(parts):                # rank r ran batch rnd * world + r
            k = ex.shape[0]
            nm, bx, sl = names[o:o + k], ex[:, 0:4], ex[:, 4].astype(np.int64)
            o += k
            if r == rank and rec["own"]:                      # my own frames: pixels, times and numbers are here
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

    def retire(item):
        pending.append(issue(item))
        while len(pending) > 1:
            consume(pending.pop(0))

    def uploads():
        """this rank's batches with the upload one batch AHEAD: submit() blocks on the detector's single read-back, so
        the next batch's host -> HBM copy has to be in flight before it, or copy and detection would take turns"""
        it = source.rank_batches(n_frames, rank, world)
        nxt = next(it, None)
        up = (uploader.upload(nxt[1]) + (uploader.last_slot,)) if (nxt is not None and uploader is not None) else None
        while nxt is not None:
            cur, cur_up = nxt, up
            nxt = next(it, None)
            up = (uploader.upload(nxt[1]) + (uploader.last_slot,)) if (nxt is not None and uploader is not None) else None
            yield cur, cur_up

    rounds = 0
    for (b, q, inf), up in uploads():
        if state["shape"] is None:
            state["shape"] = q[0].shape                       # frames of one stream share a shape
        if up is not None:
            frames_dev, ready, slot = up
            ticket = pipe.submit(frames_dev, classify=False, ready=ready)
            ticket.upload_slot = slot
        else:
            frames_dev, _ = pipe.detector._to_device_frames(q)
            ticket = pipe.submit(frames_dev, classify=False)
        inflight.append((b // world, ticket, q, inf))
        rounds = b // world + 1
        state["processed"] += len(q)
        if log is not None:
            log(state["processed"], inf)
        while len(inflight) > lag:
            retire(inflight.pop(0))
    # Every rank joins every round's exchange, in round order, and issues no other collective: the stream's length is
    # known to every rank by now (random access: len(); a decoder: every rank has pulled every frame), so the number of
    # rounds follows locally.  A rank owns rounds 0 .. rounds-1 without gaps; only the last round can be missing.
    total_frames = int(source.total if source.total is not None else state["processed"])
    total_batches = (total_frames + n_frames - 1) // n_frames
    total_rounds = (total_batches + world - 1) // world
    for rnd in range(rounds, total_rounds):
        inflight.append((rnd, None, None, None))
    if hasattr(pipe, "flush"):
        pipe.flush()
    while inflight:
        retire(inflight.pop(0))
    while pending:
        consume(pending.pop(0))
    if uploader is not None:
        uploader.close()
    if rank == 0:
        # a frame without faces sent nothing through the exchange: its (empty) row follows from the frame count
        for num in range(1, total_frames + 1):
            if num not in rows:
                rows[num] = tracker_row(num / source.fps, num, [], [], state["shape"] or (1, 1, 3))
    return rows, state["processed"]


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
