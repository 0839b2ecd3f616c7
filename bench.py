#!/usr/bin/env python3
"""Headline benchmark: embeddings/sec @ bs=256 on InceptionResnetV1 (BASELINE.json configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of the HIP embed path (pack -> 130 MFMA convolutions + pools -> L2 norm) over
one batch of 256 synthetic 160x160 crops (N(0,1), seed 0, bf16, already resident in HBM) with
random-init generator weights (seed 0; no pretrained weights exist offline).  With N > 1 every
rank embeds its own 256 crops (weak scaling, no data-path collective) and the ranks exchange the
resulting (256,512) fp32 embeddings with one RCCL all-gather per step, overlapped with the next
step.  Rank 0 prints ONE JSON line.

The default run (no --workload) carries, inside that one line:
  * the headline leg (value / ms_per_step / roofline / cpu_baseline): bf16, BASELINE configs[1]; "in_gate": the rate of
    the f16x2 (split-f16) path that meets the 1e-4 gate, promoted from the legs;
  * "parity": measured L2 error of the timed dtype against the reference's own embeddings (tests/golden/
    irv1_seed0.npz) next to the 1e-4 north-star gate, and whether the timed 3-lane bs=256 output equals the same
    images embedded 8 at a time on one stream (bitwise) -- so the line says which path is inside the gate;
  * "legs": the same workload timed in the other compute dtypes (f16x2 = split-f16, inside the gate; f16; f32), the
    headline dtype on one lane, and BASELINE configs[4] (IR-100);
  * "pipeline": BASELINE's other half -- faces/sec end to end on synthetic 1080p frames (16 frames/step/GPU, resident in
    HBM); "pipeline_f16x2": the same with the in-gate encoder (the CLIs' default); "stream": that pipeline fed from
    pinned host memory through the asynchronous uploader, with the measured H2D ceiling.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16/f16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
BATCH = 256


def host_cores():
    """Cores this process may really use: affinity mask, cgroup CPU quota, and the GPU box's
    per-GPU CPU share (16) -- os.cpu_count() reports the whole host and oversubscribes."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 16))


def cpu_baseline(budget_s=12.0):
    """The oracle (CPU restatement of the reference's InceptionResnetV1 forward, fp32 torch-CPU
    eager) timed on this box's host cores on a bounded sample of the same workload."""
    import torch
    from oracle import irv1
    from vn_celeb_face_recognition_amd.weights import generate_state_dict
    torch.set_num_threads(host_cores())
    sd = generate_state_dict("irv1", 0, as_torch=True)
    bs = 32
    x = torch.randn((bs, 3, 160, 160), generator=torch.Generator().manual_seed(0))
    irv1.irv1_forward(sd, x[:4])  # warm up allocator / thread pool
    n = 0
    t0 = time.perf_counter()
    while True:
        irv1.irv1_forward(sd, x)
        n += bs
        dt = time.perf_counter() - t0
        if dt >= budget_s:
            break
    return {"value": round(n / dt, 2), "unit": "embeddings/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d images (batches of %d of the bs=256 workload), fp32 torch-CPU oracle, %.1f s" % (n, bs, dt)}


def cpu_baseline_pipeline(frames, template_size=160, n_frames=32, budget_s=12.0):
    """Oracle detect -> align -> embed -> classify on a bounded sample of the same frames."""
    import numpy as np
    import torch
    from oracle import align as oalign, irv1, mlp as omlp, mtcnn as om
    from vn_celeb_face_recognition_amd.weights import generate_state_dict
    torch.set_num_threads(host_cores())
    d = os.path.join(REPO, "vn_celeb_face_recognition_amd", "weights_mtcnn")
    p, r, o = (torch.load(os.path.join(d, n + ".pt"), weights_only=True) for n in ("pnet", "rnet", "onet"))
    sd, msd = generate_state_dict("irv1", 0, as_torch=True), generate_state_dict("mlp", 0, as_torch=True)
    tmpl = oalign.CENTER_POINTS["(%d, %d)" % (template_size, template_size)]
    faces = 0
    t0 = time.perf_counter()
    for f in frames[:n_frames]:
        boxes, _, points = om.mtcnn_detect([f], p, r, o, min_face_size=50, ties="table")
        crops = oalign.detect_align_faces(f, boxes[0], points[0], tmpl, template_size, template_size)
        if crops:
            x = torch.from_numpy(np.stack([oalign.transforms_default(c) for c in crops]))
            omlp.mlp_forward(msd, irv1.irv1_forward(sd, x))
        faces += len(crops)
        n_done = n_done + 1 if "n_done" in dir() else 1
        if time.perf_counter() - t0 >= budget_s:
            break
    n_frames = n_done
    dt = time.perf_counter() - t0
    return {"value": round(faces / dt, 2), "unit": "faces/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d of the synthetic 1080p frames (%d faces), oracle detect+align+embed+classify, %.1f s" % (n_frames, faces, dt)}


def make_detector(args, models, dev, nf, per):
    """--detector mtcnn (the reference's default, cfg/detection/mtcnn.json) | retina (RetinaFace mobilenet0.25 with the
    generator's synthetic weights: keep_top_k = the pasted faces per frame, so the stages after it see the same load)."""
    if args.detector == "retina":
        return models.RetinaFace("cfg_mnet", device=dev, max_batch=nf, keep_top_k=per, vis_thres=0.0, synthetic=True)   # exact-f32 plan (default)
    return models.MTCNN(keep_all=True, min_face_size=50, device=dev, max_batch=nf, max_height=1080, max_width=1920)


def run_detect(args):
    """--workload detect: SURVEY.md 8(d) config 3 -- MTCNN detect + 5-point alignment only, synthetic 1080p frames,
    16 frames per step, frames resident in HBM; roofline = HBM on the stage-1 algorithmic bytes (a lower bound for
    the whole cascade).  Two detector handles on two host threads keep the GPU busy across each other's three host
    synchronisations."""
    import threading
    import torch
    from vn_celeb_face_recognition_amd import models
    from vn_celeb_face_recognition_amd.pipeline import align_faces_device, center_point_dict
    from vn_celeb_face_recognition_amd.synth import make_frames
    from vn_celeb_face_recognition_amd.streams import side_stream
    if args.gpus != 1:
        raise SystemExit("--workload detect is a single-GPU line (frames shard like the pipeline workload)")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    NF, PER, NT = 16, 8, max(1, args.detectors)
    frames, _ = make_frames(NF * 2, PER, seed=0)
    batches = [torch.from_numpy(frames[i * NF:(i + 1) * NF]).to(dev) for i in range(2)]
    dets = [make_detector(args, models, dev, NF, PER) for _ in range(NT)]
    tmpl = center_point_dict["(160, 160)"]
    faces = [0] * NT

    def work(k, first, n):
        torch.cuda.set_device(dev)
        st = side_stream(dev, k)
        with torch.cuda.stream(st):
            for i in range(first, first + n):
                counts, boxes, probs, points = dets[k].detect_device(batches[i & 1])
                nb = len(boxes)
                if nb:
                    fidx, bx, _, pt = dets[k].results_device(nb, dev)
                    align_faces_device(batches[i & 1], fidx, bx, pt, tmpl, 160, want_u8=False, norm_dtype=torch.bfloat16)
                faces[k] += nb
        st.synchronize()

    def run(total):
        per = [total // NT + (1 if k < total % NT else 0) for k in range(NT)]
        ths = [threading.Thread(target=work, args=(k, sum(per[:k]), per[k])) for k in range(NT)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()

    run(args.warmup)
    torch.cuda.synchronize()
    faces[:] = [0] * NT
    t0 = time.perf_counter()
    run(args.steps)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    stage1_bytes = 13_316_400.0 * NF
    achieved = stage1_bytes * args.steps / wall / 1e9
    print(json.dumps({
        "metric": "frames/sec %s detect+align on 1080p frames (SURVEY 8d config 3)" % ("MTCNN" if args.detector == "mtcnn" else "RetinaFace"),
        "value": round(NF * args.steps / wall, 1),
        "unit": "frames/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(wall * 1e3 / args.steps, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s + 5-point warp, synthetic 1080p frames, %d frames/step, %d pasted "
                               "faces/frame, %d detector handle(s)/thread(s)" % (
                                   "MTCNN P/R/O cascade (min_face_size 50)" if args.detector == "mtcnn" else
                                   "RetinaFace mobilenet0.25 (synthetic weights, keep_top_k 8)", NF, PER, NT),
                   "faces_per_s": round(sum(faces) / wall, 1)},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s",
                     "frac": round(achieved / 8000.0, 5), "traffic": None,
                     "kernel": "whole cascade priced on stage-1 (pyramid + P-Net) algorithmic bytes only: a lower bound"}}), flush=True)


def pipeline_leg(args, rank, world, local, steps, warmup, lanes, want_cpu, dtype=None, from_host=False):
    """BASELINE.json configs[2]/[3] -- faces/sec end to end (detect + align + embed + classify) on synthetic 1080p
    frames, 16 frames per step per GPU, 64 distinct frames.  from_host=False: frames resident in
    HBM when the timed region starts (the contract's `value`); from_host=True: the frames start in PINNED HOST memory (a
    decoder's staging ring) and every step uploads its 16 frames on the copy stream behind the GPU work of the steps
    before it (upload.FrameUploader) -- the rate demo_video.py sees, bounded by PCIe.  Returns the result dict on rank
    0 (None elsewhere)."""
    import torch
    import torch.distributed as dist
    from vn_celeb_face_recognition_amd import models
    from vn_celeb_face_recognition_amd.pipeline import FacePipeline
    from vn_celeb_face_recognition_amd.synth import make_frames
    from vn_celeb_face_recognition_amd.upload import FrameUploader
    dev = torch.device("cuda", local)
    dtype = dtype or args.dtype
    NF, PER, NB = 16, 8, 4
    frames, truth = make_frames(NF * NB, PER, seed=rank)
    # two detector handles: each gets a host thread and a HIP stream, so one batch's detection runs under the
    # other's host synchronisations (FacePipeline.submit)
    det = [make_detector(args, models, dev, NF, PER) for _ in range(args.detectors)]
    enc = models.InceptionResnetV1(pretrained=None, compute_dtype=dtype, max_batch=max(256, args.embed_batch)).to(dev).eval()
    clf = models.MLPModel(512, 1001).to(dev).eval()
    pipe = FacePipeline(det, enc, clf, {"label": list(range(1001)), "name": ["c%d" % i for i in range(1001)]}, 160, 0.0,
                        embed_batch=args.embed_batch, embed_lanes=lanes)
    if from_host:
        host = [torch.from_numpy(frames[i * NF:(i + 1) * NF]).pin_memory() for i in range(NB)]
        up = FrameUploader(dev, depth=2 * args.detectors + 4)
        batches = None
    else:
        batches = [torch.from_numpy(frames[i * NF:(i + 1) * NF]).to(dev) for i in range(NB)]
    # fixed-size exchange of the ranks' embeddings (weak scaling: every rank embeds its own frames; the per-batch face
    # count is data dependent, so the block carries a count row like video.run_stream's exchange -- no host read inside)
    CAP = 256
    gathered = [torch.empty((world * (CAP + 1), 512), dtype=torch.float32, device=dev) for _ in range(2)] if world > 1 else None
    blocks = [torch.zeros((CAP + 1, 512), dtype=torch.float32, device=dev) for _ in range(2)] if world > 1 else None
    state = {"n": 0}

    inflight = []

    def retire(t):
        # in submission order: the collective sequence is identical on every rank
        n = t.n_faces
        if from_host:
            if t.event is None and n:
                t.result()                                  # its faces were still waiting for a full embed batch
            up.release(t.upload_slot, t.event)              # the frames' last reader (the warp) precedes this event
        if world > 1:
            emb = t.result()[2]                             # result() orders this stream after the batch's lane
            k = state["n"] & 1
            state["n"] += 1
            blocks[k][0, 0] = float(n)
            blocks[k][1:1 + min(n, CAP)] = emb[:CAP]
            dist.all_gather_into_tensor(gathered[k], blocks[k])
        return n

    def step(i):
        # throughput mode: batches are in flight together (detection streams + embedding stream); every batch is
        # retired inside the timed region and the closing torch.cuda.synchronize() waits for all device work
        if from_host:
            # the upload runs one batch AHEAD of the submit (submit blocks on the detector's read-back): every timed
            # step issues exactly one upload and consumes the one issued in the step before
            if state.get("up") is None:
                state["up"] = up.upload(host[i % NB]) + (up.last_slot,)
            fr, ready, slot = state["up"]
            state["up"] = up.upload(host[(i + 1) % NB]) + (up.last_slot,)
            t = pipe.submit(fr, classify=True, ready=ready)
            t.upload_slot = slot
            inflight.append(t)
        else:
            inflight.append(pipe.submit(batches[i % NB], classify=True))
        return retire(inflight.pop(0)) if len(inflight) > 2 * args.detectors else 0

    def drain():
        pipe.flush()
        n = 0
        while inflight:
            n += retire(inflight.pop(0))
        return n

    for i in range(warmup):
        step(i)
    drain()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    faces = 0
    t0 = time.perf_counter()
    for i in range(steps):
        faces += step(i)
    faces += drain()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    pipe.close()
    tot = torch.tensor([wall, float(faces)], dtype=torch.float64, device=dev)
    if world > 1:
        mx = tot.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX); wall = float(mx[0])
        dist.all_reduce(tot); faces = int(tot[1].item())
    h2d = None
    if from_host and rank == 0:
        # the PCIe ceiling of this leg: the same 16-frame batches, upload only
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(8):
            up.upload(host[i % NB])
        up.stream.synchronize()
        dt = time.perf_counter() - t1
        h2d = {"frames_per_s": round(8 * NF / dt, 1), "GBps": round(8 * host[0].numel() / dt / 1e9, 2)}
    stage_ms = None
    if rank == 0 and hasattr(det[0], "stage_times") and not from_host:
        stage_ms = det[0].stage_times(batches[0])
    out = None
    if rank == 0:
        out = {"metric": "faces/sec end-to-end (detect+embed+classify) on 1080p frames", "value": round(faces / wall, 1),
               "unit": "faces/s", "n_gpus": world, "steps": steps, "warmup": warmup,
               "ms_per_step": round(wall * 1e3 / steps, 4), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": dtype, "data": "synthetic",
               "config": {"workload": "BASELINE.json configs[2]+[3]: " + ("MTCNN" if args.detector == "mtcnn" else "RetinaFace (synthetic weights)") +
                                      " detect + align + IRv1 embed + MLP classify, "
                                      "synthetic 1080p frames (%d distinct), %d frames/step/GPU, %d pasted faces/frame; detection and embedding on "
                                      "separate streams, faces embedded in groups of >= %d; %s" % (
                                          NF * NB, NF, PER, args.embed_batch,
                                          "frames start in pinned host memory, one asynchronous upload per step" if from_host
                                          else "frames resident in HBM"),
                          "frames_per_s": round(world * NF * steps / wall, 1), "min_face_size": 50,
                          "detector_handles": args.detectors, "encoder_dtype": dtype}}
        if from_host:
            out["h2d_ceiling"] = h2d
        else:
            out["roofline"] = pipeline_roofline(stage_ms, NF)
        if want_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline_pipeline(frames)
    if from_host:
        up.close()
    del pipe, det, enc, clf
    torch.cuda.empty_cache()
    return out


def pipeline_roofline(stages, nf):
    """HBM roofline of the detection front end, per kernel: algorithmic bytes of one launch (SURVEY.md 8d terms from the
    level table x the frames one launch processes, computed by the library) / that kernel's device time, measured live
    with HIP events on the detector's stream (vnf_mtcnn_stage_times).  The headline entry is the pyramid kernel: it
    is the one that reads every frame byte (6 220 800 B u8 + 2 890 236 B of fp32 levels written, per 1080p frame)."""
    if not stages:
        return {"bound": "hbm", "achieved": None, "peak": 8000.0, "unit": "GB/s", "frac": None, "traffic": None}
    per = {}
    for name, v in stages.items():
        if v["bytes"] > 0 and v["ms"] > 0:
            gbps = v["bytes"] / v["ms"] / 1e6
            per[name] = {"bytes_per_launch": v["bytes"], "ms": round(v["ms"], 4), "GBps": round(gbps, 1), "frac": round(gbps / 8000.0, 4)}
    dom = per.get("pyramid", {})
    return {"bound": "hbm", "achieved": dom.get("GBps"), "peak": 8000.0, "unit": "GB/s", "frac": dom.get("frac"), "traffic": None,
            "kernel": "pyramid_rows_kernel (all pyramid levels of the %d frames in one launch): (frame u8 bytes + fp32 level bytes) x %d "
                      "frames per launch / its HIP-event time on the detector stream" % (nf, nf),
            "per_kernel": per, "stage_ms": {k: round(v["ms"], 4) for k, v in stages.items()}}


ROUND = "r03"   # profiles/<ROUND>_traffic.json is quoted in roofline.traffic
# float64 sum of |embedding| over the bs=256 seed-0 bf16 input of the headline leg; tests/test_gpu_bench_config.py
# asserts the same value on the same configuration (deterministic kernels: any change of the arithmetic shows here)
CHECKSUMS = {("irv1", "bf16"): 4542.960144015937, ("irv1", "f16"): 4543.280890313676,
             ("irv1", "f16x2"): 4543.173085557737, ("irv1", "f32"): 4543.173369363214}


class LegError(Exception):
    """a leg failed on SOME rank: raised on every rank after the ranks agreed (so nobody is left inside a collective)"""


def agree(failed, world, dev):
    """True on every rank if `failed` is true on any rank (one all_reduce; identity for a single process)."""
    if world <= 1:
        return bool(failed)
    import torch
    import torch.distributed as dist
    t = torch.tensor([1.0 if failed else 0.0], device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return bool(t.item() > 0)


def embed_leg(args, dtype, rank, world, dev, n_lanes, steps, warmup, want_cpu, with_parity=True, model_name=None):
    """Time the embed path in one compute dtype: `warmup` untimed + exactly `steps` timed steps bracketed by barrier +
    synchronize; consecutive (independent) batches rotate over `n_lanes` streams / activation contexts.  Returns the
    result dict (rank 0) or None."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from vn_celeb_face_recognition_amd import dist as vdist
    from vn_celeb_face_recognition_amd.models import InceptionResnetV1, iresnet100
    from vn_celeb_face_recognition_amd.weights import IR100_MACS_PER_IMAGE, IRV1_MACS_PER_IMAGE

    # input tensor dtype: the 16-bit storage paths take 16-bit crops (BASELINE configs[1]: "synthetic 160x160 bf16"),
    # the fp32-class paths (f32, f16x2) take fp32 crops
    tdt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32, "f16x2": torch.float32}[dtype]
    model_name = model_name or args.model
    model, err = None, None
    try:     # creation is the part that can fail on one rank only (memory, a missing kernel): agree before any collective
        if model_name == "irv1":
            model = InceptionResnetV1(pretrained=None, device=dev, compute_dtype=dtype, max_batch=BATCH).eval()
            size, macs, mname = 160, IRV1_MACS_PER_IMAGE, "InceptionResnetV1"
        else:
            model = iresnet100(pretrained=False, compute_dtype=dtype, max_batch=BATCH).to(dev).eval()
            size, macs, mname = 112, IR100_MACS_PER_IMAGE, "IResNet-100"
        model(torch.zeros((2, 3, size, size), dtype=tdt, device=dev))
        torch.cuda.synchronize()
    except Exception as e:
        err = "%s: %s" % (type(e).__name__, e)
    if agree(err is not None, world, dev):
        raise LegError(err or "another rank failed to create the %s / %s encoder" % (model_name, dtype))
    g = torch.Generator().manual_seed(rank)
    x = torch.randn((BATCH, 3, size, size), generator=g).to(dev).to(tdt)
    gathered = [torch.empty((world * BATCH, 512), dtype=torch.float32, device=dev) for _ in range(2)] if world > 1 else None

    # Throughput mode: consecutive batches are independent, so step i runs on stream i % lanes over the encoder's
    # activation-buffer contexts (vnf_encoder_set_contexts) with whole-batch launches (no internal half-batch forks):
    # the latency-bound tail of one batch overlaps the throughput-bound stem of the next.  Every step's work is
    # complete when the timed region's closing synchronize returns.
    from vn_celeb_face_recognition_amd.streams import side_streams
    lanes = side_streams(dev, max(1, n_lanes))     # process-wide stream objects (streams.py)
    if len(lanes) > 1:
        model.set_streams(1)
        model.set_contexts(len(lanes))

    def step(i, pending):
        with torch.cuda.stream(lanes[i % len(lanes)]):
            emb = model(x)
            if world > 1:
                if pending is not None:
                    pending.wait()
                pending = vdist.all_gather_fixed(gathered[i & 1], emb, async_op=True)
        return emb, pending

    pending = None
    for i in range(warmup):
        emb, pending = step(i, pending)
    if pending is not None:
        pending.wait()
        pending = None
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for s_ in lanes:
        s_.wait_event(ev0)
    for i in range(steps):
        emb, pending = step(i, pending)
    cur = torch.cuda.current_stream(dev)
    for s_ in lanes:
        cur.wait_stream(s_)
    ev1.record()
    if pending is not None:
        pending.wait()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)  # HIP events on the stream the kernels were launched on
    if world > 1:
        t = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    emb_timed = emb.clone()
    checksum = float(emb_timed.double().abs().sum().item())
    want = CHECKSUMS.get((model_name, dtype))
    bad = not (checksum == checksum) or (want is not None and rank == 0 and abs(checksum - want) > 1e-9 * want)
    if agree(bad, world, dev):      # every rank leaves together: nobody waits in the next barrier for a rank that exited
        raise LegError("embedding checksum %.12f of the timed configuration (rank %d) differs from the pinned %r, or is not finite "
                       "on some rank" % (checksum, rank, want))

    out = None
    if rank == 0:
        parity = None
        if with_parity:
            # (a) the timed configuration against the plain one: the same 256 images, 8 at a time, one stream, one context
            model.set_contexts(1)
            serial = torch.cat([model(x[i:i + 8]) for i in range(0, BATCH, 8)])
            same = bool(torch.equal(serial, emb_timed))
            parity = {"timed_output_bitwise_equals_serial_8_at_a_time": same, "checksum_abs_sum": checksum}
            # (b) this dtype against the reference's own embeddings (golden inputs: 4 seeded + 2 real crops)
            if model_name == "irv1":
                gd = np.load(os.path.join(REPO, "tests", "golden", "irv1_seed0.npz"))
                xg = torch.randn((6, 3, 160, 160), generator=torch.Generator().manual_seed(int(gd["input_seed"])))
                xg[4:6] = torch.from_numpy(gd["real_inputs"].astype(np.float32))
                yg = model(xg.to(dev).to(tdt)).cpu().numpy()
                err = float(np.linalg.norm(yg - gd["embeddings"], axis=1).max())
                parity.update({"l2_err_vs_reference_golden": err, "gate": 1e-4, "within_gate": bool(err <= 1e-4),
                               "golden": "tests/golden/irv1_seed0.npz (embeddings produced by the reference's InceptionResnetV1)"})
        ms_per_step = wall * 1e3 / steps
        value = world * BATCH * steps / wall
        flop_per_step = 2.0 * macs * BATCH          # 2.8353 (IRv1) / 24.179 (IR-100) GFLOP per image
        achieved = flop_per_step / (dev_ms / steps * 1e-3) / 1e12
        peak = PEAK_BF16_TFLOPS if dtype != "f32" else 157.3
        alg, executed = model.flops_per_image()
        traffic = None   # HBM bytes per step from rocprofv3 PMC passes (collected separately, profiles/)
        tpath = os.path.join(REPO, "profiles", "%s_traffic.json" % ROUND)
        if dtype == "bf16" and model_name == "irv1" and os.path.exists(tpath):
            with open(tpath) as f:
                traffic = json.load(f).get("hbm_bytes_per_step")
        out = {
            "metric": "embeddings/sec @ bs=256 (%s, %dx%d)" % (mname, size, size),
            "value": round(value, 1), "unit": "embeddings/s", "n_gpus": world, "steps": steps,
            "warmup": warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[%d]: %s embedding only, synthetic %dx%d %s, "
                                   "bs=256 per GPU, generator weights seed 0" % (1 if model_name == "irv1" else 4, mname, size, size, dtype),
                       "batch_per_gpu": BATCH, "global_batch": BATCH * world,
                       "parallelism": "dp%d (frames sharded, all-gather of embeddings)" % world,
                       "lanes": len(lanes)},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4), "traffic": traffic,
                         "traffic_note": "memory-side bytes per step, FETCH_SIZE (x2 gfx950 correction) + WRITE_SIZE from separate "
                                         "rocprofv3 --pmc passes (profiles/%s_traffic.json); these L2 fabric counters include "
                                         "Infinity-Cache hits, so HBM proper is at most this; algorithmic floor 87 MB" % ROUND,
                         "kernel": "every kernel of one embed step (fused inception-block kernels + implicit-GEMM convolutions; tile "
                                   "configuration per layer chosen by the create-time autotuner); device time by HIP events; "
                                   "achieved = algorithmic FLOP (2 x 1 417 662 304 MAC x 256) / device time per step; the 16-bit MFMA "
                                   "peak is the yardstick for bf16, f16 and f16x2 (f16x2 issues 3 MFMAs per algorithmic one: hi.hi + hi.lo + lo.hi)",
                         "flop_per_step_algorithmic": flop_per_step,
                         "flop_per_image_executed": executed, "flop_per_image_counted_by_engine": alg,
                         "device_ms_per_step": round(dev_ms / steps, 4)},
        }
        if parity is not None:
            out["parity"] = parity
        if want_cpu and world == 1 and model_name == "irv1":
            out["cpu_baseline"] = cpu_baseline()
    del model
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="all", choices=["all", "embed", "pipeline", "stream", "detect"],
                    help="all (default): the embed headline leg + the other legs + the pipeline legs in ONE JSON line; "
                         "embed / pipeline / stream (pipeline from pinned host memory) / detect: that leg alone")
    ap.add_argument("--embed-batch", type=int, default=256,
                    help="pipeline workload: faces of consecutive frame batches are embedded together once this many wait "
                         "(0: every frame batch on its own)")
    ap.add_argument("--detector", default="mtcnn", choices=["mtcnn", "retina"], help="detect / pipeline workloads: the detector plugin")
    ap.add_argument("--detectors", type=int, default=0,
                    help="detector handles (host threads + streams) per GPU; default: 2 for the pipeline legs (the faster count since the "
                         "round-3 cascade), 1 for the stream leg (its upload look-ahead is built around one submit thread) and detect")
    ap.add_argument("--lanes", type=int, default=0,
                    help="streams / encoder activation contexts that consecutive embed launches rotate over (1: one stream; the "
                         "embed workload's encoder then splits each batch over two internal streams instead); default 3 for the "
                         "embed workload, 1 for the pipeline (measured best)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)   # 20 warm-up + 100 timed iterations
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f16x2", "f32"])
    ap.add_argument("--legs", default="f16x2,f16,f32",
                    help="workload all: further compute dtypes timed after the headline leg (comma list, '' for none)")
    ap.add_argument("--model", default="irv1", choices=["irv1", "ir100"],
                    help="irv1 = BASELINE configs[1] (default); ir100 = configs[4], the ArcFace IR-100 swap-in")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    if args.workload == "detect":
        return run_detect(args)

    import torch
    import torch.distributed as dist
    from vn_celeb_face_recognition_amd import dist as vdist

    rank, world, local = vdist.init_from_env("nccl")
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run" % (args.gpus, world))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    want_cpu = not args.no_cpu_baseline
    emb_lanes = args.lanes if args.lanes > 0 else 3
    pipe_lanes = args.lanes if args.lanes > 0 else 1

    out = None
    p = None
    pipe_first = os.environ.get("BENCH_PIPE_FIRST", "0") != "0"

    def run_pipeline(dtype=None, from_host=False, cpu=True):
        st = args.steps   # every pipeline leg times the full K steps: with 50 the fill / drain of the two-handle pipeline showed (87 k against 95 k)
        import copy
        a = copy.copy(args)
        # two handles (host threads) only in the single-process run: the multi-rank line keeps the one-handle pipeline
        # that the collective exchange was written and tested around
        a.detectors = args.detectors if args.detectors > 0 else (1 if (from_host or world > 1) else 2)
        try:
            return pipeline_leg(a, rank, world, local, st, args.warmup, pipe_lanes, want_cpu and cpu, dtype=dtype, from_host=from_host)
        except Exception as e:
            if args.workload in ("pipeline", "stream"):
                raise
            return {"error": "%s: %s" % (type(e).__name__, e)}

    # legs of one process share the package's process-wide side streams (streams.py): with fresh streams per leg the
    # second leg's streams could share a hardware queue and lose 15-20 %
    if args.workload in ("all", "pipeline") and pipe_first:
        p = run_pipeline()
    if args.workload in ("all", "embed"):
        out = embed_leg(args, args.dtype, rank, world, dev, emb_lanes, args.steps, args.warmup, want_cpu)
    # the pipeline leg runs right behind the headline leg: after the minute of side legs below the chip holds a lower
    # clock (MI355X_MICROARCH.md, DVFS) and the same leg reads 8-9 % lower (87 k against 95 k faces/s)
    if args.workload == "all" and not pipe_first:
        p = run_pipeline()
    if args.workload == "all" and args.model == "irv1":
        # the other compute dtypes: shorter legs (the f32 leg runs ~7 ms steps), no CPU baseline; then the headline dtype
        # on ONE lane (no overlap of independent batches) and BASELINE configs[4] (IR-100 swap-in encoder)
        legs = {}
        keep = ("value", "unit", "steps", "warmup", "ms_per_step", "dtype", "config", "roofline", "parity")
        plan = [(dt, dt, emb_lanes, "irv1") for dt in args.legs.split(",") if dt and dt != args.dtype]
        plan += [("%s_lanes1" % args.dtype, args.dtype, 1, "irv1"), ("ir100_%s" % args.dtype, args.dtype, emb_lanes, "ir100")]
        for name, dt, ln, mdl in plan:
            st = max(5, args.steps // (5 if (dt == "f32" or mdl == "ir100") else 2))
            try:
                r = embed_leg(args, dt, rank, world, dev, ln, st, max(3, args.warmup // 2), False, with_parity=(mdl == "irv1" and ln > 1),
                              model_name=mdl)
            except Exception as e:    # a failing side leg must not take the headline line with it (the ranks agreed: LegError)
                r = {"error": "%s: %s" % (type(e).__name__, e)}
            if rank == 0:
                legs[name] = r if "error" in r else {k: r[k] for k in keep if k in r}
        if rank == 0:
            out["legs"] = legs
            # the headline dtype is BASELINE's (bf16, outside the 1e-4 gate); the in-gate path's rate next to it
            ig = legs.get("f16x2")
            if ig and "error" not in ig:
                out["in_gate"] = {"dtype": "f16x2", "value": ig["value"], "unit": ig["unit"],
                                  "within_gate": bool(ig.get("parity", {}).get("within_gate")), "frac": ig["roofline"]["frac"]}
    if args.workload == "pipeline" and not pipe_first:
        p = run_pipeline()
    if args.workload == "stream":
        p = run_pipeline(dtype=args.dtype, from_host=True)
    if rank == 0 and args.workload in ("all", "pipeline", "stream"):
        if args.workload in ("pipeline", "stream"):
            out = p
        else:
            out["pipeline"] = p
    if args.workload == "all":
        # the same pipeline with the in-gate encoder (what the CLIs run by default), then from pinned host memory
        q = run_pipeline(dtype="f16x2", cpu=False)
        r = run_pipeline(dtype="f16x2", from_host=True, cpu=False)
        # ... and the headline dtype from pinned host memory: its compute no longer hides the PCIe transfer (the H2D ceiling
        # of 16 x 1080p frames per step is ~9.1 k frames/s = 73 k faces/s)
        rb = run_pipeline(dtype=args.dtype, from_host=True, cpu=False)
        if rank == 0:
            out["pipeline_f16x2"] = q
            out["stream"] = r
            out["stream_%s" % args.dtype] = rb
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
