#!/usr/bin/env python3
"""Face recognition on a frame stream: drop-in for /root/reference/demo_video.py (main 46-199, CLI
202-287): same flags, tracker CSV (header Time,Names,Frame_idx,Bboxes; rows 155-168) and console lines.

Each queue of --n_frames frames goes through the resident pipeline (detect -> align -> embed ->
classify in HBM).  Launched under torch.distributed.run, frame batch b is handled by rank
b % world_size (frames are independent, demo_video.py:186-188), every rank all-gathers the
per-batch embeddings over RCCL, and rank 0 writes the tracker rows in frame order.
Annotated frames are written only with -sfr (the reference's test at l.149 is always true and
PNG-encodes every frame).  Input: a directory of frames, a .npy array of
(T,H,W,3) RGB frames, or a video file when OpenCV is installed."""
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from demo_image import build_models, build_parser
from vn_celeb_face_recognition_amd import dist as vdist
from vn_celeb_face_recognition_amd.cli_utils import (append_log_to_file, convert_sec_to_max_time_quantity,
                                                     draw_boxes_on_image, open_frame_source, write_rgb)
from vn_celeb_face_recognition_amd.pipeline import FacePipeline, identify_names


def tracker_row(time_in_video, frame_idx, names, bboxes, frame_shape):
    """demo_video.py:155-168 (one CSV row)."""
    row = [str(time_in_video), '"' + str(names) + '"', str(frame_idx)]
    if len(bboxes) == 0:
        scaled_bboxes = []
    else:
        h, w, _ = frame_shape
        scale = np.array([w, h, w, h])
        scaled_bboxes = [list(x / scale) for x in bboxes]
    row.append('"' + str(scaled_bboxes) + '"')
    return ','.join(row) + '\n'


def main(args, pipe, rank, world):
    if rank == 0:
        os.makedirs(args.output_frame, exist_ok=True)
        with open(args.output_tracker, 'w') as f:
            f.write('')
        append_log_to_file(args.output_tracker, ['Time', 'Names', 'Frame_idx', 'Bboxes'])
    print('Method: {}'.format(args.inference_method))
    frames_iter, fps = open_frame_source(args.video_path)
    count = processed_frame = batch_id = 0
    start_time = time.time()
    queue, info = [], []
    rows = {}

    pending = []  # this rank's batch of the current round (at most one)

    def take(q, inf):
        """Queue a full batch: round r gives batch r*world + k to rank k."""
        nonlocal batch_id
        if q and (batch_id % world) == rank:
            pending.append((q, inf))
        if q:
            batch_id += 1
            if batch_id % world == 0:
                end_round()

    inflight = []  # single rank: tickets of the throughput pipeline, retired in order

    def retire(item):
        t, q, inf = item
        counts, boxes, emb, amax, prob = t.result()
        names = identify_names(amax, prob, pipe.classifier.num_classes, pipe.label2name, pipe.threshold) if len(boxes) else []
        o = 0
        for idx, c in enumerate(counts):
            nm, bx = names[o:o + c], [boxes[k] for k in range(o, o + c)]
            o += c
            if args.save_frame_recognized:
                img = draw_boxes_on_image(q[idx], bx, nm) if nm else q[idx]
                write_rgb(os.path.join(args.output_frame, 'frame_{}.png'.format(inf[idx][1])), img)
            rows[inf[idx][1]] = tracker_row(inf[idx][0], inf[idx][1], nm, bx, q[idx].shape)

    def end_round():
        """Every rank runs its batch (or none), then ALL ranks meet in the embedding all-gather.  A single rank has
        nobody to meet: its batches go through FacePipeline.submit (detection and embedding streams overlap, faces of
        consecutive batches embedded together) and are retired two batches later."""
        nonlocal processed_frame
        emb = torch.empty((0, 512), dtype=torch.float32, device='cuda')
        if pending:
            q, inf = pending.pop()
            processed_frame += len(q)
            if (processed_frame % args.log_step) == 0:
                print('Processing for frame: {}, time: {}'.format(inf[-1][1], convert_sec_to_max_time_quantity(inf[-1][0])))
            if world == 1:
                frames_dev, _ = pipe.detector._to_device_frames(q)
                inflight.append((pipe.submit(frames_dev), q, inf))
                while len(inflight) > 2:
                    retire(inflight.pop(0))
                return
            bth_names, bth_boxes, emb = pipe.recognize_frames(q)
            for idx, names in enumerate(bth_names):
                if args.save_frame_recognized:
                    img = draw_boxes_on_image(q[idx], bth_boxes[idx], names) if names else q[idx]
                    write_rgb(os.path.join(args.output_frame, 'frame_{}.png'.format(inf[idx][1])), img)
                rows[inf[idx][1]] = tracker_row(inf[idx][0], inf[idx][1], names, bth_boxes[idx], q[idx].shape)
        if world > 1:
            vdist.all_gather_embeddings(emb)      # the one exchange step (north_star)

    for frame in frames_iter:
        count += 1
        queue.append(frame)
        info.append([count / fps, count])
        if len(queue) == args.n_frames:
            take(queue, info)
            queue, info = [], []
    take(queue, info)
    if batch_id % world != 0:
        end_round()
    pipe.flush()
    while inflight:
        retire(inflight.pop(0))
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, rows)
        rows = {k: v for d in gathered for k, v in d.items()}
        tot = torch.tensor([processed_frame], device='cuda')
        dist.all_reduce(tot)
        processed_frame = int(tot.item())
    if rank == 0:
        with open(args.output_tracker, 'a') as f:
            f.write(''.join(rows[k] for k in sorted(rows)))
        processed_time = time.time() - start_time
        print('Saved tracker file in {} ...'.format(args.output_tracker))
        print('FPS for recognition face: {}'.format(int(processed_frame / processed_time)))


if __name__ == '__main__':
    p = build_parser('Face recognition on a video')
    p.add_argument('-i', '--video_path', default='video.mp4', type=str)
    p.add_argument('-o', '--output_frame', default='output_frame', type=str)
    p.add_argument('-ot', '--output_tracker', default='tracker.csv', type=str)
    p.add_argument('-ov', '--output_video', default='', type=str)
    p.add_argument('-fps', '--fps_video', default=25.0, type=float)
    p.add_argument('-sfr', '--save_frame_recognized', action='store_true')
    p.add_argument('--log_step', default=100, type=int)
    p.add_argument('--n_frames', default=16, type=int)
    args = p.parse_args()
    if args.inference_method != 'par_fd_vs_aln':
        raise SystemExit("use --inference_method par_fd_vs_aln (seq_fd_vs_aln needs the FAN landmark network, outside "
                         "the hot path and broken in the reference for list input)")
    if args.output_video:
        raise SystemExit("-ov needs OpenCV's VideoWriter, which is not installed; keep the frames with -sfr")
    rank, world, local = vdist.init_from_env()
    device = 'cuda:%d' % local
    torch.cuda.set_device(local)
    label2name_df, detection_md, emb_model, classify_model = build_models(args, device)
    pipe = FacePipeline(detection_md, emb_model, classify_model, label2name_df, args.target_face_size, args.recog_threshold,
                        embed_batch=256 if world == 1 else 0)
    main(args, pipe, rank, world)
    if world > 1:
        dist.destroy_process_group()
