#!/usr/bin/env python3
"""Face recognition on a frame stream: drop-in for /root/reference/demo_video.py (main 46-199, CLI
202-287): same flags, tracker CSV (header Time,Names,Frame_idx,Bboxes; rows 155-168) and console lines.

Each queue of --n_frames frames goes through the resident throughput pipeline (FacePipeline.submit: detect ->
align -> embed in HBM, detection and embedding streams overlapped).  Launched under torch.distributed.run,
frame batch b is handled -- and read -- by rank b % world_size only (frames are independent,
demo_video.py:186-188); per round the ranks all-gather their embeddings + boxes over RCCL on a side stream
and rank 0 classifies the gathered tensor and writes the tracker rows in frame order (video.run_stream).
Annotated frames are written only with -sfr (the reference's test at l.149 is always true and
PNG-encodes every frame).  Input: a directory of frames, a .npy array of
(T,H,W,3) RGB frames, a Motion-JPEG .avi, or any video file when OpenCV is installed; -ov exports the annotated frames
as a video (MP4V with OpenCV, Motion-JPEG .avi without)."""
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from demo_image import build_models, build_parser
from vn_celeb_face_recognition_amd import dist as vdist
from vn_celeb_face_recognition_amd.cli_utils import (append_log_to_file, convert_sec_to_max_time_quantity,
                                                     draw_boxes_on_image, export_video_face_recognition, open_frame_source,
                                                     write_rgb)
from vn_celeb_face_recognition_amd.pipeline import FacePipeline
from vn_celeb_face_recognition_amd.video import run_stream, tracker_row  # noqa: F401  (tracker_row: part of this module's surface)


def main(args, pipe, rank, world, source=None, device=None):
    """demo_video.py:46-199 on the resident pipeline; identical control flow for every world size (video.run_stream)."""
    if rank == 0:
        os.makedirs(args.output_frame, exist_ok=True)
        with open(args.output_tracker, 'w') as f:
            f.write('')
        append_log_to_file(args.output_tracker, ['Time', 'Names', 'Frame_idx', 'Bboxes'])
    print('Method: {}'.format(args.inference_method))
    if source is None:
        source = open_frame_source(args.video_path)
    start_time = time.time()

    def on_frame(frame, number, names, boxes):
        img = draw_boxes_on_image(frame, boxes, names) if names else frame
        write_rgb(os.path.join(args.output_frame, 'frame_{}.png'.format(number)), img)

    def log(processed, inf):
        if (processed % args.log_step) == 0:
            print('Processing for frame: {}, time: {}'.format(inf[-1][1], convert_sec_to_max_time_quantity(inf[-1][0])))

    rows, processed = run_stream(source, pipe, args.n_frames, rank, world, device=device,
                                 on_frame=on_frame if args.save_frame_recognized else None, log=log)
    if world > 1:
        tot = torch.tensor([processed], device=device if device is not None else 'cuda')
        dist.all_reduce(tot)
        processed = int(tot.item())
    if rank == 0:
        with open(args.output_tracker, 'a') as f:
            f.write(''.join(rows[k] for k in sorted(rows)))
        processed_time = time.time() - start_time
        print('Saved tracker file in {} ...'.format(args.output_tracker))
        print('FPS for recognition face: {}'.format(int(processed / processed_time)))


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
    if args.output_video and not args.save_frame_recognized:
        raise SystemExit("-ov assembles the annotated frames of --output_frame: add -sfr (the reference writes every frame "
                         "unconditionally, demo_video.py:149)")
    rank, world, local = vdist.init_from_env()
    device = 'cuda:%d' % local
    torch.cuda.set_device(local)
    label2name_df, detection_md, emb_model, classify_model = build_models(args, device)
    pipe = FacePipeline(detection_md, emb_model, classify_model, label2name_df, args.target_face_size, args.recog_threshold,
                        embed_batch=256)
    main(args, pipe, rank, world, device=device)
    if args.output_video and rank == 0:
        export_video_face_recognition(args.output_frame, args.fps_video, args.output_video)     # demo_video.py:285-287
    if world > 1:
        dist.destroy_process_group()
