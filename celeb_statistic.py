#!/usr/bin/env python3
"""Celebrity statistics over a frame stream: drop-in for /root/reference/celeb_statistic.py.

Same flags and files: frames are sub-sampled per second of video with -fidx (l.180-187), queued in batches of
--n_frames, recognised with per-class thresholds (--local_thresholds JSON, or --recog_threshold for every class,
l.127-136), logged to the tracker CSV (Time,Names,Frame_idx[,Bboxes], l.137-147,253-276; an existing tracker file is
re-used, l.393-399) and summarised into the interval JSON (`dynamic_itv` / `fixed_itv`, l.32-107, 401-412).

The recognition itself is the resident MI355X pipeline in throughput mode (FacePipeline.submit: detection and
embedding streams overlap, faces of consecutive batches embedded together).  Input: a directory of frames or a
.npy array of (T,H,W,3) RGB frames with -fps (OpenCV / pafy are not installed: no container decode, no YouTube);
--recog_emotion and seq_fd_vs_aln are outside the hot path and refused."""
import os
import time

import numpy as np
import torch

from demo_image import build_models, build_parser
from vn_celeb_face_recognition_amd.cli_utils import (append_log_to_file, draw_boxes_on_image, open_frame_source,
                                                     write_rgb)
from vn_celeb_face_recognition_amd.pipeline import FacePipeline, identify_names
from vn_celeb_face_recognition_amd.statistics import (build_thresholds, convert_sec_to_max_time_quantity,
                                                      export_json_stat_dynamic_itv, export_json_stat_fixed_itv,
                                                      frame_is_sampled, read_tracker_csv, tracker_header, tracker_row)


def main(args, pipe, threshold, frame_idxes):
    os.makedirs(args.output_frame, exist_ok=True)
    with open(args.output_tracker, 'w') as f:
        f.write('')
    append_log_to_file(args.output_tracker, tracker_header(args.track_bbox))
    frames_iter = open_frame_source(args.video_path)
    fps = frames_iter.fps
    if args.fps_video > 0:
        fps = args.fps_video
    count = processed_frame = 0
    start_time = time.time()
    queue, info, inflight = [], [], []

    def retire(item):
        t, q, inf = item
        counts, boxes, emb, amax, prob = t.result()
        names = identify_names(amax, prob, pipe.classifier.num_classes, pipe.label2name, threshold) if len(boxes) else []
        rows, o = [], 0
        for idx, c in enumerate(counts):
            nm, bx = names[o:o + c], [boxes[k] for k in range(o, o + c)]
            o += c
            if args.save_frame_recognized:
                img = draw_boxes_on_image(q[idx], bx, nm) if nm else q[idx]
                write_rgb(os.path.join(args.output_frame, 'frame_{}.png'.format(inf[idx][1])), img)
            rows.append(tracker_row(inf[idx][0], nm, inf[idx][1], bx, q[idx].shape[:2], args.track_bbox))
        with open(args.output_tracker, 'a') as f:
            f.write(''.join(rows))

    def flush_queue():
        nonlocal processed_frame, queue, info
        if not queue:
            return
        processed_frame += len(queue)
        if (processed_frame % args.log_step) == 0:
            print('Processing for frame: {}, time: {}'.format(info[-1][1], convert_sec_to_max_time_quantity(info[-1][0])))
        frames_dev, _ = pipe.detector._to_device_frames(queue)
        inflight.append((pipe.submit(frames_dev), queue, info))
        queue, info = [], []
        while len(inflight) > 2:
            retire(inflight.pop(0))

    for frame in frames_iter:
        count += 1
        if not frame_is_sampled(count, fps, frame_idxes):
            continue
        queue.append(frame)
        info.append([count / fps, count])
        if len(queue) == args.n_frames:
            flush_queue()
    flush_queue()
    pipe.flush()
    while inflight:
        retire(inflight.pop(0))
    processed_time = time.time() - start_time
    print('Saved tracker file in {} ...'.format(args.output_tracker))
    print('FPS for recognition face: {}'.format(int(processed_frame / max(processed_time, 1e-9))))
    return read_tracker_csv(args.output_tracker)


if __name__ == '__main__':
    p = build_parser('Face recognition on a video')
    p.add_argument('-i', '--video_path', default='video.mp4', type=str)
    p.add_argument('-o', '--output_frame', default='output_frame', type=str)
    p.add_argument('-ot', '--output_tracker', default='tracker.csv', type=str)
    p.add_argument('-sfr', '--save_frame_recognized', action='store_true')
    p.add_argument('-jst', '--json_tracker', default='tracker.json', type=str)
    p.add_argument('-fidx', '--frame_idxes', nargs='+', type=int, required=True)
    p.add_argument('-ign', '--ignored_name', default='Unknown', type=str)
    p.add_argument('-nvi', '--n_video_intervals', default=5, type=int)
    p.add_argument('-tap', '--n_time_appear', default=8, type=int)
    p.add_argument('--statistic_mode', default='dynamic_itv', type=str, help='dynamic_itv or fixed_itv')
    p.add_argument('--time_an_interval', default=5, type=int)
    p.add_argument('--log_step', default=100, type=int)
    p.add_argument('--local_thresholds', default='', type=str)
    p.add_argument('--track_bbox', action='store_true')
    p.add_argument('--youtube_video', action='store_true')
    p.add_argument('--n_frames', default=16, type=int)
    p.add_argument('-fps', '--fps_video', default=0.0, type=float, help='frame rate of a frame directory / .npy input')
    p.set_defaults(recog_threshold=0.7)          # celeb_statistic.py:349 (demo_image's default is 0)
    args = p.parse_args()
    if args.youtube_video:
        raise SystemExit("--youtube_video needs pafy and network access, neither of which this build has")
    if args.inference_method != 'par_fd_vs_aln':
        raise SystemExit("use --inference_method par_fd_vs_aln (seq_fd_vs_aln needs the FAN landmark network, outside "
                         "the hot path and broken in the reference for list input)")
    frame_idxes = list(args.frame_idxes)
    if not os.path.exists(args.output_tracker):
        print('Create tracker file {}'.format(args.output_tracker))
        torch.cuda.set_device(0)
        label2name_df, detection_md, emb_model, classify_model = build_models(args, 'cuda:0')
        if args.local_thresholds != '':
            print('Using local thresholds !')
        else:
            print('Using global a threshold !')
        threshold = build_thresholds(args.local_thresholds, args.num_classes, args.recog_threshold)
        pipe = FacePipeline(detection_md, emb_model, classify_model, label2name_df, args.target_face_size, threshold,
                            embed_batch=256)
        tracker_df = main(args, pipe, threshold, frame_idxes)
    else:
        print('Re-use tracker file {}'.format(args.output_tracker))
        tracker_df = read_tracker_csv(args.output_tracker)
    print('Statistic mode: {}'.format(args.statistic_mode))
    if not args.track_bbox and 'Bboxes' not in tracker_df:
        raise SystemExit("the interval statistics need the Bboxes column: run with --track_bbox (the reference raises "
                         "KeyError here, celeb_statistic.py:81-82)")
    if args.statistic_mode == 'dynamic_itv':
        export_json_stat_dynamic_itv(tracker_df, args.json_tracker, args.n_video_intervals, args.n_time_appear,
                                     args.ignored_name)
    elif args.statistic_mode == 'fixed_itv':
        n_rows_in_itv = args.time_an_interval * len(frame_idxes) * 60
        export_json_stat_fixed_itv(tracker_df, args.json_tracker, n_rows_in_itv, args.n_time_appear, args.ignored_name)
    else:
        print('This statistic mode {} is not supported !'.format(args.statistic_mode))
