#!/usr/bin/env python3
"""Face recognition on one image: drop-in for /root/reference/demo_image.py (CLI 308-425).
Same flags; models are resolved by name from JSON kwargs (361-374).  `par_fd_vs_aln` is the
functional method (the reference's default `seq_fd_vs_aln` needs the FAN landmark net and is
broken for its list argument)."""
import argparse

import numpy as np

from vn_celeb_face_recognition_amd import models as model_md
from vn_celeb_face_recognition_amd.classifier import load_model_classify
from vn_celeb_face_recognition_amd.cli_utils import draw_boxes_on_image, read_json, read_label2name, read_rgb, write_rgb
from vn_celeb_face_recognition_amd.pipeline import (center_point_dict, parallel_detect_and_align, recognize_celeb,
                                                    transforms_default)


def build_parser(desc):
    p = argparse.ArgumentParser(description=desc)
    p.add_argument('-fs', '--face_size', default=160, type=int)
    p.add_argument('-mfs', '--min_face_size', default=50, type=int)
    p.add_argument('-m', '--classify_model', default='model_best.pth', type=str)
    p.add_argument('-l2n', '--label2name', default='label2name.csv', type=str)
    p.add_argument('-w', '--pre_trained_emb', default='vggface2', type=str)
    p.add_argument('-dv', '--device', default='GPU', type=str)
    p.add_argument('-id', '--input_dim_emb', default=512, type=int)
    p.add_argument('-nc', '--num_classes', default=1001, type=int)
    p.add_argument('-enc', '--encoder', default='InceptionResnetV1', type=str)
    p.add_argument('-det', '--detection', default='MTCNN', type=str)
    p.add_argument('-eargs', '--encoder_args', default='cfg/embedding/iresnet100_enc.json', type=str)
    p.add_argument('-dargs', '--detection_args', default='cfg/detection/mtcnn.json', type=str)
    p.add_argument('-tg_fs', '--target_face_size', default=112, type=int)
    p.add_argument('--inference_method', default='seq_fd_vs_aln', type=str)
    p.add_argument('--min_dim_box', default=50, type=int)
    p.add_argument('--box_ratio', default=2.0, type=float)
    p.add_argument('--recog_threshold', default=0.0, type=float)
    p.add_argument('--recog_emotion', action='store_true')
    p.add_argument('-emt', '--emotion', default='resnet_2branch_50', type=str)
    p.add_argument('-emtargs', '--emotion_args', default='cfg/emotion/resnet50_2_branch.json', type=str)
    p.add_argument('-t2i', '--etag2idx_file', default='meta_data/emotion_recognition/etag2idx.pkl.keep', type=str)
    p.add_argument('--topk_emotions', default=6, type=int)
    return p


def build_models(args, device):
    """demo_image.py:359-376 / demo_video.py:257-275."""
    if args.device != 'GPU':
        raise SystemExit("this build runs on MI355X only: use -dv GPU (there is no CPU path)")
    if args.recog_emotion:
        raise SystemExit("--recog_emotion: the emotion network is outside the hot path (SURVEY.md section 8)")
    label2name_df = read_label2name(args.label2name)
    det_args = read_json(args.detection_args)
    det_args['device'] = device
    detection_md = getattr(model_md, args.detection)(**det_args)
    detection_md.eval()
    emb_model = getattr(model_md, args.encoder)(**read_json(args.encoder_args)).to(device)
    classify_model = model_md.MLPModel(args.input_dim_emb, args.num_classes)
    load_model_classify(args.classify_model, classify_model)
    classify_model = classify_model.to(device)
    return label2name_df, detection_md, emb_model, classify_model


if __name__ == '__main__':
    args_parser = build_parser('Face recognition on a image')
    args_parser.add_argument('-i', '--image_path', default='demo.png', type=str)
    args_parser.add_argument('-o', '--output_path', default='demo_recognition.png', type=str)
    args = args_parser.parse_args()
    device = 'cuda:0'
    label2name_df, detection_md, emb_model, classify_model = build_models(args, device)
    target_fs = (args.target_face_size, args.target_face_size)
    center_point = center_point_dict[str(target_fs)]
    rgb_image = read_rgb(args.image_path)
    rgb_images = [rgb_image]
    if args.inference_method == 'par_fd_vs_aln':
        bth_alg_faces, bth_chosen_boxes = parallel_detect_and_align(rgb_images, detection_md, center_point, target_fs, True)
    elif args.inference_method == 'seq_fd_vs_aln':
        raise SystemExit("seq_fd_vs_aln needs the face_alignment (FAN) landmark network, which is outside the hot path "
                         "and broken in the reference for list input; use --inference_method par_fd_vs_aln")
    else:
        raise SystemExit('Do not support {} method.'.format(args.inference_method))
    bth_names = recognize_celeb(bth_alg_faces, device, emb_model, classify_model, transforms_default, label2name_df,
                                args.recog_threshold)
    np_image_recog = draw_boxes_on_image(rgb_image, bth_chosen_boxes[0], bth_names[0])
    write_rgb(args.output_path, np_image_recog)
    print('Face recognized image saved at {} ...'.format(args.output_path))
