#!/usr/bin/env python3
"""Batch-embed a directory of face crops: drop-in for /root/reference/find_embedding.py
(flags 66-75, cal_embedding 45-59, one <stem>.npz per image holding arr_0 = (512,) fp32, 34-42).

Documented deviations from the reference's defects: the transform is
transforms_default (the reference imports a name that does not exist); images are grouped by
shape so mixed 181x181 / 127x127 directories work, and crops that are not 160x160 are
centre-cropped or zero-padded to the encoder's 160x160 input; the empty trailing batch the
reference always appends is skipped.  `-w` may be a local state_dict path; `-dv GPU` is the only
device (there is no CPU path)."""
import argparse
import os
from pathlib import Path

import numpy as np
import torch
from PIL import Image

from vn_celeb_face_recognition_amd import dist as vdist
from vn_celeb_face_recognition_amd.models import InceptionResnetV1
from vn_celeb_face_recognition_amd.pipeline import transforms_default


def create_batch_images(list_files, batch_size):
    n_batchs = len(list_files) // batch_size
    list_batch_files = [list_files[i * batch_size: (i + 1) * batch_size] for i in range(n_batchs)]
    if len(list_files) > n_batchs * batch_size:
        list_batch_files.append(list_files[n_batchs * batch_size:])
    return list_batch_files, n_batchs


def _fit(img, size):
    """centre-crop / zero-pad an HWC uint8 image to size x size (no resampling library involved)."""
    h, w = img.shape[:2]
    out = np.zeros((size, size, 3), dtype=np.uint8)
    sy, sx = max((h - size) // 2, 0), max((w - size) // 2, 0)
    dy, dx = max((size - h) // 2, 0), max((size - w) // 2, 0)
    ch, cw = min(h, size), min(w, size)
    out[dy:dy + ch, dx:dx + cw] = img[sy:sy + ch, sx:sx + cw]
    return out


def create_image_tensors(data_dir_path, list_files, transforms, size=160):
    tensors = []
    for img_file in list_files:
        img = np.asarray(Image.open(str(data_dir_path / img_file)).convert('RGB'))
        tensors.append(transforms(_fit(img, size)))
    return torch.stack(tensors, 0)


def save_embeddings(embeddings, list_files, output_dir):
    for i in range(embeddings.shape[0]):
        emb_path = str(Path(output_dir) / '{}.npz'.format(list_files[i].split('.')[0]))
        np.savez_compressed(emb_path, embeddings[i])
        print('Save embedding for {} ...'.format(list_files[i]))


def cal_embedding(data_dir, batch_size, model, transforms, output_dir, device, rank=0, world=1):
    os.makedirs(output_dir, exist_ok=True)
    model.eval()
    list_files = sorted(os.listdir(data_dir))
    lo, hi = vdist.shard_range(len(list_files), rank, world)     # files are independent units
    list_batch_files, n_batchs = create_batch_images(list_files[lo:hi], batch_size)
    def batches():
        for idx, batch_file in enumerate(list_batch_files):
            print('Processing for {}/{} batchs:'.format(idx, n_batchs))
            yield create_image_tensors(Path(data_dir), batch_file, transforms)

    # batches are independent: up to three are in flight on rotating streams (the next batch's decode and upload
    # and the previous batch's .npz writes overlap the GPU work)
    for idx, emb, ready in model.embed_stream(batches(), lanes=3):
        ready.synchronize()
        save_embeddings(emb.detach().cpu().numpy(), list_batch_files[idx], output_dir)


if __name__ == "__main__":
    args_parser = argparse.ArgumentParser(description='Find embedding vertors for all images in trainning set')
    args_parser.add_argument('-d', '--data_dir', default='train')
    args_parser.add_argument('-bz', '--batch_size', default=10, type=int)
    args_parser.add_argument('-o', '--output_dir', default='train_embedding')
    args_parser.add_argument('-w', '--pre_trained', default='vggface2')
    args_parser.add_argument('-dv', '--device', default='GPU')
    args_parser.add_argument('--compute_dtype', default='f16x2', choices=['f16x2', 'f32', 'bf16', 'f16'],
                             help='f16x2 (default): the <=1e-4 parity path all CLIs share; bf16/f16: faster, 5e-3 / 6e-4 embedding error')
    args = args_parser.parse_args()
    if args.device != 'GPU':
        raise SystemExit("this build runs on MI355X only: use -dv GPU (there is no CPU path)")
    rank, world, local = vdist.init_from_env()
    device = 'cuda:%d' % local
    pre = None if args.pre_trained in ('none', 'None', 'generator') else args.pre_trained
    model = InceptionResnetV1(pretrained=pre, device=device, compute_dtype=args.compute_dtype,
                              max_batch=max(args.batch_size, 1))
    cal_embedding(args.data_dir, args.batch_size, model, transforms_default, args.output_dir, device, rank, world)
