#!/usr/bin/env python3
"""Train the MLP identity classifier on precomputed embeddings: drop-in for /root/reference/train.py (22-76) with
cfg/train_cfg_emb_classify.json's keys.  The optimisation step runs in libvnface.so (csrc/mlp_train.hip); checkpoints
are the reference's dict (trainer/base_trainer.py:83-105), readable by demo_image.py / demo_video.py (-m).

    python train.py -c cfg/train_cfg_emb_classify.json -d GPU

Only the embedding-classifier training of the README workflow (readme.md:16-34) is covered: model MLPModel, dataset
VNCelebEmbDataset, loss neg_log_llhood, metric accuracy, Adam + ReduceLROnPlateau ("transforms": "none")."""
import argparse
import json

import numpy as np
import torch
from torch.utils.data import DataLoader

from vn_celeb_face_recognition_amd.trainer import ClassificationTrainer, ReduceLROnPlateau, TrainableMLP, VNCelebEmbDataset

SEED = 123   # train.py:16-20


def main(config, run_id=None, device="cuda:0"):
    torch.manual_seed(SEED)
    np.random.seed(SEED)
    if config["model"]["name"] != "MLPModel" or config["train_dataset"]["name"] != "VNCelebEmbDataset":
        raise SystemExit("this build trains MLPModel on VNCelebEmbDataset only (SURVEY.md 8 f-4)")
    if config["optimizer"]["name"] != "Adam" or config["lr_scheduler"]["name"] != "ReduceLROnPlateau":
        raise SystemExit("optimizer Adam + lr_scheduler ReduceLROnPlateau only (cfg/train_cfg_emb_classify.json)")
    if config["trainer"].get("device", "GPU") != "GPU":
        raise SystemExit("this build runs on MI355X only: trainer.device must be GPU (there is no CPU path)")
    train_dataset = VNCelebEmbDataset(**config["train_dataset"]["args"], transforms=None)
    train_loader = DataLoader(dataset=train_dataset, **config["train_data_loader"]["args"])
    val_dataset = VNCelebEmbDataset(**config["val_dataset"]["args"], transforms=None)
    val_loader = DataLoader(dataset=val_dataset, **config["val_data_loader"]["args"])
    oargs = dict(config["optimizer"]["args"])
    bs = max(config["train_data_loader"]["args"]["batch_size"], config["val_data_loader"]["args"]["batch_size"])
    model = TrainableMLP(**config["model"]["args"], lr=oargs.get("lr", 1e-3), betas=oargs.get("betas", (0.9, 0.999)),
                         eps=oargs.get("eps", 1e-8), weight_decay=oargs.get("weight_decay", 0.0), max_batch=bs, device=device)
    sargs = {k: v for k, v in config["lr_scheduler"]["args"].items() if k != "verbose"}
    trainer = ClassificationTrainer(config, model, ReduceLROnPlateau(model, **sargs), run_id=run_id)
    trainer.setup_loader(train_loader, val_loader)
    trainer.train(config["trainer"]["track4plot"])
    return trainer


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description="VNCeleb - Face Recognition")
    ap.add_argument("-c", "--config", default=None, type=str, help="Path of config file")
    ap.add_argument("-d", "--device", default=None, type=str, help="Indices of GPUs")
    args = ap.parse_args()
    with open(args.config) as fp:
        main(json.load(fp))
