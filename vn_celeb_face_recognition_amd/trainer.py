"""MLP training on precomputed embeddings: host-side mirror of the reference's training stack, with
the optimisation step itself in libvnface.so (csrc/mlp_train.hip):

  VNCelebEmbDataset       <- /root/reference/data_loader/vn_celeb_dataset.py:12-47, vn_celeb_emb_dataset.py:6-23
  TrainableMLP            <- /root/reference/models/mlp_model.py:4-15 (+ torch.optim.Adam, train.py:60-62)
  ReduceLROnPlateau       <- torch.optim.lr_scheduler.ReduceLROnPlateau as train.py:64-66 configures it
  MetricTracker           <- /root/reference/utils/utils.py:13-37
  ClassificationTrainer   <- /root/reference/trainer/base_trainer.py:11-190, classification_trainer.py:5-98

Same config keys (cfg/train_cfg_emb_classify.json), same checkpoint dict (base_trainer.py:83-105: arch, epoch,
state_dict, optimizer in torch.optim.Adam's state_dict layout, monitor_best, config), same log_loss.txt.  torch is
plumbing: the DataLoader / sampler (batch order), the initial weights (nn.Linear's init) and the dropout draws come from
torch's CPU generator in the reference's order, so a run seeded like train.py:16-20 follows the reference's loss curve."""
import ctypes
import json
import logging
import os
from collections import OrderedDict
from datetime import datetime
from pathlib import Path

import numpy as np
import torch

from . import _lib

PARAMS = ("dense_1.weight", "dense_1.bias", "dense_2.weight", "dense_2.bias")


class VNCelebEmbDataset(torch.utils.data.Dataset):
    """label json {class: [image names]} + <data_dir>/<stem>.npz (arr_0) -> (embedding, label, path)."""

    def __init__(self, data_dir, label_file, transforms=None):
        self.data_dir = Path(data_dir)
        with open(label_file) as fp:
            self.label_dict = json.load(fp)
        self.transforms = transforms
        self.n_samples = sum(len(v) for v in self.label_dict.values())
        self.n_classes = len(self.label_dict)
        self.img_names, self.labels = [], []
        for k, v in self.label_dict.items():
            names = sorted(v)
            self.img_names += names
            self.labels += len(names) * [int(k)]

    def __len__(self):
        return self.n_samples

    def __getitem__(self, index):
        emb_path = self.data_dir / "{}.npz".format(self.img_names[index].split(".")[0])
        emb = np.load(str(emb_path))["arr_0"]
        data = self.transforms(emb) if self.transforms else torch.from_numpy(emb)
        return data, self.labels[index], str(emb_path)


class TrainableMLP:
    """MLPModel(input_dim, num_classes) + its Adam state, resident on the GPU (vnf_mlp_trainer_*).  Initial weights are
    drawn exactly as `nn.Linear(input_dim, 2048); nn.Linear(2048, num_classes)` draws them (same generator calls)."""

    def __init__(self, input_dim, num_classes, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_batch=1024,
                 device="cuda:0"):
        self.input_dim, self.num_classes, self.max_batch = int(input_dim), int(num_classes), int(max_batch)
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), tuple(betas), float(eps), float(weight_decay)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("training runs on MI355X only (there is no CPU path)")
        self.training = True
        d1, d2 = torch.nn.Linear(self.input_dim, 2048), torch.nn.Linear(2048, self.num_classes)
        sd = OrderedDict([("dense_1.weight", d1.weight), ("dense_1.bias", d1.bias), ("dense_2.weight", d2.weight),
                          ("dense_2.bias", d2.bias)])
        self._shapes = {k: tuple(v.shape) for k, v in sd.items()}
        lib = _lib.load()
        dev = self.device.index if self.device.index is not None else torch.cuda.current_device()
        with torch.cuda.device(dev):
            _lib.check(lib.vnf_init(dev))
            descs, n, keep = _lib.make_descs(OrderedDict((k, v.detach()) for k, v in sd.items()))
            h = ctypes.c_void_p()
            _lib.check(lib.vnf_mlp_trainer_create(descs, n, self.input_dim, self.num_classes, self.max_batch, self.betas[0],
                                                  self.betas[1], self.eps, self.weight_decay, ctypes.byref(h)))
            del keep
        self._h = h
        self._loss = torch.zeros(1, dtype=torch.float32, device=self.device)
        self._hits = torch.zeros(1, dtype=torch.int32, device=self.device)

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None:
                _lib.load().vnf_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def train(self, mode=True):
        self.training = bool(mode)
        return self

    def eval(self):
        return self.train(False)

    def to(self, device):
        return self

    def step(self, data, target, train):
        """One batch: forward + loss (+ backward + Adam when train).  Returns (mean NLL, correct count).  In training
        mode the dropout factors of F.dropout(x, 0.5) are drawn from torch's CPU generator, as the reference's forward
        on a CPU tensor would draw them (models/mlp_model.py:12)."""
        b = int(data.shape[0])
        x = data.to(self.device, dtype=torch.float32).contiguous()
        th = torch.as_tensor(target).to(dtype=torch.int64)
        if th.numel() and (int(th.min()) < 0 or int(th.max()) >= self.num_classes):
            # torch's nll_loss raises on such a label (trainer/classification_trainer.py:22 F.nll_loss)
            raise IndexError("Target %d is out of bounds." % int(th.max() if int(th.max()) >= self.num_classes else th.min()))
        t = th.to(self.device).contiguous()
        mask = None
        if train:
            mask = (torch.empty((b, 2048), dtype=torch.float32).bernoulli_(0.5) / 0.5).to(self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().vnf_mlp_train_step(
                self._h, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(t.data_ptr()), b,
                ctypes.c_void_p(mask.data_ptr()) if mask is not None else None, self.lr, 1 if train else 0,
                ctypes.c_void_p(self._loss.data_ptr()), ctypes.c_void_p(self._hits.data_ptr()), _lib.current_stream_ptr()))
        return float(self._loss.item()), int(self._hits.item())

    # ---- checkpoint access
    def _get(self, name, kind):
        a = np.empty(self._shapes[name], dtype=np.float32)
        _lib.check(_lib.load().vnf_mlp_trainer_get(self._h, name.encode(), kind, a.ctypes.data, a.size))
        return torch.from_numpy(a)

    def _set(self, name, kind, value):
        a = np.ascontiguousarray(torch.as_tensor(value).detach().cpu().float().numpy())
        if a.shape != self._shapes[name]:
            raise RuntimeError("size mismatch for %s: %s vs %s" % (name, a.shape, self._shapes[name]))
        _lib.check(_lib.load().vnf_mlp_trainer_set(self._h, name.encode(), kind, a.ctypes.data, a.size))

    def state_dict(self):
        return OrderedDict((k, self._get(k, 0)) for k in PARAMS)

    def load_state_dict(self, sd):
        for k in PARAMS:
            if k not in sd:
                raise RuntimeError("Missing key(s) in state_dict: %s" % k)
            self._set(k, 0, sd[k])

    def _step_count(self, value=None):
        c = ctypes.c_int64(0 if value is None else int(value))
        _lib.check(_lib.load().vnf_mlp_trainer_step_count(self._h, ctypes.byref(c), 0 if value is None else 1))
        return int(c.value)

    def optimizer_state_dict(self):
        """torch.optim.Adam.state_dict() layout (params 0..3 in state_dict order), so the reference's
        resume_checkpoint (base_trainer.py:73-80) can load it into a torch Adam and vice versa."""
        step = float(self._step_count())
        state = {i: {"step": torch.tensor(step), "exp_avg": self._get(k, 1), "exp_avg_sq": self._get(k, 2)}
                 for i, k in enumerate(PARAMS)} if step > 0 else {}
        group = torch.optim.Adam([torch.zeros(1)], lr=self.lr, betas=self.betas, eps=self.eps,
                                 weight_decay=self.weight_decay).state_dict()["param_groups"][0]
        group["params"] = list(range(len(PARAMS)))
        return {"state": state, "param_groups": [group]}

    def load_optimizer_state_dict(self, osd):
        g = osd["param_groups"][0]
        self.lr = float(g["lr"])
        if osd["state"]:
            for i, k in enumerate(PARAMS):
                self._set(k, 1, osd["state"][i]["exp_avg"])
                self._set(k, 2, osd["state"][i]["exp_avg_sq"])
            self._step_count(int(float(osd["state"][0]["step"])))


class ReduceLROnPlateau:
    """torch.optim.lr_scheduler.ReduceLROnPlateau (cooldown 0, eps 1e-8) acting on TrainableMLP.lr."""

    def __init__(self, model, mode="min", factor=0.1, patience=10, threshold=1e-4, threshold_mode="rel", cooldown=0,
                 min_lr=0.0, eps=1e-8, verbose=False):
        self.model, self.mode, self.factor, self.patience = model, mode, factor, patience
        self.threshold, self.threshold_mode, self.cooldown, self.min_lr, self.eps = threshold, threshold_mode, cooldown, min_lr, eps
        self.best = float("inf") if mode == "min" else -float("inf")
        self.num_bad_epochs, self.cooldown_counter = 0, 0

    def _better(self, a, best):
        if self.mode == "min":
            return a < (best * (1.0 - self.threshold) if self.threshold_mode == "rel" else best - self.threshold)
        return a > (best * (self.threshold + 1.0) if self.threshold_mode == "rel" else best + self.threshold)

    def step(self, metric):
        current = float(metric)
        if self._better(current, self.best):
            self.best, self.num_bad_epochs = current, 0
        else:
            self.num_bad_epochs += 1
        if self.cooldown_counter > 0:
            self.cooldown_counter -= 1
            self.num_bad_epochs = 0
        if self.num_bad_epochs > self.patience:
            new_lr = max(self.model.lr * self.factor, self.min_lr)
            if self.model.lr - new_lr > self.eps:
                self.model.lr = new_lr
            self.cooldown_counter, self.num_bad_epochs = self.cooldown, 0


class MetricTracker:
    def __init__(self, *keys):
        self.keys = keys
        self.reset()

    def reset(self):
        self.total = {k: 0.0 for k in self.keys}
        self.counts = {k: 0 for k in self.keys}

    def update(self, key, value, n=1):
        self.total[key] += value * n
        self.counts[key] += n

    def avg(self, key):
        return self.total[key] / self.counts[key] if self.counts[key] else 0

    def result(self):
        return {k: self.avg(k) for k in self.keys}


class ClassificationTrainer:
    def __init__(self, config, model, lr_scheduler, run_id=None):
        self.config, self.model, self.lr_scheduler = config, model, lr_scheduler
        tc = config["trainer"]
        self.start_epoch, self.epochs = 1, tc["epochs"]
        self.tracked_metric, self.mode_monitor = tc["tracked_metric"]
        self.early_stop, self.save_step, self.log_step = tc["patience"], tc["save_period"], tc["log_step"]
        self.loss_name, self.metric_names = config["loss"], list(config["metrics"])
        if self.loss_name != "neg_log_llhood" or self.metric_names != ["accuracy"]:
            raise NotImplementedError("the fused step computes NLLLoss and accuracy (losses/__init__.py, metrics.py)")
        self.train_loss, self.train_metrics = MetricTracker(self.loss_name), MetricTracker(*self.metric_names)
        self.val_loss, self.val_metrics = MetricTracker(self.loss_name), MetricTracker(*self.metric_names)
        run_id = run_id or datetime.now().strftime(r"%m%d_%H%M%S")
        self.save_dir = Path(tc["save_dir"]) / "models" / run_id
        self.log_dir = Path(tc["save_dir"]) / "logs" / run_id
        os.makedirs(self.save_dir, exist_ok=True)
        os.makedirs(self.log_dir, exist_ok=True)
        logging.basicConfig(level=logging.INFO)
        self.logger = logging.getLogger("trainer")
        self.do_val, self.val_step = tc["do_validation"], tc["validation_step"]
        self.mnt_best = float("inf") if self.mode_monitor == "min" else -float("inf")
        if tc["resume_path"] != "":
            self.resume_checkpoint(tc["resume_path"])

    def setup_loader(self, train_loader, val_loader):
        self.train_loader, self.val_loader = train_loader, val_loader

    def resume_checkpoint(self, checkpoint_path):
        cp = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
        self.logger.info("Loading checkpoint: {} ...".format(checkpoint_path))
        self.start_epoch = cp["epoch"] + 1
        self.mnt_best = cp["monitor_best"]
        self.model.load_state_dict(cp["state_dict"])
        self.model.load_optimizer_state_dict(cp["optimizer"])
        self.logger.info("Checkpoint loaded. Resume training from epoch {}".format(self.start_epoch))

    def save_checkpoint(self, epoch, save_best):
        state = {"arch": "MLPModel", "epoch": epoch, "state_dict": self.model.state_dict(),
                 "optimizer": self.model.optimizer_state_dict(), "monitor_best": self.mnt_best, "config": self.config}
        filename = str(self.save_dir / "checkpoint-epoch{}.pth".format(epoch))
        torch.save(state, filename)
        self.logger.info("Saving checkpoint: {} ...".format(filename))
        if save_best:
            torch.save(state, str(self.save_dir / "model_best.pth"))
            self.logger.info("Saving current best: model_best.pth ...")

    def _train_epoch(self, epoch):
        self.model.train()
        for t in (self.train_loss, self.train_metrics, self.val_loss, self.val_metrics):
            t.reset()
        for batch_idx, (data, target, _id) in enumerate(self.train_loader):
            loss, hits = self.model.step(data, target, train=True)
            self.train_loss.update(self.loss_name, loss)
            self.train_metrics.update("accuracy", hits / data.size(0), n=data.size(0))
            if batch_idx % self.log_step == 0:
                self.logger.info("Train Epoch: {} [{}]/[{}] with NLLLoss, Loss: {:.6f}".format(
                    epoch, batch_idx, len(self.train_loader), self.train_loss.avg(self.loss_name)))
                self.logger.info("accuracy: {:.6f}".format(self.train_metrics.avg("accuracy")))
        log = self.train_loss.result()
        log.update(self.train_metrics.result())
        if self.do_val and (epoch % self.val_step == 0):
            log.update(self._validate_epoch(epoch))
        if isinstance(self.lr_scheduler, ReduceLROnPlateau):
            self.lr_scheduler.step(self.val_loss.avg(self.loss_name))
        return log

    def _validate_epoch(self, epoch):
        self.model.eval()
        self.val_loss.reset()
        self.val_metrics.reset()
        self.logger.info("Validation: ")
        for batch_idx, (data, target, _id) in enumerate(self.val_loader):
            loss, hits = self.model.step(data, target, train=False)
            self.val_loss.update(self.loss_name, loss)
            self.val_metrics.update("accuracy", hits / data.size(0), n=data.size(0))
        log = self.val_loss.result()
        log.update(self.val_metrics.result())
        return {"val_{}".format(k): v for k, v in log.items()}

    def train(self, track4plot=False):
        not_improve_count = 0
        if track4plot:
            self.track4plot = str(self.log_dir / "log_loss.txt")
            with open(self.track4plot, "a") as f:
                f.write(",".join(["Epoch", "Train_loss", "Validation_loss"]) + "\n")
        for epoch in range(self.start_epoch, self.epochs + 1):
            result = self._train_epoch(epoch)
            if track4plot:
                with open(self.track4plot, "a") as f:
                    f.write(",".join(str(x) for x in [epoch, result.get(self.loss_name), result.get("val_" + self.loss_name)]) + "\n")
            log = {"epoch": epoch}
            log.update(result)
            for key, value in log.items():
                self.logger.info("    {:15s}: {}".format(str(key), value))
            best = False
            tracked = log.get(self.tracked_metric)
            if tracked:
                improved = (self.mode_monitor == "min" and tracked < self.mnt_best) or \
                           (self.mode_monitor == "max" and tracked > self.mnt_best)
                if improved:
                    self.mnt_best, not_improve_count, best = tracked, 0, True
                else:
                    not_improve_count += 1
            if not_improve_count > self.early_stop:
                self.logger.info("Validation performance didn't improve for {} epochs. Training stops.".format(self.early_stop))
                break
            if epoch % self.save_step == 0:
                self.save_checkpoint(epoch, save_best=best)
