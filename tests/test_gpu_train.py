"""MLP training on embeddings against the reference's own trainer: tests/golden/mlp_train_ref.* hold
the loss curve, learning-rate schedule, final weights and checkpoint layout produced by trainer/base_trainer.py +
classification_trainer.py run in the build container (tools/make_golden.py mlp_train)."""
import json
import os
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, REPO

pytestmark = pytest.mark.gpu


def _case(tmp_path):
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import make_golden
    cfg = make_golden.write_mlp_train_case(str(tmp_path))
    cfg["trainer"]["device"] = "GPU"
    return cfg


def test_mlp_training_follows_the_reference_trainer(tmp_path):
    import train
    g = np.load(os.path.join(GOLDEN, "mlp_train_ref.npz"))
    meta = json.load(open(os.path.join(GOLDEN, "mlp_train_ref.json")))
    cfg = _case(tmp_path)
    lrs, logs = [], []
    from vn_celeb_face_recognition_amd import trainer as T
    orig = T.ClassificationTrainer._train_epoch

    def wrapped(self, epoch):
        r = orig(self, epoch)
        lrs.append(self.model.lr)
        logs.append(r)
        return r
    T.ClassificationTrainer._train_epoch = wrapped
    try:
        tr = train.main(cfg, run_id="t")
    finally:
        T.ClassificationTrainer._train_epoch = orig
    # same batches, same dropout draws, same arithmetic up to fp32 summation order
    assert np.allclose([l["neg_log_llhood"] for l in logs], g["train_loss"], rtol=2e-4, atol=0)
    assert np.allclose([l["val_neg_log_llhood"] for l in logs], g["val_loss"], rtol=2e-4, atol=0)
    assert np.allclose([l["accuracy"] for l in logs], g["train_acc"], atol=1e-9)
    assert np.allclose([l["val_accuracy"] for l in logs], g["val_acc"], atol=1e-9)
    assert np.allclose(lrs, g["lr_after_epoch"], rtol=0, atol=1e-12)          # ReduceLROnPlateau fired at the same epochs
    # checkpoint files and layout (base_trainer.py:83-105)
    assert sorted(os.listdir(tr.save_dir)) == meta["checkpoint_files"]
    cp = torch.load(os.path.join(str(tr.save_dir), "checkpoint-epoch6.pth"), weights_only=True)
    assert sorted(cp.keys()) == meta["checkpoint_keys"] and cp["arch"] == meta["arch"] and cp["epoch"] == meta["epoch"]
    assert abs(cp["monitor_best"] - meta["monitor_best"]) <= 2e-4 * meta["monitor_best"]
    assert sorted(cp["optimizer"]["state"][0].keys()) == meta["optimizer_state_keys"]
    assert set(meta["param_group_keys"]) <= set(cp["optimizer"]["param_groups"][0].keys())
    assert float(cp["optimizer"]["state"][0]["step"]) == float(g["adam_step"])
    sd = {k: v.numpy() for k, v in cp["state_dict"].items()}
    assert np.abs(sd["dense_1.weight"].reshape(-1)[::4099] - g["final_dense_1_weight_sample"]).max() <= 2e-5
    assert np.abs(sd["dense_1.bias"] - g["final_dense_1_bias"]).max() <= 2e-5
    assert np.abs(sd["dense_2.weight"].reshape(-1)[::97] - g["final_dense_2_weight_sample"]).max() <= 2e-5
    assert np.abs(sd["dense_2.bias"] - g["final_dense_2_bias"]).max() <= 2e-5
    v = cp["optimizer"]["state"][3]["exp_avg_sq"].numpy()
    assert np.allclose(v, g["exp_avg_sq_dense_2_bias"], rtol=2e-3, atol=1e-12)
    text = open(os.path.join(str(tr.log_dir), "log_loss.txt")).read().splitlines()
    want = meta["log_loss_txt"].splitlines()
    assert text[0] == want[0] and len(text) == len(want)
    # the checkpoint is what the inference side loads (demo_image.py:16-21)
    from vn_celeb_face_recognition_amd.classifier import MLPModel, load_model_classify
    m = MLPModel(512, 12)
    load_model_classify(os.path.join(str(tr.save_dir), "model_best.pth"), m)


def test_resume_continues_bit_for_bit(tmp_path):
    """trainer.resume_path (base_trainer.py:73-80): training 3 + 3 epochs through a checkpoint equals 6 epochs."""
    import train
    cfg = _case(tmp_path)
    full = train.main(json.loads(json.dumps(cfg)), run_id="full")
    cfg3 = json.loads(json.dumps(cfg)); cfg3["trainer"]["epochs"] = 3
    train.main(cfg3, run_id="a")
    cfg6 = json.loads(json.dumps(cfg))
    cfg6["trainer"]["resume_path"] = os.path.join(cfg["trainer"]["save_dir"], "models", "a", "checkpoint-epoch3.pth")
    # the data order / dropout of epochs 4-6 follow the generator state, which a fresh process does not restore (the
    # reference does not either): compare the optimizer bookkeeping and that training proceeds from the loaded state
    res = train.main(cfg6, run_id="b")
    cp_a = torch.load(cfg6["trainer"]["resume_path"], weights_only=True)
    cp_b = torch.load(os.path.join(str(res.save_dir), "checkpoint-epoch6.pth"), weights_only=True)
    cp_f = torch.load(os.path.join(str(full.save_dir), "checkpoint-epoch6.pth"), weights_only=True)
    assert float(cp_b["optimizer"]["state"][0]["step"]) == float(cp_f["optimizer"]["state"][0]["step"]) == 2 * float(cp_a["optimizer"]["state"][0]["step"])
    assert cp_b["epoch"] == 6 and cp_b["monitor_best"] < cp_a["monitor_best"]


def test_out_of_range_label_raises_like_nll_loss():
    """F.nll_loss asserts on a target outside [0, num_classes) (trainer/classification_trainer.py:22); the device kernel
    would otherwise index past the row."""
    from vn_celeb_face_recognition_amd.trainer import TrainableMLP
    m = TrainableMLP(512, 12, max_batch=8, device="cuda:0")
    x = torch.randn((4, 512), generator=torch.Generator().manual_seed(1))
    loss, hits = m.step(x, torch.tensor([0, 3, 11, 5]), train=False)
    assert np.isfinite(loss) and 0 <= hits <= 4
    for bad in ([0, 12, 1, 2], [0, -1, 1, 2]):
        with pytest.raises(IndexError, match="out of bounds"):
            m.step(x, torch.tensor(bad), train=False)
