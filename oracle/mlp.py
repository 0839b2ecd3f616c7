"""Oracle: MLP identity classifier and the name decision around it.

Restates /root/reference/models/mlp_model.py:10-15 (MLPModel.forward) and
/root/reference/demo_image.py:113-147 (identify_person).  Test infrastructure only.
"""
import numpy as np
import torch
import torch.nn.functional as F


def _t(sd, k):
    v = sd[k]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(v)


def mlp_forward(sd, emb):
    """(F,512) -> (F,C) log-probabilities (eval mode: dropout is the identity, mlp_model.py:12)."""
    with torch.no_grad():
        x = F.relu(F.linear(emb.float(), _t(sd, "dense_1.weight"), _t(sd, "dense_1.bias")))
        x = F.linear(x, _t(sd, "dense_2.weight"), _t(sd, "dense_2.bias"))
        return F.log_softmax(x, dim=1)


def identify_person(logp, labels, names, threshold):
    """demo_image.py:117-147 on precomputed log-probs.

    labels/names: the two columns of the label2name CSV (parallel sequences).
    threshold: float (global) or dict str(class)->float (celeb_statistic.py:128-136).
    Returns (list of names, filtered predictions int64).
    """
    logp = logp.detach().cpu() if isinstance(logp, torch.Tensor) else torch.from_numpy(np.asarray(logp))
    n_classes = logp.shape[1]
    pred = torch.argmax(logp, dim=1).numpy()
    probs = torch.exp(logp).numpy()
    out_pred = []
    for i, p in enumerate(pred):
        thr = threshold if isinstance(threshold, float) else threshold[str(int(p))]
        out_pred.append(int(p) if probs[i][p] >= thr else n_classes)
    labels = list(labels)
    out_names = []
    for p in out_pred:
        hit = [n for l, n in zip(labels, names) if l == p]
        out_names.append(hit[0] if hit else "Unknown")
    return out_names, np.asarray(out_pred, dtype=np.int64)
