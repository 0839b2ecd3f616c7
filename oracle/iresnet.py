"""Oracle: ArcFace IResNet (IR-100 by default) forward, fp32 torch-CPU, from a flat state_dict.

Restates /root/reference/models/iresnet_encoder.py: IBasicBlock.forward 46-61
(bn1 -> conv3x3 -> bn2 -> PReLU -> conv3x3(stride) -> bn3, + identity/downsample),
IResNet.forward 139-159 (stem conv+bn+prelu, 4 stages, bn2, flatten, fc, features BN1d;
output is NOT L2-normalised).  BN eps = 2e-5 everywhere (37-42, 85, 96, 99).
Test infrastructure only.
"""
import torch
import torch.nn.functional as F

BN_EPS = 2e-5


def _t(sd, k):
    v = sd[k]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(v)


def _bn(sd, p, x):
    return F.batch_norm(x, _t(sd, p + ".running_mean"), _t(sd, p + ".running_var"),
                        _t(sd, p + ".weight"), _t(sd, p + ".bias"), False, 0.0, BN_EPS)


def ibasic_block(sd, p, x, stride, has_down):
    out = _bn(sd, p + ".bn1", x)
    out = F.conv2d(out, _t(sd, p + ".conv1.weight"), None, 1, 1)
    out = _bn(sd, p + ".bn2", out)
    out = F.prelu(out, _t(sd, p + ".prelu.weight"))
    out = F.conv2d(out, _t(sd, p + ".conv2.weight"), None, stride, 1)
    out = _bn(sd, p + ".bn3", out)
    identity = x
    if has_down:
        identity = F.conv2d(x, _t(sd, p + ".downsample.0.weight"), None, stride, 0)
        identity = _bn(sd, p + ".downsample.1", identity)
    return out + identity


def iresnet_forward(sd, x, layers=(3, 13, 30, 3), taps=None):
    """x: (N,3,112,112) fp32 normalised -> (N,512) features."""
    with torch.no_grad():
        x = F.conv2d(x.float(), _t(sd, "conv1.weight"), None, 1, 1)
        x = F.prelu(_bn(sd, "bn1", x), _t(sd, "prelu.weight"))
        if taps is not None:
            taps["stem"] = x
        for li, nblk in enumerate(layers, start=1):
            for b in range(nblk):
                # first block of every stage: stride 2 + downsample (iresnet_encoder.py:87-93,123-131)
                x = ibasic_block(sd, "layer%d.%d" % (li, b), x, 2 if b == 0 else 1, b == 0)
            if taps is not None:
                taps["layer%d" % li] = x
        x = _bn(sd, "bn2", x)
        x = torch.flatten(x, 1)  # Dropout2d is the identity in eval mode
        x = F.linear(x, _t(sd, "fc.weight"), _t(sd, "fc.bias"))
        return _bn(sd, "features", x)
