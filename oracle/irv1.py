"""Oracle: InceptionResnetV1 forward, fp32 torch-CPU, from a flat state_dict.

Restates /root/reference/models/inception_resnet_v1.py:
  BasicConv2d 12-33 (conv no-bias -> BN eps 1e-3 -> ReLU), Block35 36-67, Block17 70-95,
  Block8 98-126, Mixed_6a 129-149, Mixed_7a 152-181, InceptionResnetV1.forward 272-303.
Test infrastructure only (see oracle/__init__.py).
"""
import torch
import torch.nn.functional as F

BN_EPS = 1e-3


def _t(sd, k):
    v = sd[k]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(v)


def basic_conv(sd, p, x, stride=1, padding=0):
    # inception_resnet_v1.py:29-33
    x = F.conv2d(x, _t(sd, p + ".conv.weight"), None, stride, padding)
    x = F.batch_norm(x, _t(sd, p + ".bn.running_mean"), _t(sd, p + ".bn.running_var"),
                     _t(sd, p + ".bn.weight"), _t(sd, p + ".bn.bias"), False, 0.0, BN_EPS)
    return F.relu(x)


def _res(sd, p, x, cat, scale, relu=True):
    # e.g. inception_resnet_v1.py:63-67
    out = F.conv2d(cat, _t(sd, p + ".conv2d.weight"), _t(sd, p + ".conv2d.bias"))
    out = out * scale + x
    return F.relu(out) if relu else out


def block35(sd, p, x, scale):
    x0 = basic_conv(sd, p + ".branch0", x)
    x1 = basic_conv(sd, p + ".branch1.0", x)
    x1 = basic_conv(sd, p + ".branch1.1", x1, 1, 1)
    x2 = basic_conv(sd, p + ".branch2.0", x)
    x2 = basic_conv(sd, p + ".branch2.1", x2, 1, 1)
    x2 = basic_conv(sd, p + ".branch2.2", x2, 1, 1)
    return _res(sd, p, x, torch.cat((x0, x1, x2), 1), scale)


def block17(sd, p, x, scale):
    x0 = basic_conv(sd, p + ".branch0", x)
    x1 = basic_conv(sd, p + ".branch1.0", x)
    x1 = basic_conv(sd, p + ".branch1.1", x1, 1, (0, 3))
    x1 = basic_conv(sd, p + ".branch1.2", x1, 1, (3, 0))
    return _res(sd, p, x, torch.cat((x0, x1), 1), scale)


def block8(sd, p, x, scale, relu=True):
    x0 = basic_conv(sd, p + ".branch0", x)
    x1 = basic_conv(sd, p + ".branch1.0", x)
    x1 = basic_conv(sd, p + ".branch1.1", x1, 1, (0, 1))
    x1 = basic_conv(sd, p + ".branch1.2", x1, 1, (1, 0))
    return _res(sd, p, x, torch.cat((x0, x1), 1), scale, relu)


def mixed_6a(sd, x):
    x0 = basic_conv(sd, "mixed_6a.branch0", x, 2)
    x1 = basic_conv(sd, "mixed_6a.branch1.0", x)
    x1 = basic_conv(sd, "mixed_6a.branch1.1", x1, 1, 1)
    x1 = basic_conv(sd, "mixed_6a.branch1.2", x1, 2)
    x2 = F.max_pool2d(x, 3, 2)
    return torch.cat((x0, x1, x2), 1)


def mixed_7a(sd, x):
    x0 = basic_conv(sd, "mixed_7a.branch0.0", x)
    x0 = basic_conv(sd, "mixed_7a.branch0.1", x0, 2)
    x1 = basic_conv(sd, "mixed_7a.branch1.0", x)
    x1 = basic_conv(sd, "mixed_7a.branch1.1", x1, 2)
    x2 = basic_conv(sd, "mixed_7a.branch2.0", x)
    x2 = basic_conv(sd, "mixed_7a.branch2.1", x2, 1, 1)
    x2 = basic_conv(sd, "mixed_7a.branch2.2", x2, 2)
    x3 = F.max_pool2d(x, 3, 2)
    return torch.cat((x0, x1, x2, x3), 1)


def irv1_forward(sd, x, taps=None):
    """x: (N,3,160,160) fp32 already normalised -> (N,512) L2-normalised embeddings.

    `taps`, if a dict, receives named intermediate activations (NCHW) for per-stage parity.
    """
    def tap(name, v):
        if taps is not None:
            taps[name] = v
        return v

    with torch.no_grad():
        x = x.float()
        x = tap("conv2d_1a", basic_conv(sd, "conv2d_1a", x, 2))
        x = tap("conv2d_2a", basic_conv(sd, "conv2d_2a", x))
        x = tap("conv2d_2b", basic_conv(sd, "conv2d_2b", x, 1, 1))
        x = tap("maxpool_3a", F.max_pool2d(x, 3, 2))
        x = tap("conv2d_3b", basic_conv(sd, "conv2d_3b", x))
        x = tap("conv2d_4a", basic_conv(sd, "conv2d_4a", x))
        x = tap("conv2d_4b", basic_conv(sd, "conv2d_4b", x, 2))
        for i in range(5):
            x = block35(sd, "repeat_1.%d" % i, x, 0.17)
        tap("repeat_1", x)
        x = tap("mixed_6a", mixed_6a(sd, x))
        for i in range(10):
            x = block17(sd, "repeat_2.%d" % i, x, 0.10)
        tap("repeat_2", x)
        x = tap("mixed_7a", mixed_7a(sd, x))
        for i in range(5):
            x = block8(sd, "repeat_3.%d" % i, x, 0.20)
        tap("repeat_3", x)
        x = tap("block8", block8(sd, "block8", x, 1.0, relu=False))
        x = F.adaptive_avg_pool2d(x, 1)
        # dropout is the identity in eval mode (inception_resnet_v1.py:295)
        x = F.linear(x.view(x.shape[0], -1), _t(sd, "last_linear.weight"))
        x = F.batch_norm(x, _t(sd, "last_bn.running_mean"), _t(sd, "last_bn.running_var"),
                         _t(sd, "last_bn.weight"), _t(sd, "last_bn.bias"), False, 0.0, BN_EPS)
        tap("last_bn", x)
        return F.normalize(x, p=2, dim=1)


def irv1_forward_quantised(sd, x, qdtype):
    """The same forward with the HIP 16-bit path's quantisation points: BN-folded conv weights and
    every stored activation rounded to `qdtype` (torch.bfloat16 / torch.float16), accumulation,
    bias, residual and activation in fp32.  Used only to separate kernel correctness from
    16-bit precision in the GPU parity tests."""
    def q(t):
        return t.to(qdtype).float()

    def fold(p):
        w = _t(sd, p + ".conv.weight")
        s = _t(sd, p + ".bn.weight").double() / torch.sqrt(_t(sd, p + ".bn.running_var").double() + BN_EPS)
        b = _t(sd, p + ".bn.bias").double() - _t(sd, p + ".bn.running_mean").double() * s
        return q(w * s.float().view(-1, 1, 1, 1)), b.float()

    def bc(p, x, stride=1, padding=0):
        w, b = fold(p)
        return q(F.relu(F.conv2d(x, w, b, stride, padding)))

    def res(p, x, cat, scale, relu=True):
        w = q(_t(sd, p + ".conv2d.weight") * scale)
        out = F.conv2d(cat, w, _t(sd, p + ".conv2d.bias") * scale) + x
        return q(F.relu(out) if relu else out)

    with torch.no_grad():
        x = q(x.float())
        x = bc("conv2d_1a", x, 2); x = bc("conv2d_2a", x); x = bc("conv2d_2b", x, 1, 1)
        x = F.max_pool2d(x, 3, 2)
        x = bc("conv2d_3b", x); x = bc("conv2d_4a", x); x = bc("conv2d_4b", x, 2)
        for i in range(5):
            p = "repeat_1.%d" % i
            x0 = bc(p + ".branch0", x)
            x1 = bc(p + ".branch1.1", bc(p + ".branch1.0", x), 1, 1)
            x2 = bc(p + ".branch2.2", bc(p + ".branch2.1", bc(p + ".branch2.0", x), 1, 1), 1, 1)
            x = res(p, x, torch.cat((x0, x1, x2), 1), 0.17)
        x0 = bc("mixed_6a.branch0", x, 2)
        x1 = bc("mixed_6a.branch1.2", bc("mixed_6a.branch1.1", bc("mixed_6a.branch1.0", x), 1, 1), 2)
        x = torch.cat((x0, x1, F.max_pool2d(x, 3, 2)), 1)
        for i in range(10):
            p = "repeat_2.%d" % i
            x0 = bc(p + ".branch0", x)
            x1 = bc(p + ".branch1.2", bc(p + ".branch1.1", bc(p + ".branch1.0", x), 1, (0, 3)), 1, (3, 0))
            x = res(p, x, torch.cat((x0, x1), 1), 0.10)
        x0 = bc("mixed_7a.branch0.1", bc("mixed_7a.branch0.0", x), 2)
        x1 = bc("mixed_7a.branch1.1", bc("mixed_7a.branch1.0", x), 2)
        x2 = bc("mixed_7a.branch2.2", bc("mixed_7a.branch2.1", bc("mixed_7a.branch2.0", x), 1, 1), 2)
        x = torch.cat((x0, x1, x2, F.max_pool2d(x, 3, 2)), 1)
        for i in range(6):
            p = "repeat_3.%d" % i if i < 5 else "block8"
            x0 = bc(p + ".branch0", x)
            x1 = bc(p + ".branch1.2", bc(p + ".branch1.1", bc(p + ".branch1.0", x), 1, (0, 1)), 1, (1, 0))
            x = res(p, x, torch.cat((x0, x1), 1), 0.20 if i < 5 else 1.0, relu=i < 5)
        x = q(F.adaptive_avg_pool2d(x, 1).view(x.shape[0], -1))
        s = _t(sd, "last_bn.weight").double() / torch.sqrt(_t(sd, "last_bn.running_var").double() + BN_EPS)
        b = (_t(sd, "last_bn.bias").double() - _t(sd, "last_bn.running_mean").double() * s).float()
        x = F.linear(x, q(_t(sd, "last_linear.weight") * s.float().view(-1, 1)), b)
        return F.normalize(x, p=2, dim=1)
