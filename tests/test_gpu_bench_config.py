"""Parity of the configuration bench.py actually times: bs=256, bf16, `set_streams(1)` +
`set_contexts(3)` on three rotating streams.  At n=256 the engine takes branches no small test reaches -- the stem
runs in sub-batches of 128 (engine.cpp groups), three activation contexts rotate, the fused inception-block kernels
see whole-chip grids -- so the timed path itself is compared with the plain one and with the reference's goldens."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, seeded_normal

pytestmark = pytest.mark.gpu

ROWS = [0, 127, 128, 191, 192, 255]   # either side of the stem's 128-image sub-batch and of the >=192 two-stream fork


def _golden_inputs():
    g = np.load(os.path.join(GOLDEN, "irv1_seed0.npz"))
    x = seeded_normal((6, 3, 160, 160), g["input_seed"])
    x[4:6] = torch.from_numpy(g["real_inputs"].astype(np.float32))
    return g, x


def _bench_input(dtype):
    # exactly bench.py's tensor: randn(seed = rank 0) cast to the leg's input dtype
    return torch.randn((256, 3, 160, 160), generator=torch.Generator().manual_seed(0)).to(dtype)


def test_bench_configuration_bs256_bf16_three_lanes_is_bitwise_the_serial_result():
    from vn_celeb_face_recognition_amd.models import InceptionResnetV1
    dev = torch.device("cuda:0")
    x = _bench_input(torch.bfloat16).to(dev)
    m = InceptionResnetV1(pretrained=None, device=dev, compute_dtype="bf16", max_batch=256).eval()
    m.set_streams(1)
    m.set_contexts(3)
    lanes = [torch.cuda.Stream(device=dev) for _ in range(3)]
    torch.cuda.synchronize()
    outs = []
    for i in range(7):                       # more steps than contexts: every context is re-used behind its event
        with torch.cuda.stream(lanes[i % 3]):
            outs.append(m(x))
    torch.cuda.synchronize()
    # the same 256 images, 8 at a time, on one stream and one context, from a second handle sized for 8
    small = InceptionResnetV1(pretrained=None, device=dev, compute_dtype="bf16", max_batch=8).eval()
    want = torch.cat([small(x[i:i + 8]) for i in range(0, 256, 8)])
    for o in outs:
        assert torch.equal(o, want)
    # bench.py's checksum of the timed output (bench.CHECKSUMS pins the same number once measured)
    import bench
    cs = float(outs[-1].double().abs().sum().item())
    pinned = bench.CHECKSUMS.get(("irv1", "bf16"))
    print("bench checksum (irv1, bf16): %.12f" % cs)
    if pinned is not None:
        assert abs(cs - pinned) <= 1e-9 * pinned


@pytest.mark.parametrize("dt", ["f32", "f16x2"])
def test_goldens_at_the_batch_boundaries_of_a_256_batch(dt):
    """The reference's own embeddings (irv1_seed0.npz) must come back <= 1e-4 from rows 0, 127, 128, 191, 192, 255 of
    a full 256-image batch on the gate-keeping paths, in the default (internally forked) mode and in the bench's
    3-context mode."""
    from vn_celeb_face_recognition_amd.models import InceptionResnetV1
    dev = torch.device("cuda:0")
    g, xg = _golden_inputs()
    x = _bench_input(torch.float32)
    for k, r in enumerate(ROWS):
        x[r] = xg[k]
    x = x.to(dev)
    m = InceptionResnetV1(pretrained=None, device=dev, compute_dtype=dt, max_batch=256).eval()
    y = m(x)
    err = np.linalg.norm(y[ROWS].cpu().numpy() - g["embeddings"], axis=1)
    assert err.max() <= 1e-4, (dt, err)
    m.set_streams(1)
    m.set_contexts(3)
    lanes = [torch.cuda.Stream(device=dev) for _ in range(3)]
    torch.cuda.synchronize()
    outs = []
    for i in range(4):
        with torch.cuda.stream(lanes[i % 3]):
            outs.append(m(x))
    torch.cuda.synchronize()
    for o in outs:
        assert torch.equal(o, y)


def test_ir100_rows_across_its_chunk_groups():
    """IR-100 runs its first two stages in sub-batches of 32 / 64 images (engine.cpp build_ir100 groups): a 70-image
    batch crosses both; bf16 must equal the 7-at-a-time result bit for bit and the reference's golden features must
    come back <= 1e-4 (relative) from rows 31, 32, 63, 64 on the fp32 path."""
    from vn_celeb_face_recognition_amd.models import iresnet100
    dev = torch.device("cuda:0")
    g = np.load(os.path.join(GOLDEN, "ir100_seed0.npz"))
    xg = seeded_normal((2, 3, 112, 112), g["input_seed"])
    x = seeded_normal((70, 3, 112, 112), 5)
    rows = [31, 32, 63, 64]
    for k, r in enumerate(rows):
        x[r] = xg[k % 2]
    xb = x.to(dev).to(torch.bfloat16)
    big = iresnet100(pretrained=False, compute_dtype="bf16", max_batch=70).to(dev).eval()
    small = iresnet100(pretrained=False, compute_dtype="bf16", max_batch=7).to(dev).eval()
    assert torch.equal(big(xb), torch.cat([small(xb[i:i + 7]) for i in range(0, 70, 7)]))
    m32 = iresnet100(pretrained=False, compute_dtype="f32", max_batch=70).to(dev).eval()
    y = m32(x.to(dev))[rows].cpu().numpy()
    want = g["features"][[0, 1, 0, 1]]
    err = np.linalg.norm(y - want, axis=1) / np.linalg.norm(want, axis=1)
    assert err.max() <= 1e-4, err
