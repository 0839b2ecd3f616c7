"""GPU parity of the HIP encoder path (through the C ABI) against the oracle and the goldens."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, seeded_normal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def irv1_sd():
    from vn_celeb_face_recognition_amd.weights import generate_state_dict
    return generate_state_dict("irv1", 0, as_torch=True)


def _golden_inputs():
    g = np.load(os.path.join(GOLDEN, "irv1_seed0.npz"))
    x = seeded_normal((6, 3, 160, 160), g["input_seed"])
    x[4:6] = torch.from_numpy(g["real_inputs"].astype(np.float32))
    return g, x


def test_irv1_f32_matches_reference_golden_1e4():
    """fp32 MFMA path vs embeddings produced by the reference itself: L2 error <= 1e-4 (north_star)."""
    from vn_celeb_face_recognition_amd.models import InceptionResnetV1
    g, x = _golden_inputs()
    m = InceptionResnetV1(pretrained=None, device="cuda:0", compute_dtype="f32", max_batch=8).eval()
    y = m(x.cuda()).cpu().numpy()
    err = np.linalg.norm(y - g["embeddings"], axis=1)
    assert err.max() <= 1e-4, err
    assert np.allclose(np.linalg.norm(y, axis=1), 1.0, atol=1e-5)


def test_irv1_f32_stage_taps_match_oracle(irv1_sd):
    from vn_celeb_face_recognition_amd.models import InceptionResnetV1
    from oracle import irv1
    x = seeded_normal((3, 3, 160, 160), 99)
    taps = {}
    ref = irv1.irv1_forward(irv1_sd, x, taps=taps).numpy()
    m = InceptionResnetV1(pretrained=None, device="cuda:0", compute_dtype="f32", max_batch=4).eval()
    y = m(x.cuda()).cpu().numpy()
    for name in ["conv2d_1a", "conv2d_2a", "conv2d_2b", "maxpool_3a", "conv2d_3b", "conv2d_4a", "conv2d_4b",
                 "repeat_1", "mixed_6a", "repeat_2", "mixed_7a", "repeat_3", "block8"]:
        got = m.tap(name, 3)
        want = taps[name].numpy()
        assert got.shape == want.shape, name
        # fp32 with a different summation order: 1e-4 absolute on O(1..10) activations
        assert np.abs(got - want).max() <= 1e-4 * max(1.0, np.abs(want).max()), name
    assert np.linalg.norm(y - ref, axis=1).max() <= 1e-4


@pytest.mark.parametrize("dt,tol_emul,tol_f32", [("bf16", 2e-2, 6e-2), ("f16", 4e-3, 8e-3)])
def test_irv1_16bit_paths(irv1_sd, dt, tol_emul, tol_f32):
    """16-bit storage / fp32-accumulate MFMA path.  Two checks:
    (a) against the fp32 oracle: the bench dtype's accuracy, stated (L2 of unit embeddings);
    (b) against the oracle run with the SAME quantisation points (weights and every stored
        activation rounded to the 16-bit type): isolates kernel correctness from precision."""
    from vn_celeb_face_recognition_amd.models import InceptionResnetV1
    from oracle import irv1
    tdt = torch.bfloat16 if dt == "bf16" else torch.float16
    x = seeded_normal((4, 3, 160, 160), 7)
    ref32 = irv1.irv1_forward(irv1_sd, x).numpy()
    refq = irv1.irv1_forward_quantised(irv1_sd, x, tdt).numpy()
    m = InceptionResnetV1(pretrained=None, device="cuda:0", compute_dtype=dt, max_batch=4).eval()
    y = m(x.cuda()).cpu().numpy()
    e_emul = np.linalg.norm(y - refq, axis=1).max()
    e_f32 = np.linalg.norm(y - ref32, axis=1).max()
    print("irv1 %s: L2 vs quantised oracle %.3e, vs fp32 oracle %.3e" % (dt, e_emul, e_f32))
    assert e_emul <= tol_emul
    assert e_f32 <= tol_f32
    cos = (y * ref32).sum(axis=1)
    assert cos.min() >= 0.998


def test_irv1_batch_independence_and_ragged_batches():
    """eval-mode BN => per-image results do not depend on batch composition; odd batch sizes,
    batches larger than max_batch and the empty batch all work."""
    from vn_celeb_face_recognition_amd.models import InceptionResnetV1
    m = InceptionResnetV1(pretrained=None, device="cuda:0", compute_dtype="bf16", max_batch=5).eval()
    x = seeded_normal((13, 3, 160, 160), 3).cuda()
    y_all = m(x)
    y_one = torch.cat([m(x[i:i + 1]) for i in range(13)])
    assert torch.equal(y_all, y_one)
    assert m(x[:0]).shape == (0, 512)
    yb = m(x.to(torch.bfloat16))
    assert yb.shape == (13, 512)
    with pytest.raises(ValueError):
        m(torch.zeros(1, 3, 112, 112, device="cuda"))


def test_encoder_refuses_cpu():
    from vn_celeb_face_recognition_amd.models import InceptionResnetV1
    m = InceptionResnetV1(pretrained=None).eval()
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 160, 160))


def test_ir100_f32_matches_reference_golden():
    """IR-100 swap-in (config 5), fp32 path vs features produced by the reference itself.  Features
    are not normalised (|y| ~ 2.5): the gate is 1e-4 relative to the feature scale per row."""
    from vn_celeb_face_recognition_amd.models import iresnet100
    g = np.load(os.path.join(GOLDEN, "ir100_seed0.npz"))
    x = seeded_normal((2, 3, 112, 112), g["input_seed"])
    m = iresnet100(pretrained=False, compute_dtype="f32", max_batch=2).to("cuda:0").eval()
    y = m(x.cuda()).cpu().numpy()
    err = np.linalg.norm(y - g["features"], axis=1) / np.linalg.norm(g["features"], axis=1)
    assert err.max() <= 1e-4, err


def test_ir100_stage_taps_and_bf16(irv1_sd):
    from vn_celeb_face_recognition_amd.models import iresnet100
    from vn_celeb_face_recognition_amd.weights import generate_state_dict
    from oracle import iresnet
    sd = generate_state_dict("iresnet100", 0, as_torch=True)
    x = seeded_normal((3, 3, 112, 112), 21)
    taps = {}
    ref = iresnet.iresnet_forward(sd, x, taps=taps).numpy()
    m = iresnet100(pretrained=False, compute_dtype="f32", max_batch=3).to("cuda:0").eval()
    y = m(x.cuda()).cpu().numpy()
    for name in ["stem", "layer1", "layer2", "layer3", "layer4"]:
        got, want = m.tap(name, 3), taps[name].numpy()
        assert got.shape == want.shape
        assert np.abs(got - want).max() <= 2e-4 * max(1.0, np.abs(want).max()), name
    assert (np.linalg.norm(y - ref, axis=1) / np.linalg.norm(ref, axis=1)).max() <= 1e-4
    mb = iresnet100(pretrained=False, compute_dtype="bf16", max_batch=3).to("cuda:0").eval()
    yb = mb(x.cuda()).cpu().numpy()
    rel = np.linalg.norm(yb - ref, axis=1) / np.linalg.norm(ref, axis=1)
    print("ir100 bf16 relative L2 error vs fp32 oracle:", rel)
    assert rel.max() <= 5e-2
    cos = (yb * ref).sum(axis=1) / np.linalg.norm(yb, axis=1) / np.linalg.norm(ref, axis=1)
    assert cos.min() >= 0.998


def test_activation_contexts_let_calls_on_different_streams_overlap_without_changing_results():
    """vnf_encoder_set_contexts: consecutive calls rotate over private activation-buffer sets, so batches issued on
    different streams run concurrently; every embedding must equal the one-stream, one-context result bit for bit."""
    import torch
    from vn_celeb_face_recognition_amd.models import InceptionResnetV1
    dev = torch.device("cuda:0")
    m = InceptionResnetV1(pretrained=None, device=dev, compute_dtype="bf16", max_batch=24).eval()
    xs = [torch.randn((24 - 5 * i, 3, 160, 160), generator=torch.Generator().manual_seed(10 + i)).to(dev).to(torch.bfloat16)
          for i in range(3)]
    want = [m(x).cpu().numpy() for x in xs]
    m.set_streams(1)
    m.set_contexts(2)
    lanes = [torch.cuda.Stream(device=dev) for _ in range(2)]
    torch.cuda.synchronize()
    outs = []
    for rep in range(4):
        for i, x in enumerate(xs):
            with torch.cuda.stream(lanes[(rep * 3 + i) % 2]):
                outs.append((i, m(x)))
    torch.cuda.synchronize()
    for i, o in outs:
        assert np.array_equal(o.cpu().numpy(), want[i])
    m.set_contexts(1)
    assert np.array_equal(m(xs[0]).cpu().numpy(), want[0])
    # the same through the host-side helper (host tensors in, rotating lanes inside)
    got = {}
    for i, emb, ready in m.embed_stream([x.cpu() for x in xs] * 2, lanes=3):
        ready.synchronize()
        got[i] = emb.cpu().numpy()
    assert sorted(got) == list(range(6)) and all(np.array_equal(got[i], want[i % 3]) for i in got)


def test_irv1_split_f16_meets_the_1e4_gate_on_16bit_mfma(irv1_sd, monkeypatch):
    """compute_dtype="f16x2": every weight / activation is an (hi, lo) pair of halves and each product is expanded on
    v_mfma_f32_16x16x32_f16 -- a 16-bit-operand MFMA path that must sit inside the north-star gate (<= 1e-4 L2 against the
    reference's own embeddings), with every stage tap within 1e-4 of the fp32 oracle (tolerance relative to the tap's
    scale, as for the exact-f32 path)."""
    from vn_celeb_face_recognition_amd.models import InceptionResnetV1
    from oracle import irv1
    g, x = _golden_inputs()
    m = InceptionResnetV1(pretrained=None, device="cuda:0", compute_dtype="f16x2", max_batch=8).eval()
    y = m(x.cuda()).cpu().numpy()
    err = np.linalg.norm(y - g["embeddings"], axis=1)
    print("irv1 f16x2: L2 error vs reference golden", err)
    assert err.max() <= 1e-4, err
    x3 = seeded_normal((3, 3, 160, 160), 99)
    taps = {}
    ref = irv1.irv1_forward(irv1_sd, x3, taps=taps).numpy()
    y3 = m(x3.cuda()).cpu().numpy()
    # every stage tap on the per-convolution plan (VNF_FUSE=0: with the fused stem, conv2d_2a / 2b / maxpool_3a only
    # ever exist in LDS); the taps the fused kernels do produce are checked on the default plan as well
    monkeypatch.setenv("VNF_FUSE", "0")
    plan = InceptionResnetV1(pretrained=None, device="cuda:0", compute_dtype="f16x2", max_batch=8).eval()
    assert np.linalg.norm(plan(x3.cuda()).cpu().numpy() - ref, axis=1).max() <= 1e-4
    monkeypatch.delenv("VNF_FUSE")
    for name in ["conv2d_1a", "conv2d_3b", "conv2d_4b", "repeat_1", "repeat_2", "block8"]:
        got, want = m.tap(name, 3), taps[name].numpy()
        assert np.abs(got - want).max() / max(1.0, np.abs(want).max()) <= 1e-4, ("fused plan", name)
    table = []
    for name in ["conv2d_1a", "conv2d_2a", "conv2d_2b", "maxpool_3a", "conv2d_3b", "conv2d_4a", "conv2d_4b",
                 "repeat_1", "mixed_6a", "repeat_2", "mixed_7a", "repeat_3", "block8"]:
        got, want = plan.tap(name, 3), taps[name].numpy()
        assert got.shape == want.shape, name
        e = np.abs(got - want).max() / max(1.0, np.abs(want).max())
        table.append((name, e))
        assert e <= 1e-4, (name, e)
    print("irv1 f16x2 per-stage max error / scale:", ", ".join("%s %.1e" % t for t in table))
    assert np.linalg.norm(y3 - ref, axis=1).max() <= 1e-4
    # 16-bit inputs are accepted too (they are exactly representable in the split type)
    yb = m(x.cuda().to(torch.float16)).cpu().numpy()
    assert np.isfinite(yb).all()


def test_ir100_split_f16_matches_reference_golden():
    from vn_celeb_face_recognition_amd.models import iresnet100
    g = np.load(os.path.join(GOLDEN, "ir100_seed0.npz"))
    x = seeded_normal((2, 3, 112, 112), g["input_seed"])
    m = iresnet100(pretrained=False, compute_dtype="f16x2", max_batch=2).to("cuda:0").eval()
    y = m(x.cuda()).cpu().numpy()
    err = np.linalg.norm(y - g["features"], axis=1) / np.linalg.norm(g["features"], axis=1)
    assert err.max() <= 1e-4, err


@pytest.mark.parametrize("dt,tol", [("bf16", 2.5e-2), ("f16", 3e-3), ("f16x2", 2e-5)])
def test_persistent_block17_trunk_kernel_matches_the_unfused_plan_and_the_oracle(irv1_sd, monkeypatch, dt, tol):
    """repeat_2 (10 x Block17, inception_resnet_v1.py:70-95) runs as ONE persistent kernel on the 16-bit paths
    (trunk17.hip; trunk17s.hip for the planar split-f16 dtype: residual trunk in fp32 registers, intermediates in LDS).  Against the fp32 oracle's stage taps it
    must be at least as close as the unfused per-convolution plan (VNF_FUSE=0), which rounds the trunk to 16 bits after
    every block; the two plans must agree with each other to the storage precision."""
    from vn_celeb_face_recognition_amd.models import InceptionResnetV1
    from oracle import irv1
    x = seeded_normal((5, 3, 160, 160), 31)
    taps = {}
    ref = irv1.irv1_forward(irv1_sd, x, taps=taps).numpy()
    if dt == "f16x2":
        monkeypatch.setenv("VNF_FUSE", "1")   # the trunk kernel alone: the split-f16 stem / Block35 kernels are not bitwise the plan
    fused = InceptionResnetV1(pretrained=None, device="cuda:0", compute_dtype=dt, max_batch=5).eval()
    yf = fused(x.cuda()).cpu().numpy()
    tf = {n: fused.tap(n, 5) for n in ("mixed_6a", "repeat_2", "mixed_7a")}
    monkeypatch.setenv("VNF_FUSE", "0")
    plain = InceptionResnetV1(pretrained=None, device="cuda:0", compute_dtype=dt, max_batch=5).eval()
    yp = plain(x.cuda()).cpu().numpy()
    tp = {n: plain.tap(n, 5) for n in ("mixed_6a", "repeat_2", "mixed_7a")}
    assert np.array_equal(tf["mixed_6a"], tp["mixed_6a"])          # same input to the stack (fused Block35 is bit-exact)
    want = taps["repeat_2"].numpy()
    scale = np.abs(want).max()
    ef, ep = np.abs(tf["repeat_2"] - want).max() / scale, np.abs(tp["repeat_2"] - want).max() / scale
    rf = np.linalg.norm(tf["repeat_2"] - want) / np.linalg.norm(want)
    rp = np.linalg.norm(tp["repeat_2"] - want) / np.linalg.norm(want)
    print("%s repeat_2 tap vs fp32 oracle: fused max %.3e rel-L2 %.3e | unfused max %.3e rel-L2 %.3e" % (dt, ef, rf, ep, rp))
    assert ef <= tol and rf <= tol
    assert rf <= max(rp * 1.05, 3e-6)                               # fp32 trunk: not worse than the 16-bit trunk (f16x2: both at fp32 noise)
    assert np.abs(tf["repeat_2"] - tp["repeat_2"]).max() / scale <= 2 * tol
    e_f, e_p = np.linalg.norm(yf - ref, axis=1).max(), np.linalg.norm(yp - ref, axis=1).max()
    print("%s embedding L2 vs fp32 oracle: fused %.3e unfused %.3e" % (dt, e_f, e_p))
    assert e_f <= (1e-4 if dt == "f16x2" else max(6e-2 if dt == "bf16" else 8e-3, e_p * 1.2))


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("n", [1, 7])
@pytest.mark.parametrize("fuse", ["2", "18", "50"])
def test_fused_block35_is_bitwise_the_five_launch_plan(monkeypatch, dt, n, fuse):
    """repeat_1 (5 x Block35, inception_resnet_v1.py:36-67): one fused launch per block (block35.hip: all intermediates in
    LDS, pixel tiles split over the waves, weights in MFMA fragment order) -- or, VNF_FUSE bit 4, ONE launch for the five
    blocks with the residual stream in registers (trunk35.hip: x enters the reduce GEMM and the branch outputs enter the
    up projection through lane-row swaps, never through memory) -- keeps the unfused plan's rounding points and summation
    order, so conv2d_4b -> repeat_1 must come out bit for bit the same -- image borders (3x3 taps), the padding pixels
    of the 19th tile and every batch position included."""
    from vn_celeb_face_recognition_amd.models import InceptionResnetV1
    x = seeded_normal((n, 3, 160, 160), 57 + n).cuda()
    monkeypatch.setenv("VNF_FUSE", fuse)     # Block35 only: per block (2) / the stack in one launch (18) / + mixed_6a.branch1.0 (50)
    fused = InceptionResnetV1(pretrained=None, device="cuda:0", compute_dtype=dt, max_batch=n).eval()
    yf = fused(x)
    a4, r1 = fused.tap("conv2d_4b", n), fused.tap("repeat_1", n)
    monkeypatch.setenv("VNF_FUSE", "0")
    plain = InceptionResnetV1(pretrained=None, device="cuda:0", compute_dtype=dt, max_batch=n).eval()
    yp = plain(x)
    assert np.array_equal(a4, plain.tap("conv2d_4b", n))
    want = plain.tap("repeat_1", n)
    assert np.isfinite(r1).all()
    bad = np.argwhere(r1 != want)
    assert len(bad) == 0, (len(bad), bad[:8], r1[tuple(bad[0])], want[tuple(bad[0])])
    # mixed_6a (whose branch1.0 the stack kernel computes from its registers under bit 5) and the embeddings
    assert np.array_equal(fused.tap("mixed_6a", n), plain.tap("mixed_6a", n))
    assert torch.equal(yf, yp)


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("n", [1, 3, 130])
def test_fused_stem_2a_2b_maxpool_is_bitwise_the_three_launch_plan(monkeypatch, dt, n):
    """conv2d_2a -> conv2d_2b -> maxpool_3a (inception_resnet_v1.py:282-285) as one rolling-row launch (stem_mid.hip)
    keeps the unfused plan's rounding points and summation order: the pooled map and everything after it must be bit
    for bit the three-launch plan's -- first / last rows and columns (2b's zero padding, the valid-conv edge of 2a),
    the 128-image sub-batch boundary of the stem group (n = 130) included."""
    from vn_celeb_face_recognition_amd.models import InceptionResnetV1
    x = seeded_normal((n, 3, 160, 160), 91 + n).cuda()
    monkeypatch.setenv("VNF_FUSE", "4")      # the stem kernel only
    fused = InceptionResnetV1(pretrained=None, device="cuda:0", compute_dtype=dt, max_batch=n).eval()
    yf = fused(x)
    k = min(n, 4)
    pf = fused.tap("maxpool_3a", n)
    monkeypatch.setenv("VNF_FUSE", "0")
    plain = InceptionResnetV1(pretrained=None, device="cuda:0", compute_dtype=dt, max_batch=n).eval()
    yp = plain(x)
    want = plain.tap("maxpool_3a", n)
    assert np.isfinite(pf).all()
    bad = np.argwhere(pf != want)
    assert len(bad) == 0, (len(bad), bad[:8], pf[tuple(bad[0])], want[tuple(bad[0])])
    assert np.array_equal(fused.tap("conv2d_4b", k), plain.tap("conv2d_4b", k))
    assert torch.equal(yf, yp)


@pytest.mark.parametrize("dt", ["bf16", "f16"])
def test_conv2d_3b_inside_the_stem_kernel_is_bitwise_the_plan(dt, monkeypatch):
    """VNF_FUSE bit 3: conv2d_3b (1x1, 64 -> 80, inception_resnet_v1.py:286) applied to every pooled row inside
    stem_mid_kernel (the pooled tensor never reaches memory) -- same MFMA k order, bias after the sum, 16-bit rounding
    of the pooled row and of the output as the plan's convolution: bit-identical conv2d_3b output and embeddings."""
    from vn_celeb_face_recognition_amd.models import InceptionResnetV1
    n = 19
    x = torch.randn((n, 3, 160, 160), generator=torch.Generator().manual_seed(7)).cuda()
    monkeypatch.setenv("VNF_FUSE", "12")     # stem kernel + conv2d_3b inside it
    fused = InceptionResnetV1(pretrained=None, device="cuda:0", compute_dtype=dt, max_batch=n).eval()
    yf = fused(x).cpu().numpy()
    tf_ = fused.tap("conv2d_3b", n)
    monkeypatch.setenv("VNF_FUSE", "4")      # stem kernel, conv2d_3b as a plan convolution
    mid = InceptionResnetV1(pretrained=None, device="cuda:0", compute_dtype=dt, max_batch=n).eval()
    ym = mid(x).cpu().numpy()
    tm = mid.tap("conv2d_3b", n)
    monkeypatch.setenv("VNF_FUSE", "0")
    plain = InceptionResnetV1(pretrained=None, device="cuda:0", compute_dtype=dt, max_batch=n).eval()
    yp = plain(x).cpu().numpy()
    assert tf_.shape == tm.shape == (n, 80, 38, 38)
    assert np.array_equal(tf_, tm)
    assert np.array_equal(yf, ym) and np.array_equal(yf, yp)


def test_taps_inside_an_active_fused_kernel_are_refused(monkeypatch):
    """With the fused stem (bf16 / f16, default VNF_FUSE) conv2d_2a, conv2d_2b and maxpool_3a only ever exist in LDS:
    vnf_encoder_tap must say so instead of serving a never-written buffer; taps the fused kernels DO produce work."""
    from vn_celeb_face_recognition_amd import _lib
    from vn_celeb_face_recognition_amd.models import InceptionResnetV1
    monkeypatch.delenv("VNF_FUSE", raising=False)
    m = InceptionResnetV1(pretrained=None, device="cuda:0", compute_dtype="bf16", max_batch=2).eval()
    m(seeded_normal((2, 3, 160, 160), 5).cuda())
    for name in ("conv2d_2a", "conv2d_2b", "maxpool_3a"):
        with pytest.raises(_lib.VnfError, match="fused kernel"):
            m.tap(name, 2)
    for name, c in (("conv2d_1a", 32), ("conv2d_3b", 80), ("repeat_1", 256), ("repeat_2", 896)):
        t = m.tap(name, 2)
        assert t.shape[:2] == (2, c) and np.isfinite(t).all() and np.abs(t).max() > 0


@pytest.mark.parametrize("n", [1, 7])
def test_fused_block35_split_f16_matches_the_five_launch_plan_and_the_oracle(irv1_sd, monkeypatch, n):
    """block35s.hip: one launch per Block35 in the planar split-f16 dtype (branch outputs stay in registers as MFMA B
    fragments in a permuted k order, reduce weights ride the x ring).  Same products as the plan's five convolutions but
    a different summation order inside the up convolution's 32-deep steps: fp32-noise agreement with the unfused plan,
    and the 1e-4 bars against the oracle."""
    from vn_celeb_face_recognition_amd.models import InceptionResnetV1
    from oracle import irv1
    x = seeded_normal((n, 3, 160, 160), 71 + n)
    taps = {}
    ref = irv1.irv1_forward(irv1_sd, x, taps=taps).numpy()
    monkeypatch.setenv("VNF_FUSE", "2")      # Block35 only
    fused = InceptionResnetV1(pretrained=None, device="cuda:0", compute_dtype="f16x2", max_batch=n).eval()
    yf = fused(x.cuda()).cpu().numpy()
    a4, r1 = fused.tap("conv2d_4b", n), fused.tap("repeat_1", n)
    monkeypatch.setenv("VNF_FUSE", "0")
    plain = InceptionResnetV1(pretrained=None, device="cuda:0", compute_dtype="f16x2", max_batch=n).eval()
    yp = plain(x.cuda()).cpu().numpy()
    assert np.array_equal(a4, plain.tap("conv2d_4b", n))           # same input to the blocks
    want_p, want_o = plain.tap("repeat_1", n), taps["repeat_1"].numpy()
    scale = np.abs(want_o).max()
    assert np.isfinite(r1).all()
    e_plan, e_orc = np.abs(r1 - want_p).max() / scale, np.abs(r1 - want_o).max() / scale
    print("block35s repeat_1: vs unfused plan %.2e, vs oracle %.2e (unfused vs oracle %.2e)" % (e_plan, e_orc, np.abs(want_p - want_o).max() / scale))
    assert e_plan <= 2e-6 and e_orc <= 1e-5
    assert np.linalg.norm(yf - ref, axis=1).max() <= 1e-4 and np.linalg.norm(yf - yp, axis=1).max() <= 5e-6


@pytest.mark.parametrize("n", [1, 3, 130])
def test_fused_stem_split_f16_matches_the_plan_and_the_oracle(irv1_sd, monkeypatch, n):
    """stem_mids.hip: conv2d_2a -> conv2d_2b -> maxpool_3a -> conv2d_3b as one rolling-row launch in the planar split-f16
    dtype (vertical pooling maximum in registers, 2b's zero rows skipped).  Same products and the plan's (kh, kw, c)
    order; the pooled values are split after the maximum exactly as the plan's pool does: conv2d_3b must agree with the
    unfused plan to fp32 noise -- first / last rows and columns and the 128-image sub-batch boundary (n = 130) included
    -- and with the oracle to the 1e-4 bar."""
    from vn_celeb_face_recognition_amd.models import InceptionResnetV1
    from oracle import irv1
    x = seeded_normal((n, 3, 160, 160), 191 + n)
    monkeypatch.setenv("VNF_FUSE", "12")     # the stem kernel with conv2d_3b inside it
    fused = InceptionResnetV1(pretrained=None, device="cuda:0", compute_dtype="f16x2", max_batch=n).eval()
    yf = fused(x.cuda()).cpu().numpy()
    tf_ = fused.tap("conv2d_3b", n)
    monkeypatch.setenv("VNF_FUSE", "0")
    plain = InceptionResnetV1(pretrained=None, device="cuda:0", compute_dtype="f16x2", max_batch=n).eval()
    yp = plain(x.cuda()).cpu().numpy()
    tp = plain.tap("conv2d_3b", n)
    assert tf_.shape == tp.shape == (n, 80, 38, 38) and np.isfinite(tf_).all()
    scale = np.abs(tp).max()
    bad = np.argwhere(np.abs(tf_ - tp) > 2e-6 * scale)
    assert len(bad) == 0, (len(bad), bad[:8], tf_[tuple(bad[0])], tp[tuple(bad[0])])
    assert np.linalg.norm(yf - yp, axis=1).max() <= 5e-6
    k = min(n, 3)
    taps = {}
    ref = irv1.irv1_forward(irv1_sd, x[:k], taps=taps).numpy()
    assert np.abs(tf_[:k] - taps["conv2d_3b"].numpy()).max() / scale <= 1e-5
    assert np.linalg.norm(yf[:k] - ref, axis=1).max() <= 1e-4


@pytest.mark.parametrize("arch,dt", [("irv1", "bf16"), ("irv1", "f16"), ("irv1", "f16x2"), ("ir100", "bf16")])
def test_persistent_conv_kernel_is_bitwise_the_one_tile_kernel(monkeypatch, arch, dt):
    """conv_ws.hip's persistent form (a workgroup walks several output tiles, loaders run the ring across tile
    boundaries, accumulators leave through a register epilogue: lane-row swaps -> 16-byte stores, residual chunks
    fetched at tile start, PReLU slopes beside the bias) against one tile per workgroup with the LDS-staged epilogue
    (VNF_WS_PERSIST=0), on the per-convolution plan (VNF_FUSE=0) so every layer shape of the network goes through a
    convolution kernel: same K order, sums and roundings, so the embeddings must be bit for bit the same -- ragged last
    tiles, tiles that cross image boundaries and the residual / PReLU layers (Block8 / Block17 / Block35 up
    projections, IR-100 units) included."""
    from vn_celeb_face_recognition_amd.models import InceptionResnetV1, iresnet100
    n, size = (131, 160) if arch == "irv1" else (9, 112)
    x = seeded_normal((n, 3, size, size), 91).cuda()
    monkeypatch.setenv("VNF_FUSE", "0")

    def build():
        if arch == "irv1":
            return InceptionResnetV1(pretrained=None, device="cuda:0", compute_dtype=dt, max_batch=n).eval()
        return iresnet100(pretrained=False, compute_dtype=dt, max_batch=n).to("cuda:0").eval()

    ya = build()(x)                              # autotuned with the persistent form available
    monkeypatch.setenv("VNF_WS_PERSIST", "0")
    yb = build()(x)                              # autotuned again: other tile choices, same arithmetic
    assert torch.isfinite(ya).all()
    assert torch.equal(ya, yb)


@pytest.mark.parametrize("dt", ["bf16", "f16", "f16x2"])
def test_conv2d_1a_on_the_f32_mfma_is_bitwise_the_valu_kernel(monkeypatch, dt):
    """conv2d_1a (inception_resnet_v1.py:281, fp32 weights on the caller's NCHW tensor): the v_mfma_f32_16x16x4_f32
    kernel (weights as the A operand, k = (c, kh, kw) in MFMA steps of four) against the packed-FMA VALU kernel
    (VNF_STEM1A_MFMA=0): the f32 MFMA is an exact fmaf chain in k order, so the conv2d_1a tap and everything behind it
    must be bit for bit the same -- odd batch, so the last 16-pixel group is ragged."""
    from vn_celeb_face_recognition_amd.models import InceptionResnetV1
    n = 5
    x = seeded_normal((n, 3, 160, 160), 17).cuda()
    monkeypatch.setenv("VNF_FUSE", "0")          # conv2d_1a readable as a tap, per-convolution plan behind it
    m = InceptionResnetV1(pretrained=None, device="cuda:0", compute_dtype=dt, max_batch=n).eval()
    ya = m(x)
    ta = m.tap("conv2d_1a", n)
    monkeypatch.setenv("VNF_STEM1A_MFMA", "0")
    yb = m(x)
    tb = m.tap("conv2d_1a", n)
    assert np.isfinite(ta).all() and np.abs(ta).max() > 0
    assert np.array_equal(ta, tb)
    assert torch.equal(ya, yb)
