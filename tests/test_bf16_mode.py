"""Throughput mode (bf16 tensors in HBM and in the MFMA operands; fp32 accumulate / softmax / statistics): the attention,
GEMM and direct-conv cores on bf16 inputs are checked against fp32 PyTorch evaluated on the same bf16-rounded values, with
bf16-level tolerances; whole-model deviation from the fp32 goldens is bounded on the GPU.  The parity mode (fp32 tensors)
is the one held to the 1e-3 bar elsewhere."""
import os

import numpy as np
import pytest
import torch

from backend import dev, use_hip  # noqa: F401
from cenet_amd import kern, ops


def _rel(a, b):
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-12)


BF = torch.bfloat16


@pytest.fixture
def bf16_mode():
    """kept for the tests' shape: the element type of the tensors selects the kernels, nothing is global any more"""
    yield lambda: None


def bft(t, dev):
    """bf16 leaf tensor on the device"""
    return t.to(dev).to(BF).requires_grad_(True)


def f32(t):
    """the fp32 values a bf16 tensor holds (reference input)"""
    return t.detach().float().cpu().clone().requires_grad_(True)


# the last case has >= 1024 queries over 49 keys: two query tiles per wave, query-sliced dK/dV with atomics
@pytest.mark.parametrize("B,N,Nk,C,heads", [(2, 70, 49, 128, 2), (1, 130, 130, 64, 1), (1, 1030, 49, 64, 1)])
def test_sr_attention_bf16(dev, bf16_mode, B, N, Nk, C, heads):
    g = torch.Generator().manual_seed(0)
    q = bft(torch.randn(B, N, C, generator=g), dev)
    kv = bft(torch.randn(B, Nk, 2 * C, generator=g), dev)
    go = torch.randn(B, N, C, generator=g).to(dev).to(BF)
    bf16_mode()
    o = ops.sr_attention(q, kv, heads)
    o.backward(go)
    hd = C // heads
    qr, kvr = f32(q), f32(kv)
    qh = qr.reshape(B, N, heads, hd).permute(0, 2, 1, 3)
    kk = kvr.reshape(B, Nk, 2, heads, hd).permute(2, 0, 3, 1, 4)
    ref = (torch.softmax(qh @ kk[0].transpose(-1, -2) * hd ** -0.5, -1) @ kk[1]).transpose(1, 2).reshape(B, N, C)
    ref.backward(go.float().cpu())
    assert _rel(o.detach().float().cpu(), ref.detach()) < 2e-2
    assert _rel(q.grad.float().cpu(), qr.grad) < 4e-2
    assert _rel(kv.grad.float().cpu(), kvr.grad) < 4e-2


@pytest.mark.parametrize("B,N,H,hd", [(2, 96, 2, 16), (1, 1030, 1, 16)])  # second: 128-query / 128-key workgroups
def test_diff_attention_heads_bf16(dev, bf16_mode, B, N, H, hd):
    E = 2 * H * hd
    g = torch.Generator().manual_seed(1)
    q, k, v = (bft(torch.randn(B, N, E, generator=g), dev) for _ in range(3))
    go = torch.randn(B, 2 * H, N, 2 * hd, generator=g).to(dev).to(BF)
    bf16_mode()
    U = ops.diff_attention_heads(q, k, v, H)
    U.backward(go)
    qr, kr, vr = (f32(t) for t in (q, k, v))
    qh = qr.view(B, N, 2 * H, hd).transpose(1, 2)
    kh = kr.view(B, N, 2 * H, hd).transpose(1, 2)
    vh = vr.view(B, N, H, 2 * hd).transpose(1, 2).repeat_interleave(2, dim=1)
    ref = torch.softmax(qh @ kh.transpose(-1, -2) * hd ** -0.5, -1) @ vh
    ref.backward(go.float().cpu())
    assert _rel(U.detach().float().cpu(), ref.detach()) < 2e-2
    for a, b in ((q, qr), (k, kr), (v, vr)):
        assert _rel(a.grad.float().cpu(), b.grad) < 4e-2


def test_nonlocal_attention_bf16(dev, bf16_mode):
    B, C, N = 2, 64, 100
    g = torch.Generator().manual_seed(2)
    th, ph, gx = (bft(torch.randn(B, C, N, generator=g), dev) for _ in range(3))
    go = torch.randn(B, C, N, generator=g).to(dev).to(BF)
    bf16_mode()
    y = ops.nonlocal_attention(th, ph, gx)
    y.backward(go)
    tr, pr, gr = (f32(t) for t in (th, ph, gx))
    a = torch.softmax(torch.einsum("nch,ncp->nhp", tr, pr) * C ** -0.5, dim=2)
    ref = torch.einsum("nhg,ncg->nch", a, gr)
    ref.backward(go.float().cpu())
    assert _rel(y.detach().float().cpu(), ref.detach()) < 2e-2
    for a_, b_ in ((th, tr), (ph, pr), (gx, gr)):
        assert _rel(a_.grad.float().cpu(), b_.grad) < 4e-2


@pytest.mark.gpu
@pytest.mark.parametrize("preset", ["acdc", "synapse", "skin"])
def test_whole_model_bf16_mode_stays_close_to_fp32_goldens(preset):
    """Documented deviation of the throughput mode on the three presets' reference-made goldens (random filled weights, eval):
    mean |dlogit| < 2 % of the logit range, < 1 % of mask pixels flip, Dice within 1e-3 of the reference (round 5: Synapse — 9
    classes, head dimensions 20 / 16 / 8 — and skin — 3-channel input, three FEA scales, head dimensions 160 / 64 / 32 — too)."""
    from test_model_parity import build
    from oracle import cenet_oracle as O
    d = use_hip()
    net, cfg, z, x, lab = build(preset, d)
    net.eval()
    kern.set_compute_bf16(True)
    try:
        with torch.no_grad():
            le = net(x)
        assert le.dtype == torch.bfloat16
        le = le.float().cpu()
    finally:
        kern.set_compute_bf16(False)
    ref = z["logits_eval_sub"]
    diff = np.abs(le[:, :, ::9, ::9].numpy() - ref)
    pred = O.predict(le)[:, ::5, ::5].numpy()
    flips = (pred != z["pred_eval_sub"]).mean()
    ddice = abs(O.mean_class_dice(le, lab.cpu(), cfg.num_classes) - float(z["dice_eval"]))
    print(f"[bf16 eval {preset}] mean|dlogit|/range {diff.mean() / np.abs(ref).max():.4f}  mask flips {flips:.4f}  dDice {ddice:.2e}")
    assert diff.mean() < 0.02 * np.abs(ref).max()
    assert flips < 1e-2
    assert ddice < 1e-3


@pytest.mark.parametrize("Cin,Cout,k,H,W", [(32, 32, 5, 16, 40), (64, 64, 3, 12, 36), (64, 32, 3, 9, 33), (32, 32, 5, 11, 70),
                                            (64, 64, 3, 12, 48), (64, 32, 3, 9, 72), (32, 32, 5, 19, 104)])
def test_direct_conv_bf16_fwd_and_dgrad(dev, bf16_mode, monkeypatch, Cin, Cout, k, H, W):
    """conv_direct.hip (LDS-halo direct convolution of the output head) vs fp32 torch conv, forward and data-gradient;
    ragged tiles (H, W not multiples of the 8x32 tile, odd W) included; the weight gradient of the three forward shapes
    is the direct kernel as well (LDS tiles + funnel-shifted tap windows)."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(3)
    B = 2
    x = bft(torch.randn(B, Cin, H, W, generator=g), dev)
    w = torch.nn.Parameter((torch.randn(Cout, Cin, k, k, generator=g) * 0.05).to(dev))
    go = torch.randn(B, Cout, H, W, generator=g).to(dev).to(BF)
    bf16_mode()
    assert kern.conv_direct_supported(Cin, Cout, k, 1, k // 2)
    if W % 8 == 0 and H > 12:  # (weight gradient, 16-byte staging: three workgroups, so that each walks several tiles and the
        monkeypatch.setenv("CENET_WGRAD_GRID", "3")  # next tile's register prefetch / late LDS write is exercised)
    y = ops.conv2d_nchw(x, w, None, stride=1, pad=k // 2)
    y.backward(go)
    xr = f32(x)
    wr = w.detach().cpu().clone().requires_grad_(True)
    ref = F.conv2d(xr, wr, None, padding=k // 2)
    ref.backward(go.float().cpu())
    assert _rel(y.detach().float().cpu(), ref.detach()) < 2e-2
    assert _rel(x.grad.float().cpu(), xr.grad) < 2e-2
    assert _rel(w.grad.cpu(), wr.grad) < 2e-2


# K >= 128 takes the K-step-64 GEMM instances: full and ragged last K tile, ragged M / N tiles, every operand orientation
# (forward: k-contiguous x k-contiguous; dX: k-contiguous x row-contiguous; dW: row-contiguous pair with split-K atomics)
@pytest.mark.parametrize("R,K,N", [(150, 192, 70), (70, 200, 130), (300, 128, 64), (33, 320, 320)])
def test_linear_bf16_k64(dev, bf16_mode, R, K, N):
    g = torch.Generator().manual_seed(R + K + N)
    x = bft(torch.randn(R, K, generator=g), dev)
    W = torch.nn.Parameter((torch.randn(N, K, generator=g) * 0.1).to(dev))
    b = torch.nn.Parameter(torch.randn(N, generator=g).to(dev))
    go = torch.randn(R, N, generator=g).to(dev).to(BF)
    bf16_mode()
    y = ops.linear(x, W, b)
    y.backward(go)
    xr, Wr, br = (t.detach().float().cpu().clone().requires_grad_(True) for t in (x, W, b))
    ref = torch.nn.functional.linear(xr, Wr, br)
    ref.backward(go.float().cpu())
    assert _rel(y.detach().float().cpu(), ref.detach()) < 2e-2
    assert _rel(x.grad.float().cpu(), xr.grad) < 2e-2
    assert _rel(W.grad.cpu(), Wr.grad) < 2e-2
    assert _rel(b.grad.cpu(), br.grad) < 1e-4


@pytest.mark.parametrize("B,Cin,Cout,HW", [(2, 136, 72, 100), (1, 256, 130, 49)])
def test_conv1x1_bf16_k64(dev, bf16_mode, B, Cin, Cout, HW):
    g = torch.Generator().manual_seed(B + Cin + Cout + HW)
    x = bft(torch.randn(B, Cin, HW, 1, generator=g), dev)
    W = torch.nn.Parameter((torch.randn(Cout, Cin, 1, 1, generator=g) * 0.1).to(dev))
    go = torch.randn(B, Cout, HW, 1, generator=g).to(dev).to(BF)
    bf16_mode()
    y = ops.conv1x1(x, W)
    y.backward(go)
    xr, Wr = (t.detach().float().cpu().clone().requires_grad_(True) for t in (x, W))
    ref = torch.nn.functional.conv2d(xr, Wr)
    ref.backward(go.float().cpu())
    assert _rel(y.detach().float().cpu(), ref.detach()) < 2e-2
    assert _rel(x.grad.float().cpu(), xr.grad) < 2e-2
    assert _rel(W.grad.cpu(), Wr.grad) < 2e-2


# spatial-reduction conv in throughput mode: K = C*s*s >= 1024 takes the split-K forward (atomics onto a bias-filled output)
@pytest.mark.parametrize("B,C,Cout,H,W,s", [(2, 64, 40, 8, 8, 4), (1, 16, 24, 16, 8, 8)])
def test_sr_conv_tok_bf16_split_k(dev, bf16_mode, B, C, Cout, H, W, s):
    g = torch.Generator().manual_seed(C + Cout + s)
    x = bft(torch.randn(B, H * W, C, generator=g), dev)
    Wt = torch.nn.Parameter((torch.randn(Cout, C, s, s, generator=g) * 0.05).to(dev))
    b = torch.nn.Parameter(torch.randn(Cout, generator=g).to(dev))
    Ho, Wo = H // s, W // s
    go = torch.randn(B, Ho * Wo, Cout, generator=g).to(dev).to(BF)
    bf16_mode()
    y = ops.conv2d_tok(x, H, W, Wt, b, stride=s, pad=0, out_layout="tok")
    y.backward(go)
    xr, Wr, br = (t.detach().float().cpu().clone().requires_grad_(True) for t in (x, Wt, b))
    ref = torch.nn.functional.conv2d(xr.transpose(1, 2).reshape(B, C, H, W), Wr, br, stride=s)
    ref = ref.reshape(B, Cout, Ho * Wo).transpose(1, 2)
    ref.backward(go.float().cpu())
    assert _rel(y.detach().float().cpu(), ref.detach()) < 2e-2
    assert _rel(x.grad.float().cpu(), xr.grad) < 2e-2
    assert _rel(Wt.grad.cpu(), Wr.grad) < 2e-2
    assert _rel(b.grad.cpu(), br.grad) < 1e-4


def _train_step_grads(build_fn, bf16):
    """one training step (forward, Dice + CE, backward; deterministic depth) -> loss, logits, flat gradient, arena, buffers"""
    import argparse
    from cenet_amd import losses, optim
    net, K, x, lab = build_fn()
    net.train()
    net.backbone.reset_drop_path(0.0)
    arena = optim.ParamArena(net, optim.cenet_segments())
    crit = losses.Criterion(K, argparse.Namespace(loss_type="dice,ce", loss_weights="0.5,0.5"))
    kern.set_compute_bf16(bf16)
    try:
        lt = net(x)
        assert lt.dtype == (torch.bfloat16 if bf16 else torch.float32)
        loss = crit(lt, lab)
        loss.backward()
    finally:
        kern.set_compute_bf16(False)
    torch.cuda.synchronize()
    return loss.item(), lt.detach().float().cpu(), arena.grads.clone(), arena, dict(net.named_buffers())


def _cos(a, b):
    return torch.nn.functional.cosine_similarity(a.double(), b.double(), dim=0).item()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["acdc", "synapse", "skin"])
def test_whole_model_bf16_training_step_vs_goldens(name):
    """The BENCHED mode (bf16 tensors end to end) on the three whole-model goldens of the unmodified reference, one training
    step, ONE evaluation.  Pinned against the goldens: loss within 5e-3, train logits mean |d| < 2 % of the logit range, BN
    running buffers; against the fp32 product step (itself pinned to the float64 goldens at 1e-3, test_model_parity.py):
    cosine of the FULL head+decoder gradient >= 0.90 (measured 0.984 - 0.996).
    These golden points (every parameter filled at random, batch 2) are ill-conditioned for the ENCODER gradient: an all-fp32
    step whose only perturbation is the network input rounded to bf16 values already moves the encoder segments to cosine
    0.988 - 0.997 (tools/bf16_bisect.py --quick), so no bound on the whole vector is asserted here.  The gradient of the
    benched mode is held to the reference itself on the well-conditioned golden: tests/test_wellcond.py (cosine >= 0.999 per
    arena segment against the reference's float64 gradient, plus run-to-run reproducibility)."""
    from oracle.gen_golden_keys import PROBE_BUFFERS
    from test_model_parity import build
    d = use_hip()

    def build_fn():
        net, cfg, z, x, lab = build(name, d)
        return net, cfg.num_classes, x, lab
    _, _, z, _, _ = build(name, d)
    l32, _, g32, arena, _ = _train_step_grads(build_fn, False)
    l16, lt, g16, _, bufs = _train_step_grads(build_fn, True)
    assert abs(l32 - float(z["loss"])) < 2e-4
    assert abs(l16 - float(z["loss"])) < 5e-3, (l16, float(z["loss"]))
    ref = z["logits_train_sub"]
    assert np.abs(lt[:, :, ::9, ::9].numpy() - ref).mean() < 0.02 * np.abs(ref).max()
    seg = {n: (s, e) for n, s, e in arena.segments}
    s, e = seg["head+decoder"]
    assert _cos(g32[s:e], g16[s:e]) >= 0.90, _cos(g32[s:e], g16[s:e])
    for k in PROBE_BUFFERS:
        np.testing.assert_allclose(bufs[k].reshape(-1)[:16].float().cpu().numpy(), z["b." + k], rtol=5e-2, atol=5e-3, err_msg=k)


@pytest.mark.gpu
def test_bf16_training_gradient_tracks_fp32_on_the_benched_model():
    """The benched workload's own model (default initialisation, ACDC preset, batch 4): the bf16-mode gradient against the
    fp32-mode gradient of the same step.  Whole vector and head+decoder segment: cosine >= 0.9995; encoder stages >= 0.985
    (measured 1.00000 / 1.00000 / 0.990 - 0.998); loss within 1e-3."""
    import bench
    d = use_hip()

    def build_fn():
        net = bench.make_model(d)
        x, lab = bench.synthetic(4, d, 7)
        return net, 4, x, lab
    l32, _, g32, arena, _ = _train_step_grads(build_fn, False)
    l16, _, g16, _, _ = _train_step_grads(build_fn, True)
    assert abs(l16 - l32) < 1e-3
    assert _cos(g32, g16) >= 0.9995
    for n, s, e in arena.segments:
        c = _cos(g32[s:e], g16[s:e])
        assert c >= (0.9995 if n == "head+decoder" else 0.985), (n, c)


@pytest.mark.gpu
def test_ham512_three_scale_preset_trains_in_both_modes():
    """SURVEY.md §8d C5 (skin.sh:93-94): 3x512x512 input, 2 classes, heads 2/2/2 (head dims 160 / 64 / 32 at 1024 / 4096 /
    16384 tokens), THREE FEA scales 1.0 / 0.75 / 0.5, batch 2.  No oracle exists at this size (parity unpinned); the two
    product modes must agree with each other: loss within 5e-3, head+decoder gradient cosine >= 0.995, logits at full size."""
    import bench
    d = use_hip()
    cfg = bench.CONFIGS["ham512"]

    def build_fn():
        net = bench.make_model(d, cfg)
        x, lab = bench.synthetic(2, d, 11, cfg)
        return net, cfg["classes"], x, lab
    l32, lt32, g32, arena, _ = _train_step_grads(build_fn, False)
    l16, lt16, g16, _, _ = _train_step_grads(build_fn, True)
    assert lt32.shape == (2, 2, 512, 512) and lt16.shape == lt32.shape
    assert np.isfinite(l32) and abs(l16 - l32) < 5e-3, (l32, l16)
    assert torch.isfinite(g32).all() and torch.isfinite(g16).all()
    seg = {n: (s, e) for n, s, e in arena.segments}
    s, e = seg["head+decoder"]
    assert _cos(g32[s:e], g16[s:e]) >= 0.995
    assert (lt16 - lt32).abs().mean() < 0.02 * lt32.abs().max()
