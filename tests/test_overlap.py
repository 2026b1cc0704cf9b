"""The weight-gradient stream (ops.set_wgrad_overlap) must not change results: same batch, same weights, gradients and the
updated parameters with the overlap on == off (up to the order of float atomics), including through the fused optimizer."""
import argparse

import pytest
import torch

from backend import use_hip
from oracle.golden_cases import MODEL_CONFIGS


@pytest.mark.gpu
@pytest.mark.parametrize("bf16", [False, True])
def test_wgrad_overlap_is_transparent(bf16):
    from cenet_amd import kern, losses, ops, optim
    from cenet_amd.networks import CENet
    dev = use_hip()
    kw = MODEL_CONFIGS["acdc"]["kw"]
    crit = losses.Criterion(4, argparse.Namespace(loss_type="dice,ce,boundary", loss_weights="0.4,0.3,0.3"))
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 1, 224, 224, generator=g).to(dev)
    lab = torch.randint(0, 4, (2, 224, 224), generator=g).float().to(dev)
    out = {}
    kern.set_compute_bf16(bf16)
    try:
        for tag, overlap in (("off", False), ("off2", False), ("on", True)):
            ops.set_wgrad_overlap(overlap)
            torch.manual_seed(5)
            net = CENet(**kw).to(dev).train()
            net.backbone.reset_drop_path(0.0)
            arena = optim.ParamArena(net, optim.cenet_segments())
            opt = optim.FusedSGD(arena, lr=0.05, momentum=0.9, weight_decay=1e-4)
            # ONE step: with batch 2 the BatchNorm statistics amplify the float-atomic ordering noise of a first step into
            # ~1e-5 differences in the second even between two identical runs, which would hide a real ordering bug
            opt.zero_grad()
            loss = crit(net(x), lab)
            loss.backward()
            opt.step()
            torch.cuda.synchronize()
            out[tag] = (loss.item(), arena.grads.clone(), arena.params.clone())
    finally:
        ops.set_wgrad_overlap(False)
        kern.set_compute_bf16(False)
    (l0, g0, p0), (l2, g2, p2), (l1, g1, p1) = out["off"], out["off2"], out["on"]
    # self-calibrating: two identical runs differ by the order of float atomics (weight gradients; in bf16 mode also the
    # split-K forward convs, amplified by batch-2 BatchNorm); the overlap may not add to that
    scale = g0.abs().max().item()
    base_g = (g0 - g2).abs().max().item()
    base_p = (p0 - p2).abs().max().item()
    assert abs(l0 - l1) <= 30 * abs(l0 - l2) + 1e-5
    assert (g0 - g1).abs().max().item() <= 30 * base_g + 1e-6 * scale
    assert (p0 - p1).abs().max().item() <= 30 * base_p + 1e-6


@pytest.mark.gpu
def test_wgrad_overlap_with_backlogged_side_stream():
    """Conv1x1Fn / LinearFn hand the SAME gradient tensor to the weight-gradient stream and back to autograd as
    the residual gradient.  Where the residual source has a second consumer (MCA's `shortcut`, DSEB's `skip`) autograd adds
    into it — in place if nobody else holds it — while the side stream may still be reading.  The side stream is stalled
    here (a long sleep ahead of every weight-gradient kernel), so any such write-before-read changes proj_2 / mixer dW."""
    from cenet_amd import losses, ops, optim
    from cenet_amd.networks import CENet
    dev = use_hip()
    kw = MODEL_CONFIGS["acdc"]["kw"]
    crit = losses.Criterion(4, argparse.Namespace(loss_type="dice,ce", loss_weights="0.5,0.5"))
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 1, 224, 224, generator=g).to(dev)
    lab = torch.randint(0, 4, (2, 224, 224), generator=g).float().to(dev)
    out = {}
    try:
        for tag, overlap in (("off", False), ("off2", False), ("stalled", True)):
            ops.set_wgrad_overlap(overlap)
            torch.manual_seed(5)
            net = CENet(**kw).to(dev).train()
            net.backbone.reset_drop_path(0.0)
            arena = optim.ParamArena(net, optim.cenet_segments())
            opt = optim.FusedSGD(arena, lr=0.05)
            opt.zero_grad()
            loss = crit(net(x), lab)
            if overlap:
                with ops._wgrad_side():  # creates the stream; then park ~50 ms of sleep on it ahead of the backward
                    torch.cuda._sleep(int(1e8))
            loss.backward()
            ops.wgrad_join()
            torch.cuda.synchronize()
            out[tag] = arena.grads.clone()
    finally:
        ops.set_wgrad_overlap(False)
    scale = out["off"].abs().max().item()
    base = (out["off"] - out["off2"]).abs().max().item()
    assert (out["off"] - out["stalled"]).abs().max().item() <= 30 * base + 1e-6 * scale
