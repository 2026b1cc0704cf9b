"""The benched launch mode: a training step replayed from a hipGraph (cenet_amd.graph.GraphedStep — what `bench.py` picks on
one GPU) and the two-graph data-parallel form (GraphedSplitStep: forward + backward | eager all-reduce | SGD) must train like
eager launches of the same kernels: same losses and the same parameters after several steps, on the real CENet in bf16 mode
with the weight-gradient stream on.  Differences come only from the order of fp32 atomic additions."""
import argparse
import copy

import pytest
import torch

import bench
from cenet_amd import kern, losses, ops, optim
from cenet_amd.graph import GraphedSplitStep, GraphedStep, SegmentedStep

pytestmark = pytest.mark.gpu


def _setup(dev, seed=3):
    torch.manual_seed(seed)
    net = bench.make_model(dev)
    return net


def _train(net, dev, mode, steps=4, B=4):
    arena = optim.ParamArena(net, optim.cenet_segments())
    opt = optim.FusedSGD(arena, lr=0.01, momentum=0.9, weight_decay=1e-4)
    crit = losses.Criterion(4, argparse.Namespace(loss_type="dice,ce", loss_weights="0.5,0.5"))
    x, lab = bench.synthetic(B, dev, 7)
    was = torch.get_rng_state(), torch.cuda.get_rng_state()

    def fwd_bwd():
        opt.zero_grad()
        loss = crit(net(x), lab)
        loss.backward()
        return loss

    def body():
        loss = fwd_bwd()
        opt.step(sync_hyper=False)
        return loss

    out = []
    if mode == "eager":
        for _ in range(steps):
            opt.prepare()
            out.append(float(body().detach()))
    else:
        if mode == "graph":
            g = GraphedStep(body, optimizer=opt, warmup=1)
        elif mode == "split":
            g = GraphedSplitStep(fwd_bwd, opt, lambda: None, warmup=1)
        else:  # five backward graphs cut at the encoder stage outputs + the SGD graph (the N > 1 form that keeps the overlap)
            g = SegmentedStep(net, lambda: crit(net(x), lab), opt, lambda k: None, lambda: None, warmup=1)
        # warm-up + capture already trained 2 steps on this batch
        for _ in range(steps - 2):
            out.append(float(g().detach()))
    torch.cuda.synchronize()
    torch.set_rng_state(was[0])
    torch.cuda.set_rng_state(was[1])
    return out, arena.params.clone()


@pytest.mark.parametrize("mode", ["graph", "split", "segmented"])
def test_replay_trains_like_eager(mode):
    dev = torch.device("cuda:0")
    old_bf, old_ov = kern.set_compute_bf16(True), ops.set_wgrad_overlap(True)
    try:
        base = _setup(dev)
        for m in base.modules():  # DropPath draws host randomness per step: keep the two runs on the same masks
            if hasattr(m, "drop_prob"):
                m.drop_prob = 0.0
        ref, rep = base, copy.deepcopy(base)
        steps = 4
        le, pe = _train(ref, dev, "eager", steps)
        lg, pg = _train(rep, dev, mode, steps)
        assert all(torch.isfinite(torch.tensor(lg)))
        # the replayed steps are steps 3..4 of the same trajectory
        for a, b in zip(le[2:], lg):
            assert abs(a - b) <= 2e-2 * max(1.0, abs(a)), (le, lg)
        cos = torch.nn.functional.cosine_similarity((pe - pe.mean()).flatten(), (pg - pg.mean()).flatten(), dim=0).item()
        rel = ((pe - pg).norm() / pe.norm()).item()
        assert cos > 0.9999 and rel < 2e-3, (cos, rel)
    finally:
        kern.set_compute_bf16(old_bf)
        ops.set_wgrad_overlap(old_ov)


@pytest.mark.parametrize("mode", ["eager", "graph", "segmented"])
def test_branch_streams_train_alike(mode):
    """ops.set_branch_streams(True): the head's residual block on a second stream beside the decoder, its backward held to x4
    (ops.BranchGate) — off by default (measured slower under replay, ops.py) but kept correct: same trajectory as the one-stream
    step in eager launches, under one captured graph, and under the segmented captures of the N > 1 path."""
    dev = torch.device("cuda:0")
    old_bf = kern.set_compute_bf16(True)
    try:
        base = _setup(dev)
        for m in base.modules():
            if hasattr(m, "drop_prob"):
                m.drop_prob = 0.0
        ref, rep = base, copy.deepcopy(base)
        le, pe = _train(ref, dev, mode, 4)
        old = ops.set_branch_streams(True)
        try:
            lg, pg = _train(rep, dev, mode, 4)
        finally:
            ops.set_branch_streams(old)
        assert ops.branch_streams(dev)  # (the branch really ran on its own stream)
        for a, b in zip(le, lg):
            assert abs(a - b) <= 2e-2 * max(1.0, abs(a)), (le, lg)
        cos = torch.nn.functional.cosine_similarity((pe - pe.mean()).flatten(), (pg - pg.mean()).flatten(), dim=0).item()
        rel = ((pe - pg).norm() / pe.norm()).item()
        assert cos > 0.9999 and rel < 2e-3, (cos, rel)
    finally:
        kern.set_compute_bf16(old_bf)
