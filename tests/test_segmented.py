"""Segmented backward (cenet_amd.graph: enable_segment_cuts / backward_pieces / SegmentedStep): the backward pass cut at the four
encoder stage outputs into five autograd runs, one per gradient-arena segment, so that under hipGraph replay each segment's
all-reduce can start between two graphs and overlap the rest of the backward pass (bench.py at N > 1).

* single process: the five pieces leave exactly the gradient of one whole backward pass, and after piece k the arena slice of
  segment k is FINAL (`sim` = host SIMT checker on CPU, fp32 and bf16 storage; `hip` on the GPU);
* two gloo ranks on CPU: a SegmentedStep driven by GradReducer.segment_ready / finish leaves both ranks with identical
  parameters, equal to those of the hook-driven eager step on the same shards (mean of the per-shard gradients);
* GPU: the five-graph replay trains like eager launches (test_graph_replay.py, mode "segmented")."""
import argparse
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from backend import dev, use_sim  # noqa: F401
from test_parallel_gloo import _cenet_shard, _free_port


def _net(seed=7):
    from cenet_amd.networks import CENet
    from oracle import cenet_oracle as O
    from oracle.golden_cases import MODEL_CONFIGS, config_from_kwargs
    kw = MODEL_CONFIGS["acdc"]["kw"]
    net = CENet(**kw)
    net.load_state_dict(O.make_state_dict(config_from_kwargs(kw), seed=seed), strict=True)
    net.train()
    net.backbone.reset_drop_path(0.0)
    return net


def _crit():
    from cenet_amd import losses
    return losses.Criterion(4, argparse.Namespace(loss_type="dice,ce", loss_weights="0.5,0.5"))


def _oracle_grads(seed, shard):
    """the oracle's gradient of one training forward on `shard` from the state make_state_dict(seed): name -> tensor"""
    from oracle import cenet_oracle as O
    from oracle.golden_cases import MODEL_CONFIGS, config_from_kwargs
    cfg = config_from_kwargs(MODEL_CONFIGS["acdc"]["kw"])
    sd = O.make_state_dict(cfg, seed=seed)
    params = {k: v.requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running_" not in k}
    x, lab = shard
    O.criterion(O.cenet_forward(sd, x, cfg, training=True), lab, 4).backward()
    return {k: v.grad for k, v in params.items()}, {k: v.detach() for k, v in params.items()}


@pytest.mark.slow
def test_pieces_leave_the_oracle_gradient_and_each_segment_is_final_after_its_piece():
    """host SIMT checker, fp32: ONE cut training pass.  After piece k the arena slice of segment k never changes again, every
    piece flushes its own recorded weight gradients, and the five pieces together leave the gradient of the whole model (the
    oracle's, relative L2 < 1e-2: batch-2 BatchNorm at 1x1 .. 8x8 maps amplifies fp32 noise, see test_parallel_gloo.py)."""
    from cenet_amd import graph, ops, optim
    dev_ = use_sim()
    try:
        net = _net().to(dev_)
        arena = optim.ParamArena(net, optim.cenet_segments())
        crit = _crit()
        x, lab = _cenet_shard(0)
        cuts = graph.enable_segment_cuts(net)
        arena.zero_grad()
        loss = crit(net(x), lab)
        assert len(cuts) == 4 and all(leaf.is_leaf and leaf.requires_grad for _, leaf in cuts)
        snaps = []
        for k, piece in enumerate(graph.backward_pieces(loss, cuts)):
            piece()
            assert not ops._WgradQueue.items
            _, s, e = arena.segments[k]
            snaps.append(arena.grads[s:e].clone())
        graph.disable_segment_cuts(net)
        for k, (name, s, e) in enumerate(arena.segments):
            assert torch.equal(snaps[k], arena.grads[s:e]), f"segment {name} changed after its piece"
        want, _ = _oracle_grads(7, (x, lab))
        num = den = 0.0
        for name, (off, n) in arena.index.items():
            w = want[name].reshape(-1)
            num += float(((arena.grads[off:off + n] - w) ** 2).sum())
            den += float((w ** 2).sum())
        assert (num / den) ** 0.5 < 1e-2, (num / den) ** 0.5
    finally:
        from cenet_amd import _lib
        _lib._LIB, _lib._HOSTSIM = None, False


@pytest.mark.gpu
@pytest.mark.parametrize("bf16", [False, True], ids=["f32", "bf16"])
def test_pieces_equal_one_backward_on_the_gpu(bf16):
    """the cut backward against the whole backward of the same forward, both storage modes, at 224x224 — on the
    well-conditioned point of tests/test_wellcond.py (reference initialisation, batch 8): at the random-filled golden points
    two bf16 evaluations of the SAME backward differ by ~20 % in relative L2 through the order of fp32 atomics alone"""
    from backend import use_hip
    from cenet_amd import graph, kern, ops, optim
    import test_wellcond as W
    dev_ = use_hip()
    net, x, lab = W.build_product(W.golden(), dev_)
    net.train()
    net.backbone.reset_drop_path(0.0)
    arena = optim.ParamArena(net, optim.cenet_segments())
    crit = _crit()
    old = kern.set_compute_bf16(bf16)
    try:
        arena.zero_grad()
        crit(net(x), lab).backward()
        ops.wgrad_join()
        whole = arena.grads.clone()
        cuts = graph.enable_segment_cuts(net)
        arena.zero_grad()
        loss = crit(net(x), lab)
        snaps = []
        for k, piece in enumerate(graph.backward_pieces(loss, cuts)):
            piece()
            _, s, e = arena.segments[k]
            snaps.append(arena.grads[s:e].clone())
        graph.disable_segment_cuts(net)
    finally:
        kern.set_compute_bf16(old)
    torch.cuda.synchronize()
    for k, (name, s, e) in enumerate(arena.segments):
        assert torch.equal(snaps[k], arena.grads[s:e]), f"segment {name} changed after its piece"
    # same kernels on the same values; only the order of float atomics (and, in bf16, of the two-term leaf sums) differs
    assert ((arena.grads - whole).norm() / whole.norm()).item() < (1e-2 if bf16 else 2e-5)


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from cenet_amd import graph, optim, parallel
    torch.set_num_threads(2)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        use_sim()
        net = _net(seed=7 + 13 * rank)  # rank 1 starts elsewhere: the broadcast fixes it
        arena = optim.ParamArena(net, optim.cenet_segments())
        red = parallel.GradReducer(arena)
        red.broadcast_state(net)
        opt = optim.FusedSGD(arena, lr=LR, momentum=0.9, weight_decay=WD, grad_scale=red.grad_scale)
        crit = _crit()
        x, lab = _cenet_shard(rank)
        started = []
        orig = red.segment_ready

        def on_segment(i):
            started.append(i)
            orig(i)
        step = graph.SegmentedStep(net, lambda: crit(net(x), lab), opt, on_segment, red.finish, graphs=False)
        loss = step().item()
        q.put((rank, loss, arena.params.clone().numpy(), started, {n: arena.index[n] for n in arena.index}))
    finally:
        dist.destroy_process_group()


LR, WD = 0.05, 1e-4


@pytest.mark.slow
def test_two_ranks_segmented_step_takes_the_mean_gradient_step():
    """2 gloo ranks on the host checker, one SegmentedStep (eager pieces, GradReducer.segment_ready after each): one collective
    per segment in arena order, both ranks end with IDENTICAL parameters, and the update equals the first SGD step on the mean
    of the per-shard ORACLE gradients: p - lr * (mean g + wd * p)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=900) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    torch.testing.assert_close(torch.from_numpy(res[0][2]), torch.from_numpy(res[1][2]), rtol=0, atol=0)  # lock-step
    assert res[0][3] == [0, 1, 2, 3, 4] and res[1][3] == [0, 1, 2, 3, 4]
    g0, p0 = _oracle_grads(7, _cenet_shard(0))
    g1, _ = _oracle_grads(7, _cenet_shard(1))
    got, index = torch.from_numpy(res[0][2]), res[0][4]
    num = den = 0.0
    for name, (off, n) in index.items():
        p = p0[name].reshape(-1)
        want = -LR * ((g0[name] + g1[name]).reshape(-1) / world + WD * p)
        upd = got[off:off + n] - p
        num += float(((upd - want) ** 2).sum())
        den += float((want ** 2).sum())
    assert (num / den) ** 0.5 < 1e-2, (num / den) ** 0.5
