"""Segmented backward (cenet_amd.graph: enable_segment_cuts / backward_pieces / SegmentedStep): the backward pass cut at the four
encoder stage outputs into five autograd runs, one per gradient-arena segment, so that under hipGraph replay each segment's
all-reduce can start between two graphs and overlap the rest of the backward pass (bench.py at N > 1).

* GPU, single process: the five pieces leave the gradient of one whole backward pass, and after piece k the arena slice of
  segment k is FINAL (fp32 and bf16 storage);
* two gloo ranks on CPU (host SIMT checker): a SegmentedStep as eager pieces — callback order, finality of every segment after
  its piece, identical parameters on both ranks, update == SGD step on the mean of the per-shard ORACLE gradients;
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
        started, snaps, final = [], {}, {}

        def on_segment(i):
            # what the segment's all-reduce would read if it started now (it is started by finish() below instead, so that
            # the LOCAL slice can be compared with what it is when the whole backward pass has ended)
            started.append(i)
            _, s, e = arena.segments[i]
            snaps[i] = arena.grads[s:e].clone()

        def finish():
            for i, (_, s, e) in enumerate(arena.segments):
                final[i] = bool(torch.equal(snaps[i], arena.grads[s:e]))
            red.finish()
        step = graph.SegmentedStep(net, lambda: crit(net(x), lab), opt, on_segment, finish, graphs=False)
        loss = step().item()
        from cenet_amd import ops
        assert not ops.wgrad_pending()
        q.put((rank, loss, arena.params.clone().numpy(), started, {n: arena.index[n] for n in arena.index}, final))
    finally:
        dist.destroy_process_group()


LR, WD = 0.05, 1e-4


@pytest.mark.slow
def test_two_ranks_segmented_step_takes_the_mean_gradient_step():
    """2 gloo ranks on the host checker, one SegmentedStep run as eager pieces: the segment callback fires once per arena segment
    in arena order; after piece k the LOCAL gradient slice of segment k never changes again (it is final when its all-reduce
    would start); both ranks end with IDENTICAL parameters, and the update equals the first SGD step on the mean of the
    per-shard ORACLE gradients: p - lr * (mean g + wd * p)."""
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
    assert all(res[0][5].values()) and all(res[1][5].values()), f"a segment changed after its piece: {res[0][5]} {res[1][5]}"
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
