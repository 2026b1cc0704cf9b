"""Two data-parallel RANKS on the GPU (both processes share cuda:0; the collectives run over gloo, which takes CUDA tensors —
RCCL refuses two ranks on one device, and only a one-GPU box is available to the tests).  This exercises, with world_size 2 and
real device streams, everything of the N > 1 path except RCCL itself: the initial broadcast, per-rank shards, the three launch
forms of the step that bench.py chooses from —

    eager       backward hooks start each arena segment's all-reduce on the communication stream
    split       forward + backward as one hipGraph, the five all-reduces, the SGD graph (cenet_amd.graph.GraphedSplitStep)
    segmented   the backward cut into five hipGraphs with segment k's all-reduce issued behind graph k (SegmentedStep)

— and asserts for each form that the two ranks end with IDENTICAL parameters (lock-step) and that all three forms arrive at
the same parameters as each other (same shards, same arithmetic up to the order of float atomics).  bf16 mode, the
well-conditioned model of tests/test_wellcond.py, batch 4 per rank, three steps."""
import argparse
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
STEPS = 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, form, q):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import test_wellcond as W
        from cenet_amd import graph, kern, losses, optim, parallel
        from oracle import cenet_oracle as O
        dev = torch.device("cuda:0")
        kern.set_compute_bf16(True)
        net, _, _ = W.build_product(W.golden(), dev)
        if rank == 1:  # rank 1 starts from different parameters: the broadcast must fix it
            with torch.no_grad():
                for p in net.parameters():
                    p.mul_(1.01)
        net.train()
        net.backbone.reset_drop_path(0.0)
        arena = optim.ParamArena(net, optim.cenet_segments())
        red = parallel.GradReducer(arena)
        red.broadcast_state(net)
        opt = optim.FusedSGD(arena, lr=0.01, momentum=0.9, weight_decay=1e-4, grad_scale=red.grad_scale)
        crit = losses.Criterion(4, argparse.Namespace(loss_type="dice,ce", loss_weights="0.5,0.5"))
        x, lab = O.synthetic_batch(4, 1, 4, seed=100 + rank)
        x, lab = x.to(dev), lab.to(dev)

        def fwd_bwd():
            opt.zero_grad()
            loss = crit(net(x), lab)
            loss.backward()
            return loss

        losses_ = []
        if form == "eager":
            parallel.attach(net, red)
            for _ in range(STEPS):
                loss = fwd_bwd()
                red.finish()
                opt.step()
                losses_.append(float(loss))
        else:
            if form == "split":
                step = graph.GraphedSplitStep(fwd_bwd, opt, red.finish, warmup=1)
            else:
                step = graph.SegmentedStep(net, lambda: crit(net(x), lab), opt, red.segment_ready, red.finish, warmup=1)
            # warm-up (1) + capture (1) already trained two steps
            for _ in range(STEPS - 2):
                losses_.append(float(step()))
        torch.cuda.synchronize()
        q.put((rank, form, losses_, arena.params.detach().cpu().numpy()))
    finally:
        dist.destroy_process_group()


def _run(form):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, form, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=900) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


def test_three_launch_forms_keep_two_ranks_in_lock_step_and_agree():
    out = {form: _run(form) for form in ("eager", "split", "segmented")}
    for form, (r0, r1) in out.items():
        a, b = torch.from_numpy(r0[3]), torch.from_numpy(r1[3])
        assert torch.equal(a, b), f"{form}: the ranks' parameters differ after {STEPS} steps"
        assert all(torch.isfinite(torch.tensor(r0[2]))), (form, r0[2])
    ref = torch.from_numpy(out["eager"][0][3])
    start = None
    for form in ("split", "segmented"):
        got = torch.from_numpy(out[form][0][3])
        rel = ((got - ref).norm() / ref.norm()).item()
        cos = torch.nn.functional.cosine_similarity(got - got.mean(), ref - ref.mean(), dim=0).item()
        assert rel < 2e-3 and cos > 0.9999, (form, rel, cos)
