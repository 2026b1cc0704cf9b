"""Data-parallel semantics of cenet_amd.parallel.GradReducer with 2 gloo ranks on CPU:
n-rank gradient == mean of the single-rank gradients on the same shards, parameters stay in lock-step after the
update, BN running buffers stay per-rank (no SyncBN), segments are reduced from backward hooks (overlap path) and
the leftovers at finish().  The model is a toy whose parameters live in a ParamArena exactly like CENet's; the
reducer is model-agnostic."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


class Toy(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Sequential(nn.Conv2d(1, 4, 3, padding=1), nn.BatchNorm2d(4), nn.ReLU())
        self.b = nn.Sequential(nn.Conv2d(4, 4, 3, padding=1), nn.BatchNorm2d(4), nn.ReLU())
        self.c = nn.Conv2d(4, 2, 1)
        self._grad_sync = None

    def forward(self, x):
        h1 = self.a(x)
        h2 = self.b(h1)
        if self._grad_sync is not None and h1.requires_grad:
            h2.register_hook(self._grad_sync.hook(0))  # segment 0 ("c") final when backward reaches h2
            h1.register_hook(self._grad_sync.hook(1))  # segment 1 ("b") final when backward reaches h1
        return self.c(h2)


SEGS = [("c", lambda n: n.startswith("c.")), ("b", lambda n: n.startswith("b.")), ("a", lambda n: n.startswith("a."))]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make(seed=0):
    torch.manual_seed(seed)
    return Toy()


def _data(rank):
    g = torch.Generator().manual_seed(100 + rank)
    return torch.randn(3, 1, 8, 8, generator=g), torch.randn(3, 2, 8, 8, generator=g)


def _worker(rank, world, port, q):
    from cenet_amd import optim, parallel
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        net = _make(seed=rank)  # different init per rank: broadcast_state must fix it
        arena = optim.ParamArena(net, SEGS)
        red = parallel.GradReducer(arena)
        red.broadcast_state(net)
        parallel.attach(net, red)
        x, y = _data(rank)
        fired = []
        orig = red.segment_ready
        red.segment_ready = lambda i: (fired.append(i), orig(i))[1]
        loss = ((net(x) - y) ** 2).mean()
        loss.backward()
        hooks_fired = list(fired)
        red.finish()
        grads = arena.grads.clone() * red.grad_scale
        with torch.no_grad():
            arena.params.sub_(0.1 * grads)
        # plain numpy through the queue: torch tensors travel as shared-memory handles that the parent can only open while
        # this process is still alive (a race that showed up as FileNotFoundError under a loaded machine)
        q.put((rank, grads.numpy(), arena.params.clone().numpy(), net.a[1].running_mean.clone().numpy(), hooks_fired))
    finally:
        dist.destroy_process_group()


def test_two_rank_gradients_are_the_mean_of_shard_gradients():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    res = [(r, torch.from_numpy(g), torch.from_numpy(p_), torch.from_numpy(rm), fired) for r, g, p_, rm, fired in res]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process oracle: rank-0 weights on each shard, mean of the gradients
    from cenet_amd import optim
    ref_grads, ref_rm = [], []
    for r in range(world):
        net = _make(seed=0)
        arena = optim.ParamArena(net, SEGS)
        x, y = _data(r)
        ((net(x) - y) ** 2).mean().backward()
        ref_grads.append(arena.grads.clone())
        ref_rm.append(net.a[1].running_mean.clone())
    mean_grad = sum(ref_grads) / world
    for r, grads, params, rm, fired in res:
        torch.testing.assert_close(grads, mean_grad, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(rm, ref_rm[r], rtol=1e-6, atol=1e-7)  # BN buffers are per rank (no SyncBN)
        assert fired[:2] == [0, 1], f"segments must be reduced from backward hooks in order, got {fired}"
    torch.testing.assert_close(res[0][2], res[1][2], rtol=0, atol=0)  # parameters identical after the step


def test_arena_views_and_segments():
    from cenet_amd import optim
    net = _make()
    before = {n: p.detach().clone() for n, p in net.named_parameters()}
    arena = optim.ParamArena(net, SEGS)
    assert [s[0] for s in arena.segments] == ["c", "b", "a"]
    for n, p in net.named_parameters():
        torch.testing.assert_close(p.detach(), before[n])
        off, cnt = arena.index[n]
        assert p.data_ptr() == arena.params.data_ptr() + 4 * off
        assert p.grad.data_ptr() == arena.grads.data_ptr() + 4 * off
        assert off % optim.ALIGN == 0
    with pytest.raises(ValueError):
        optim.ParamArena(_make(), [("c", lambda n: n.startswith("c."))])
