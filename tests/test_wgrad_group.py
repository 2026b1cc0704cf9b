"""Grouped weight gradients (cenet_wgrad_group_bf16, gemm_group.hip) and the deferral queue of cenet_amd.ops.

* the C-ABI entry against plain fp32 PyTorch on mixed problem lists: token-major Linear problems (both operands row-fast),
  NCHW 1x1-conv problems (both k-fast, K batches, unaligned 7x7 / 14x14 planes), ragged M / N / K, bias row sums, K slices
  (workspace + fold launch), two problems adding into ONE C (atomic path), more problems than one launch holds;
* results do not depend on scheduling: two runs are bit-identical;
* LinearFn / MultiLinearFn / Conv1x1Fn with the queue on give the gradients of the one-launch-per-layer path.
`sim` = the same kernel source on the host SIMT checker (CPU); `hip` = the gfx950 library (marker gpu)."""
import pytest
import torch

from backend import dev  # noqa: F401
from cenet_amd import kern, ops

BF = torch.bfloat16


def _lin_problem(g, R, N, K, dev_, bias=True):
    """token-major Linear: dY [R, N], X [R, K] -> dW [N, K] += dY^T X, db[N] += column sums of dY"""
    dy = (torch.randn(R, N, generator=g) * 0.5).to(BF)
    x = torch.randn(R, K, generator=g).to(BF)
    ref = dy.float().t() @ x.float()
    rb = dy.float().sum(0)
    return dict(kind="lin", dy=dy.to(dev_), x=x.to(dev_), M=N, N=K, K=R, nkb=1, ref=ref, rb=rb if bias else None)


def _conv_problem(g, B, Cout, Cin, HW, dev_, bias=True):
    """NCHW 1x1 conv: dY [B, Cout, HW], X [B, Cin, HW] -> dW [Cout, Cin] += sum_b dY_b X_b^T"""
    dy = (torch.randn(B, Cout, HW, generator=g) * 0.5).to(BF)
    x = torch.randn(B, Cin, HW, generator=g).to(BF)
    ref = torch.einsum("bop,bip->oi", dy.float(), x.float())
    rb = dy.float().sum((0, 2))
    return dict(kind="conv", dy=dy.to(dev_), x=x.to(dev_), M=Cout, N=Cin, K=HW, nkb=B, ref=ref, rb=rb if bias else None)


def _run(probs, dev_, pre=None):
    items, outs = [], []
    for i, p in enumerate(probs):
        dW = torch.zeros(p["M"], p["N"], device=dev_) if pre is None else pre[i][0]
        db = (torch.zeros(p["M"], device=dev_) if pre is None else pre[i][1]) if p["rb"] is not None else None
        outs.append((dW, db))
        if p["kind"] == "lin":
            items.append((p["dy"].data_ptr(), p["x"].data_ptr(), dW.data_ptr(), db.data_ptr() if db is not None else None,
                          p["M"], p["N"], 0, 0, p["M"], p["N"], p["K"], 1, 0))
        else:
            items.append((p["dy"].data_ptr(), p["x"].data_ptr(), dW.data_ptr(), db.data_ptr() if db is not None else None,
                          p["K"], p["K"], p["M"] * p["K"], p["N"] * p["K"], p["M"], p["N"], p["K"], p["nkb"], 1))
    kern.wgrad_group(items, dev_)
    if dev_.type == "cuda":
        torch.cuda.synchronize()
    return outs


def _check(probs, outs, tol=2e-3):
    for p, (dW, db) in zip(probs, outs):
        scale = p["ref"].abs().max().item() + 1e-6
        assert (dW.cpu() - p["ref"]).abs().max().item() < tol * scale, (p["kind"], p["M"], p["N"], p["K"], p["nkb"])
        if p["rb"] is not None:
            assert (db.cpu() - p["rb"]).abs().max().item() < tol * (p["rb"].abs().max().item() + 1e-6)


def test_mixed_problem_list(dev):
    g = torch.Generator().manual_seed(3)
    probs = [_lin_problem(g, 200, 64, 64, dev), _lin_problem(g, 130, 72, 136, dev), _lin_problem(g, 77, 48, 200, dev, bias=False),
             _conv_problem(g, 3, 64, 64, 49, dev), _conv_problem(g, 2, 80, 56, 196, dev), _conv_problem(g, 2, 96, 64, 64, dev, bias=False),
             # both sides >= 128: the 128x128 tile class (a second launch per orientation inside the same call)
             _lin_problem(g, 150, 136, 192, dev), _conv_problem(g, 2, 128, 160, 100, dev), _lin_problem(g, 90, 256, 128, dev, bias=False)]
    outs = _run(probs, dev)
    _check(probs, outs)


def test_long_reductions_are_sliced_and_folded(dev, monkeypatch):
    """K = 6 000 / 33 x 196 with a small depth target: several K slices per tile, partial tiles through the workspace"""
    monkeypatch.setenv("CENET_GROUP_DEPTH", "8")
    g = torch.Generator().manual_seed(4)
    probs = [_lin_problem(g, 6000, 64, 128, dev), _conv_problem(g, 33, 64, 64, 196, dev), _lin_problem(g, 300, 64, 64, dev)]
    outs = _run(probs, dev)
    _check(probs, outs)
    outs2 = _run(probs, dev)
    for (a, ab), (b, bb) in zip(outs, outs2):
        assert torch.equal(a, b) and (ab is None or torch.equal(ab, bb)), "the two-pass reduction is deterministic"


def test_accumulates_into_existing_gradient_and_shared_destination(dev):
    g = torch.Generator().manual_seed(5)
    p0, p1 = _lin_problem(g, 500, 64, 64, dev), _lin_problem(g, 260, 64, 64, dev)
    dW = torch.full((64, 64), 0.25, device=dev)
    db = torch.full((64,), -1.0, device=dev)
    _run([p0, p1], dev, pre=[(dW, db), (dW, db)])  # one parameter used twice: both problems add into the same C / asum
    ref = 0.25 + p0["ref"] + p1["ref"]
    assert (dW.cpu() - ref).abs().max().item() < 2e-3 * ref.abs().max().item()
    assert (db.cpu() - (-1.0 + p0["rb"] + p1["rb"])).abs().max().item() < 2e-3 * p0["rb"].abs().max().item()


def test_more_problems_than_one_launch(dev):
    g = torch.Generator().manual_seed(6)
    probs = [_lin_problem(g, 64 + 8 * i, 48 + 8 * (i % 3), 64, dev, bias=(i % 2 == 0)) for i in range(61)]
    outs = _run(probs, dev)
    _check(probs, outs)


def _grads(fn, params):
    for p in params:
        p.grad = None
    fn()
    ops.wgrad_join()
    return [p.grad.detach().clone() for p in params]


def test_deferred_layers_match_the_per_layer_launches(dev):
    """Linear (bias, residual, tap), the batched q / k / v projection and a 1x1 conv (bias) in one backward pass: the queue
    on (one grouped launch at the end of the pass) against the queue off (a launch per layer), same bf16 inputs"""
    g = torch.Generator().manual_seed(7)
    x = torch.randn(2, 70, 64, generator=g).to(BF).to(dev).requires_grad_(True)
    W1 = torch.nn.Parameter((torch.randn(96, 64, generator=g) * 0.1).to(dev))
    b1 = torch.nn.Parameter(torch.randn(96, generator=g).to(dev))
    W2 = torch.nn.Parameter((torch.randn(64, 96, generator=g) * 0.1).to(dev))
    Wq = torch.nn.Parameter((torch.randn(3, 64, 64, generator=g) * 0.1).to(dev))
    Wc = torch.nn.Parameter((torch.randn(56, 70, 1, 1, generator=g) * 0.1).to(dev))
    bc = torch.nn.Parameter(torch.randn(56, generator=g).to(dev))
    cot = torch.randn(2, 56, 64, generator=g).to(BF).to(dev)
    params = [W1, b1, W2, Wq, Wc, bc]

    def run():
        h = ops.linear(x, W1, b1)
        y = ops.linear(h, W2, None, resid=x)
        q, k, v = ops.multi_linear(y, Wq)
        z = ops.conv1x1((q + k + v).contiguous(), Wc, bc)  # [2, 70, 64] read as NCHW [B, C=70, HW=64]
        z.backward(cot)

    old = ops.set_wgrad_grouping(False)
    try:
        ref = _grads(run, params)
        ops.set_wgrad_grouping(True)
        got = _grads(run, params)
        assert not ops.wgrad_pending()
    finally:
        ops.set_wgrad_grouping(old)
    for r, t, n in zip(ref, got, ["W1", "b1", "W2", "Wq", "Wc", "bc"]):
        assert (r - t).abs().max().item() <= 2e-3 * r.abs().max().item() + 1e-6, n


def test_a_backward_that_raises_does_not_poison_the_next_one(dev):
    """a backward pass that raises after its first record never runs the end-of-backward callback; the NEXT pass must
    still flush by itself (a caller with a torch optimizer never calls wgrad_join) and must not add the failed pass's records."""
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 64, 64, generator=g).to(BF).to(dev)
    W1 = torch.nn.Parameter((torch.randn(64, 64, generator=g) * 0.1).to(dev))
    W2 = torch.nn.Parameter((torch.randn(64, 64, generator=g) * 0.1).to(dev))

    class Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return t.clone()

        @staticmethod
        def backward(ctx, gr):
            raise RuntimeError("boom")

    def run(fail):
        h = ops.linear(x, W1)
        if fail:
            h = Boom.apply(h)
        ops.linear(h, W2).float().sum().backward()

    run(False)
    ref = [W1.grad.clone(), W2.grad.clone()]
    W1.grad.zero_(), W2.grad.zero_()
    with pytest.raises(RuntimeError, match="boom"):
        run(True)  # W2's weight gradient is recorded, then the pass dies
    W1.grad.zero_(), W2.grad.zero_()
    run(False)  # no wgrad_join / wgrad_flush by the caller
    if dev.type == "cuda":
        torch.cuda.synchronize()
    assert not ops.wgrad_pending()
    for r, p in zip(ref, (W1, W2)):
        assert torch.equal(r, p.grad)


def test_group_plan_is_the_librarys_partition(dev):
    """bench.py brackets the grouped launches one by one: the partition comes from the library (cenet_wgrad_group_plan), not from
    constants copied into the bench — problems of one launch share orientation and tile, no launch exceeds 56 problems, launch
    indices are dense and ordered like the library issues them (row-fast before k-fast, skinny tiles before 128 x 128)."""
    t = torch.zeros(8, dtype=BF, device=dev)
    w = torch.zeros(8, device=dev)
    probs = []
    for i in range(130):
        M, N = (64, 320) if i % 3 == 0 else (256, 512)
        probs.append((t.data_ptr(), t.data_ptr(), w.data_ptr(), None, N, M, 0, 0, M, N, 64, 1, i % 2))
    plan = kern.wgrad_group_plan(probs)
    assert sorted({p[0] for p in plan}) == list(range(max(p[0] for p in plan) + 1))
    for li in {p[0] for p in plan}:
        members = [(pr, pl) for pr, pl in zip(probs, plan) if pl[0] == li]
        assert len(members) <= 56
        assert len({pr[12] for pr, _ in members}) == 1 and len({pl[1:] for _, pl in members}) == 1
        bm, bn, _ = members[0][1][1:]
        assert all((min(pr[8], pr[9]) >= 128) == (bm == 128 and bn == 128) for pr, _ in members)
    order = [(probs[i][12], plan[i][1]) for i in sorted(range(len(plan)), key=lambda i: plan[i][0])]
    assert order == sorted(order)


@pytest.mark.gpu
def test_whole_model_grouping_on_equals_off_and_memory_is_bounded():
    """the recorded (grouped) weight gradients against the per-layer launches on the WHOLE model (bf16 mode, the
    reference-initialised ACDC model, batch 8): every gradient-arena segment agrees as well as two identical runs agree with each
    other (the paths differ by the order of fp32 additions only), and the operands the queue keeps alive until the flush cost a bounded amount of peak memory."""
    import test_wellcond as W
    from backend import use_hip
    from cenet_amd import losses, optim
    import argparse
    dev = use_hip()
    z = W.golden("acdc")
    out = {}
    kern.set_compute_bf16(True)
    try:
        for tag, mode in (("off", False), ("off2", False), ("on", True)):
            old = ops.set_wgrad_grouping(mode)
            try:
                net, x, lab = W.build_product(z, dev)
                net.train()
                net.backbone.reset_drop_path(0.0)
                arena = optim.ParamArena(net, optim.cenet_segments())
                crit = losses.Criterion(4, argparse.Namespace(loss_type="dice,ce", loss_weights="0.5,0.5"))
                torch.cuda.synchronize()
                torch.cuda.reset_peak_memory_stats()
                crit(net(x), lab).backward()
                ops.wgrad_join()
                torch.cuda.synchronize()
                out[tag] = (arena.grads.detach().clone(), torch.cuda.max_memory_allocated(), arena.segments)
                del net, arena
            finally:
                ops.set_wgrad_grouping(old)
    finally:
        kern.set_compute_bf16(False)
    (g0, m0, segs), (g2, _, _), (g1, m1, _) = out["off"], out["off2"], out["on"]
    # self-calibrating (as tests/test_overlap.py): two identical runs differ by the order of float atomics upstream (split-K forward
    # convs, attention dK / dV), most in the deepest encoder stage; grouping may not add to that
    for name, s, e in segs:
        def dist(a, b):
            return 1.0 - torch.nn.functional.cosine_similarity(a[s:e].double(), b[s:e].double(), dim=0).item()
        assert dist(g0, g1) <= 4 * dist(g0, g2) + 1e-6, (name, dist(g0, g1), dist(g0, g2))
    assert (g0 - g1).norm().item() <= 4 * (g0 - g2).norm().item() + 1e-5 * g0.norm().item()
    # batch 8 at 224x224: the whole backward's dY / X operands stay alive with grouping on (measured +0.3 GB); bounded by the
    # queue's CENET_WGRAD_HOLD_MB (3 GB) whatever the batch
    assert m1 - m0 < 1.0 * (1 << 30), (m0, m1)
