"""The reference's CALLER, restated: what src/main_acdc.py does with the model object between construction and the end of a
training iteration (main_acdc.py:110-132 construction + `.cuda()` + the deepcopy of utils/utils.py:113; :171-180 checkpoint
load and the `nn.DataParallel` wrap; :185-199 `.train()`, criterion, AMP switch; :237-257 the step body with
`torch.optim.SGD(net.parameters(), ...)` of utils/core.py:19-21 and the poly `LambdaLR` of core.py:31).  The script itself is
never shipped or imported; these tests replay its calls on `cenet_amd.networks.CENet`.

* plain-fp32 protocol: three iterations driven by torch.optim.SGD + LambdaLR give the loss trajectory of the fused path
  (ParamArena + FusedSGD + PolyLR) AND of the oracle trained on the host with the same optimizer;
* the same under `copy.deepcopy`, `nn.DataParallel` on one device, `state_dict()` -> `load_state_dict(strict=True)`;
* AMP protocol (`autocast('cuda')` + `GradScaler`): the forward runs in the bf16 mode, the scaler neither skips a step nor
  changes its scale, the trajectory follows the fp32 one within the bf16 tolerance;
* `torch.compile(net, mode='default', fullgraph=True)` + `torch.compile(criterion)` (main_acdc.py:188-191): the forward is one
  opaque `cenet_amd::forward` operator for the tracer (cenet_amd/opaque.py); the compiled model trains like the eager one;
* `print_param_flops` (utils/utils.py:171-181): the FLOP table entry of that operator reproduces the reference's 12.76 G;
* two models trained alternately in one process (no process-global state in the weight-gradient machinery)."""
import argparse
import copy

import pytest
import torch
import torch.nn as nn

from backend import use_hip
from oracle import cenet_oracle as O

ACDC_ARGS = dict(input_channels=1, num_classes=4, scale_factors=[1.0, 0.5], encoder="pvt_v2_b2", enc_pretrain=False,
                 freeze_bb=False, skip_mode="cat", diffatt_num_heads=[4, 4, 4], dec_up_block="eucb", out_merge_mode="cat",
                 out_up_block="upcn", out_up_ks=3, base_ptdir=".")
BASE_LR, WD, MAX_IT = 0.01, 1e-4, 100  # (the reference presets' base_lr, acdc.sh)
STEPS = 3


def _criterion():
    from cenet_amd import losses
    return losses.Criterion(4, argparse.Namespace(loss_type="dice,ce", loss_weights="0.5,0.5"))


def _build(dev):
    """main_acdc.py:110-126: CENet(**flags).cuda() — with reproducible, non-degenerate weights loaded the way a checkpoint is
    (main_acdc.py:175)"""
    from cenet_amd.networks import CENet
    net = CENet(**ACDC_ARGS).to(dev)
    sd = O.make_state_dict(O.CENetConfig(), seed=3)
    net.load_state_dict(sd, strict=True)
    return net, sd


def _drive(net, x, lab, amp=False):
    """main_acdc.py:185-257 with get_optimizer('sgd') / get_scheduler('poly') of utils/core.py"""
    net.train()
    (net.module if isinstance(net, nn.DataParallel) else net).backbone.reset_drop_path(0.0)  # (deterministic trajectory)
    criterion = _criterion()
    optimizer = torch.optim.SGD(net.parameters(), lr=BASE_LR, weight_decay=WD, momentum=0.9)
    scheduler = torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda step: (1 - step / MAX_IT) ** 0.9)
    scaler = torch.amp.GradScaler() if amp else None
    losses_, lrs = [], []
    for _ in range(STEPS):
        optimizer.zero_grad()
        if amp:
            with torch.amp.autocast(device_type="cuda"):
                outputs = net(x)
                loss = criterion(outputs, lab[:])
            scaler.scale(loss).backward()
            scaler.step(optimizer)
            scaler.update()
        else:
            outputs = net(x)
            loss = criterion(outputs, lab[:])
            loss.backward()
            optimizer.step()
        lrs.append(scheduler.get_last_lr()[0])
        scheduler.step()
        losses_.append(loss.item())
    return losses_, lrs, outputs, scaler


def _oracle_trajectory(sd, x, lab):
    sdo = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running_" not in k else v.clone())
           for k, v in sd.items()}
    params = [v for v in sdo.values() if v.requires_grad]
    optimizer = torch.optim.SGD(params, lr=BASE_LR, weight_decay=WD, momentum=0.9)
    scheduler = torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda step: (1 - step / MAX_IT) ** 0.9)
    out = []
    for _ in range(STEPS):
        optimizer.zero_grad()
        loss = O.criterion(O.cenet_forward(sdo, x, O.CENetConfig(), training=True), lab, 4)
        loss.backward()
        optimizer.step()
        scheduler.step()
        out.append(loss.item())
    return out


@pytest.mark.gpu
def test_fp32_protocol_matches_the_fused_path_and_the_oracle():
    from cenet_amd import optim
    dev = use_hip()
    x, lab = O.synthetic_batch(2, 1, 4, seed=9)
    xd, labd = x.to(dev), lab.to(dev)
    net, sd = _build(dev)
    clone = copy.deepcopy(net)  # utils/utils.py:113 (CalParams) — and an independent replica for the fused path
    got, lrs, outputs, _ = _drive(net, xd, labd)
    assert outputs.shape == (2, 4, 224, 224) and outputs.dtype == torch.float32
    assert lrs == pytest.approx([BASE_LR * (1 - i / MAX_IT) ** 0.9 for i in range(STEPS)])
    # the fused path (ParamArena + FusedSGD + PolyLR) on the deep copy
    clone.train()
    clone.backbone.reset_drop_path(0.0)
    arena = optim.ParamArena(clone, optim.cenet_segments())
    opt = optim.FusedSGD(arena, lr=BASE_LR, momentum=0.9, weight_decay=WD)
    sched = optim.PolyLR(opt, max_iterations=MAX_IT)
    crit = _criterion()
    fused = []
    for _ in range(STEPS):
        opt.zero_grad()
        loss = crit(clone(xd), labd)
        loss.backward()
        opt.step()
        sched.step()
        fused.append(loss.item())
    ref = _oracle_trajectory(sd, x, lab)
    for i, (a, b, c) in enumerate(zip(got, fused, ref)):
        # (round 5: with a ParamArena the small decoder levels run the channel-local fused chains, without one the launch chains —
        # the same arithmetic in a different summation order, which the updates amplify like any other fp32 round-off)
        assert abs(a - b) < (2e-5 if i == 0 else 1e-4), (got, fused)
        # first iteration: the north-star bound on the loss; later iterations compare two fp32 TRAINING trajectories, whose
        # rounding differences the updates amplify (random-filled weights, batch 2: 8e-5 after one update at lr 0.05)
        assert abs(a - c) < (2e-4 if i == 0 else 5e-4), (got, ref)
    assert got[-1] < got[0]  # and it trains
    # the parameters the two optimizers arrive at
    pa, pb = dict(net.named_parameters()), dict(clone.named_parameters())
    for k in ("out.out.1.conv.conv.weight", "decoder.dec2.mca.gate.weight", "backbone.block3.2.mlp.fc1.weight",
              "backbone.patch_embed1.proj.weight"):
        assert (pa[k] - pb[k]).abs().max().item() < 1e-5 + 1e-3 * pb[k].abs().max().item(), k


def test_multi_device_dataparallel_replica_is_refused():
    """main_acdc.py:178-179 with several devices: nn.DataParallel replicates the module per device (torch.nn.parallel.replicate marks
    every copy `_is_replica`); the copies' parameters are Broadcast outputs, and the in-place gradient accumulation of the HIP path would
    never reach the wrapped module.  CENet.forward refuses such a replica with a message that names the two supported forms — it does
    not run on one device silently, and it does not train with lost gradients."""
    from torch.nn.parallel import replicate  # noqa: F401  (the attribute below is what it sets on every replica)
    from cenet_amd.networks import CENet
    net = CENet(**ACDC_ARGS)
    rep = copy.copy(net)
    rep._is_replica = True
    with pytest.raises(RuntimeError, match="multi-device nn.DataParallel.*one process per GPU"):
        rep(torch.zeros(1, 1, 224, 224))


@pytest.mark.gpu
def test_dataparallel_wrap_and_state_dict_round_trip():
    dev = use_hip()
    x, lab = O.synthetic_batch(2, 1, 4, seed=9)
    xd, labd = x.to(dev), lab.to(dev)
    net, sd = _build(dev)
    plain, _, _, _ = _drive(copy.deepcopy(net), xd, labd)
    wrapped = nn.DataParallel(net, device_ids=[0]).cuda()  # main_acdc.py:178-180 on a one-GPU box
    got, _, outputs, _ = _drive(wrapped, xd, labd)
    assert got == pytest.approx(plain, abs=2e-5)
    # main_acdc.py:278 saves net.state_dict(); :175 loads it strictly into a freshly built model
    from cenet_amd.networks import CENet
    saved = {k: v.detach().cpu().clone() for k, v in wrapped.module.state_dict().items()}
    import json
    import os
    schema = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "schema_acdc.json")))
    assert list(saved) == list(schema)  # the reference's 801 keys, in the reference's order (fixture written from its state_dict)
    fresh = CENet(**ACDC_ARGS).to(dev)
    fresh.load_state_dict(saved, strict=True)
    fresh.eval(), wrapped.eval()
    with torch.no_grad():
        assert torch.equal(fresh(xd), wrapped(xd))


@pytest.mark.gpu
def test_amp_protocol_runs_the_bf16_mode_under_the_callers_gradscaler():
    dev = use_hip()
    x, lab = O.synthetic_batch(2, 1, 4, seed=9)
    xd, labd = x.to(dev), lab.to(dev)
    net, _ = _build(dev)
    fp32, _, _, _ = _drive(copy.deepcopy(net), xd, labd)
    before = net.out.out[1].conv.conv.weight.detach().clone()
    got, _, outputs, scaler = _drive(net, xd, labd, amp=True)
    assert outputs.dtype == torch.bfloat16  # the autocast region selected the bf16 mode (net.py forward)
    assert scaler.get_scale() == 65536.0    # no overflow was ever seen: no step skipped, scale never backed off
    assert not torch.equal(before, net.out.out[1].conv.conv.weight)  # the steps were taken
    for a, b in zip(got, fp32):
        assert abs(a - b) < 5e-3, (got, fp32)
    # outside the region the model is back in fp32 (val(), main_acdc.py:218-231)
    net.eval()
    with torch.no_grad():
        assert net(xd).dtype == torch.float32


def _one_step(model, net, x, lab, criterion):
    for p in net.parameters():
        p.grad = None
    out = model(x)
    loss = criterion(out, lab)
    loss.backward()
    return loss.item(), out.detach().clone(), {n: p.grad.detach().clone() for n, p in net.named_parameters()}


@pytest.mark.gpu
def test_torch_compile_fullgraph_trains_like_eager():
    """main_acdc.py:188-191: `net = torch.compile(net, mode='default', fullgraph=True); criterion = torch.compile(criterion)`,
    then the step body of :237-257.  One training step of the compiled pair against the eager pair from the same state (fp32
    mode, DropPath off): loss and logits equal, every parameter gradient present and equal up to the order of fp32 atomics."""
    dev = use_hip()
    net, _ = _build(dev)
    net.train()
    net.backbone.reset_drop_path(0.0)
    ref = copy.deepcopy(net)
    x, lab = O.synthetic_batch(2, 1, 4, seed=9)
    xd, labd = x.to(dev), lab.to(dev)
    crit = _criterion()
    l0, o0, g0 = _one_step(ref, ref, xd, labd, crit)
    cnet = torch.compile(net, mode="default", fullgraph=True)
    ccrit = torch.compile(_criterion())
    l1, o1, g1 = _one_step(cnet, net, xd, labd, ccrit)
    assert abs(l0 - l1) < 1e-5 and torch.allclose(o0, o1, rtol=0, atol=1e-5)
    assert set(g0) == set(g1) and len(g1) == 630
    for n in g0:
        d = (g0[n] - g1[n]).norm().item()
        assert d <= 2e-3 * g0[n].norm().item() + 1e-7, (n, d, g0[n].norm().item())
    # a second call replays the compiled graph (no recompilation per step), eval mode is its own graph
    l2, _, _ = _one_step(cnet, net, xd, labd, ccrit)
    assert abs(l2 - l1) < 1e-5
    net.eval()
    with torch.no_grad():
        assert torch.equal(cnet(xd), net(xd))  # (same module, same BatchNorm buffers: the traced and the eager forward)


def test_torch_compile_fullgraph_on_the_host_checker():
    """the tracing protocol itself needs no GPU: one training step at 32x32 on the host SIMT checker, eager against
    torch.compile(fullgraph=True) (aot_eager backend: the tracer, functionalisation and the joint forward / backward graph are the
    real ones, only code generation is skipped)"""
    from backend import use_sim
    import test_segmented as TS
    from test_parallel_gloo import _cenet_shard
    from cenet_amd import _lib
    use_sim()
    try:
        net = TS._net(seed=7)
        ref = copy.deepcopy(net)
        crit = TS._crit()
        x, lab = _cenet_shard(0)
        l0, o0, g0 = _one_step(ref, ref, x, lab, crit)
        cnet = torch.compile(net, mode="default", fullgraph=True, backend="aot_eager")
        l1, o1, g1 = _one_step(cnet, net, x, lab, crit)
        assert l0 == l1 and torch.equal(o0, o1) and set(g0) == set(g1)
        for n in g0:
            assert torch.allclose(g0[n], g1[n], rtol=1e-5, atol=1e-7), n
        # (torch.jit.trace of the same module: test_jit_trace_of_the_model_completes, on the GPU — three more forwards of the
        # full network cost the host checker a minute)
    finally:
        _lib._LIB, _lib._HOSTSIM = None, False


def test_opaque_backward_finds_its_own_forward():
    """`cenet_amd::backward` looks its graph up by the token its forward returned, not by arrival order: a train-mode
    forward whose backward never runs (metrics only), then two forwards whose backwards run in REVERSE order — each gradient is
    the gradient of ITS batch.  (The pairing logic is independent of what the network computes: a two-parameter stand-in with the
    attributes the operator reads keeps this test at a second; the real network under `torch.compile` is
    test_torch_compile_fullgraph_on_the_host_checker.)"""
    import types
    from cenet_amd import opaque

    class Tiny(nn.Module):
        def __init__(self):
            super().__init__()
            self.p = nn.Parameter(torch.tensor([0.5, -1.5, 2.0]))
            conv = types.SimpleNamespace(conv=types.SimpleNamespace(out_channels=3))
            self.out = types.SimpleNamespace(w=self.p, out=[None, types.SimpleNamespace(conv=conv)])
            opaque.register(self)

        def _forward(self, x):  # [B, 1, H, W] -> [B, 3, H, W]
            return x * self.p.view(1, 3, 1, 1) + (x * x) * (self.p ** 2).view(1, 3, 1, 1)

    net = Tiny().train()
    xa, xb = torch.randn(2, 1, 4, 4), torch.randn(2, 1, 4, 4)
    want = []
    for x in (xa, xb):
        net.p.grad = None
        net._forward(x).sum().backward()
        want.append(net.p.grad.clone())
    opaque.forward(net, xa, False)  # a forward nobody differentiates
    net.p.grad = None
    ya, yb = opaque.forward(net, xa, False), opaque.forward(net, xb, False)
    assert len(net._cenet_live) == 3
    yb.sum().backward()
    assert torch.allclose(net.p.grad, want[1])
    net.p.grad = None
    ya.sum().backward()
    assert torch.allclose(net.p.grad, want[0])
    assert len(net._cenet_live) == 1  # (the undifferentiated one; it is dropped once _MAX_LIVE forwards pile up)
    for _ in range(opaque._MAX_LIVE + 2):
        opaque.forward(net, xa, False)
    assert len(net._cenet_live) == opaque._MAX_LIVE
    with pytest.raises(RuntimeError, match="graph of this forward is gone"):
        torch.ops.cenet_amd.backward(torch.zeros(1), torch.tensor([10 ** 9]), net._cenet_handle)


def test_flop_count_of_the_opaque_operator_is_the_references():
    """utils/utils.py:171-181 prints `FlopCountAnalysis(net, x).total() / 1e9` = 12.76 G for the ACDC preset (SURVEY.md section 6,
    the paper's figure).  fvcore is absent here; cenet_amd.flops prices the operator by fvcore's rules on a dry run (no kernel
    is launched, no GPU needed) and registers itself in fvcore's table when fvcore is importable."""
    from cenet_amd import flops
    from cenet_amd.networks import CENet
    net = CENet(**ACDC_ARGS)
    total = flops.count(net, (1, 1, 224, 224))
    assert abs(total / 1e9 - 12.76) < 0.01 * 12.76, total
    by = flops.count(net, (1, 1, 224, 224), by_op=True)
    assert by["gemm"] > 0.95 * total  # convolutions, linears and attention products dominate
    assert abs(flops.count(net, (2, 1, 224, 224)) - 2 * total) < 1e-6 * total  # (B == 1 skips the CCU's BatchNorm1d, cfam.py:258)
    assert net.training  # the mode is restored


@pytest.mark.gpu
def test_two_models_in_one_process_train_independently():
    """SURVEY section 8b "no global mutable state": two models built in one process, trained in alternation (bf16 mode, recorded
    weight gradients, the weight-gradient stream on), end where each ends when trained alone."""
    from cenet_amd import kern, ops, optim
    dev = use_hip()
    x, lab = O.synthetic_batch(2, 1, 4, seed=9)
    xd, labd = x.to(dev), lab.to(dev)
    crit = _criterion()

    def fresh(seed):
        from cenet_amd.networks import CENet
        net = CENet(**ACDC_ARGS).to(dev)
        net.load_state_dict(O.make_state_dict(O.CENetConfig(), seed=seed), strict=True)
        net.train()
        net.backbone.reset_drop_path(0.0)
        arena = optim.ParamArena(net, optim.cenet_segments())
        return net, optim.FusedSGD(arena, lr=0.01, momentum=0.9, weight_decay=1e-4), arena

    def step(net, opt):
        opt.zero_grad()
        loss = crit(net(xd), labd)
        loss.backward()
        opt.step()
        return loss.item()

    kern.set_compute_bf16(True)
    old = ops.set_wgrad_overlap(True)
    try:
        alone = []
        for seed in (3, 4):
            net, opt, arena = fresh(seed)
            alone.append(([step(net, opt) for _ in range(2)], arena.params.clone()))
        a, b = fresh(3), fresh(4)
        la, lb = [], []
        for _ in range(2):
            # interleaved: both forwards first, then both backwards (recorded weight gradients of two models in flight)
            a[1].zero_grad(), b[1].zero_grad()
            loss_a, loss_b = crit(a[0](xd), labd), crit(b[0](xd), labd)
            loss_a.backward()
            loss_b.backward()
            a[1].step(), b[1].step()
            la.append(loss_a.item()), lb.append(loss_b.item())
        torch.cuda.synchronize()
        for (l_ref, p_ref), l_got, arena in ((alone[0], la, a[2]), (alone[1], lb, b[2])):
            assert all(abs(u - v) < 2e-3 for u, v in zip(l_ref, l_got)), (l_ref, l_got)
            cos = torch.nn.functional.cosine_similarity(arena.params - p_ref * 0, p_ref, dim=0).item()
            assert cos > 0.999999, cos
            assert (arena.params - p_ref).norm().item() < 1e-3 * p_ref.norm().item()
    finally:
        ops.set_wgrad_overlap(old)
        kern.set_compute_bf16(False)


@pytest.mark.gpu
def test_jit_trace_of_the_model_completes():
    """utils/utils.py:171-185 (`print_param_flops`, main_acdc.py:128): fvcore's FlopCountAnalysis is a `torch.jit.trace` of the
    model followed by `parameter_count`.  fvcore is absent from this image; its mechanism is exercised directly: the trace
    completes, the traced module reproduces the eager output, and the parameter count is the reference's 33.38 M."""
    dev = use_hip()
    net, _ = _build(dev)
    net.eval()
    x, _ = O.synthetic_batch(1, 1, 4, seed=9)
    xd = x.to(dev)
    with torch.no_grad():
        traced = torch.jit.trace(net, xd, check_trace=False, strict=False)
        assert torch.equal(traced(xd), net(xd))
    assert sum(p.numel() for p in net.parameters()) == 33384872  # SURVEY.md §6 (ACDC preset)

