"""MultiOrderDWConv with its three dilated branches run as ONE chain of launches (parameters of equal role back to back in a
ParamArena: one BatchNorm launch over 3g channels, one batched GEMM for the three pointwise convs) against the same module
run branch by branch: outputs, input gradient, every parameter gradient, BatchNorm running statistics and counters.
Reference: networks/cenet/modules/cfam.py:162-241."""
import copy

import pytest
import torch

from backend import dev  # noqa: F401  (fixture: host SIMT checker / MI355X)
from cenet_amd import kern, ops, optim
from cenet_amd.networks.cenet.modules.cfam import MultiOrderDWConv


def _run(mod, arena, x, g, steps=2):
    outs = []
    for _ in range(steps):
        arena.zero_grad()
        xi = x.clone().requires_grad_(True)
        y = mod(xi)
        y.backward(g)
        ops.wgrad_join()
        outs.append((y.detach().float().clone(), xi.grad.detach().float().clone(), arena.grads.clone()))
    return outs


@pytest.mark.parametrize("C,hw,bf16", [(64, 14, False), (64, 14, True), (32, 7, False), (128, 8, True)])
def test_merged_equals_branchwise(dev, C, hw, bf16):  # noqa: F811
    torch.manual_seed(C + hw)
    old = kern.set_compute_bf16(bf16)
    try:
        ref = MultiOrderDWConv(C, rates=[1, 2, 3]).to(dev).train()
        with torch.no_grad():  # non-trivial BatchNorm parameters and statistics
            for m in list(ref.dlps)[:3]:
                for bn in (m.depthwise_bn, m.pointwise_bn):
                    bn.weight.uniform_(0.5, 1.5)
                    bn.bias.uniform_(-0.3, 0.3)
                    bn.running_mean.uniform_(-0.2, 0.2)
                    bn.running_var.uniform_(0.5, 1.5)
        mer = copy.deepcopy(ref)
        ref.arena_groups = lambda: []          # (instance attribute: this copy keeps the one-parameter-per-slot layout ...)
        ref._merged = lambda: None             # ... and runs branch by branch
        a_ref, a_mer = optim.ParamArena(ref), optim.ParamArena(mer)
        assert mer._merged() is not None, "arena_groups must make the branches' parameters mergeable"
        dt = torch.bfloat16 if bf16 else torch.float32
        x = torch.randn(3, C, hw, hw, device=dev).to(dt)
        g = torch.randn(3, C, hw, hw, device=dev).to(dt)
        r, m = _run(ref, a_ref, x, g), _run(mer, a_mer, x, g)
        tol = 3e-2 if bf16 else 2e-5
        for (yr, dxr, _), (ym, dxm, _) in zip(r, m):
            assert (yr - ym).abs().max() <= tol * max(1.0, yr.abs().max().item())
            assert (dxr - dxm).abs().max() <= tol * max(1.0, dxr.abs().max().item())
        # parameter gradients by name (the two arenas order their slots differently)
        for (n, pr), (_, pm) in zip(ref.named_parameters(), mer.named_parameters()):
            gr, gm = pr.grad.float(), pm.grad.float()
            assert (gr - gm).abs().max() <= tol * max(1.0, gr.abs().max().item()), n
        for (n, br), (_, bm) in zip(ref.named_buffers(), mer.named_buffers()):
            if n.endswith("num_batches_tracked"):
                assert int(br) == int(bm) == 2, n
            else:
                assert (br - bm).abs().max() <= 1e-5 + (1e-2 if bf16 else 0.0), n
        # eval mode: the joint running statistics feed one normalisation launch
        ref.eval(), mer.eval()
        with torch.no_grad():
            yr, ym = ref(x).float(), mer(x).float()
        assert (yr - ym).abs().max() <= tol * max(1.0, yr.abs().max().item())
    finally:
        kern.set_compute_bf16(old)


def test_arena_groups_layout(dev):  # noqa: F811
    """members of a group sit back to back in parameter, gradient and bf16 shadow slots; ungrouped parameters keep their
    ALIGN-padded slots; a state_dict round trip goes through the aliased tensors"""
    mod = MultiOrderDWConv(64, rates=[1, 2, 3]).to(dev)
    arena = optim.ParamArena(mod)
    b = list(mod.dlps)[:3]
    g = b[0].pointwise.weight.numel()
    for j in (1, 2):
        assert b[j].pointwise.weight.data_ptr() == b[0].pointwise.weight.data_ptr() + 4 * j * g
        assert b[j].pointwise.weight.grad.data_ptr() == b[0].pointwise.weight.grad.data_ptr() + 4 * j * g
    assert mod.PW_conv.weight.data_ptr() % (4 * optim.ALIGN) == arena.params.data_ptr() % (4 * optim.ALIGN)
    mg = mod._merged()
    assert mg is not None and mg["pw"].shape[0] == 3 and mg["pw"].data_ptr() == b[0].pointwise.weight.data_ptr()
    sd = {k: v.clone() + 1 for k, v in mod.state_dict().items()}
    mod.load_state_dict(sd)
    assert torch.equal(mg["pw"][1].reshape(-1).cpu(), (sd["dlps.1.pointwise.weight"]).reshape(-1).cpu())
    assert torch.equal(mg["dbn_running_mean"][2].cpu(), sd["dlps.2.depthwise_bn.running_mean"].cpu())


@pytest.mark.parametrize("E,H,N,bf16", [(64, 2, 40, False), (64, 2, 40, True), (128, 4, 70, True)])
def test_merged_qkv_projection(dev, E, H, N, bf16):  # noqa: F811
    """MultiheadDiffAttn with its q / k / v projections as one batched launch per pass (weights back to back in a ParamArena,
    ops.multi_linear) against the same module running three Linear ops; reference: modules/multihead_diffattn.py:79-81.
    fp32 tensors: the attention backward returns separate gradients (the chained data-gradient path); bf16: one buffer (the
    K-batched GEMM)."""
    from cenet_amd.networks.cenet.modules.multihead_diffattn import MultiheadDiffAttn
    torch.manual_seed(E + N)
    old = kern.set_compute_bf16(bf16)
    try:
        ref = MultiheadDiffAttn(E, depth=1, num_heads=H).to(dev).train()
        mer = copy.deepcopy(ref)
        ref.arena_groups = lambda: []
        ref._merged_qkv = lambda: None
        a_ref, a_mer = optim.ParamArena(ref), optim.ParamArena(mer)
        assert mer._merged_qkv() is not None
        dt = torch.bfloat16 if bf16 else torch.float32
        x = torch.randn(2, N, E, device=dev).to(dt)
        g = torch.randn(2, N, E, device=dev).to(dt)
        r, m = _run(ref, a_ref, x, g, steps=1), _run(mer, a_mer, x, g, steps=1)
        tol = 3e-2 if bf16 else 2e-5
        (yr, dxr, _), (ym, dxm, _) = r[0], m[0]
        assert (yr - ym).abs().max() <= tol * max(1.0, yr.abs().max().item())
        assert (dxr - dxm).abs().max() <= tol * max(1.0, dxr.abs().max().item())
        for (n, pr), (_, pm) in zip(ref.named_parameters(), mer.named_parameters()):
            gr, gm = pr.grad.float(), pm.grad.float()
            assert (gr - gm).abs().max() <= tol * max(1.0, gr.abs().max().item()), n
    finally:
        kern.set_compute_bf16(old)


@pytest.mark.parametrize("C,hw,bf16", [(64, 7, False), (64, 16, True), (128, 7, True), (32, 5, False)])
def test_merged_nonlocal_projections(dev, C, hw, bf16):  # noqa: F811
    """Nonlocal with conv_theta / conv_phi / conv_g as ONE 1x1 conv whose [B, 3C, N] output the attention reads in place
    (ops.nonlocal_attention_joint) against the three-conv path; reference: modules/nlb.py:102-148.  C = 64 with 256 tokens
    takes the token-major pair kernels (bf16), the others the tiled / materialised paths."""
    from cenet_amd.networks.cenet.modules.nlb import Nonlocal
    torch.manual_seed(C + hw)
    old = kern.set_compute_bf16(bf16)
    try:
        ref = Nonlocal(C).to(dev).train()
        with torch.no_grad():
            ref.bn.weight.uniform_(0.5, 1.5)  # (zero-initialised in the reference: would hide the attention branch)
        mer = copy.deepcopy(ref)
        ref.arena_groups = lambda: []
        ref._merged_tpg = lambda: None
        a_ref, a_mer = optim.ParamArena(ref), optim.ParamArena(mer)
        assert mer._merged_tpg() is not None
        dt = torch.bfloat16 if bf16 else torch.float32
        x = torch.randn(2, C, hw, hw, device=dev).to(dt)
        g = torch.randn(2, C, hw, hw, device=dev).to(dt)
        r, m = _run(ref, a_ref, x, g, steps=1), _run(mer, a_mer, x, g, steps=1)
        tol = 4e-2 if bf16 else 3e-5
        (yr, dxr, _), (ym, dxm, _) = r[0], m[0]
        assert (yr - ym).abs().max() <= tol * max(1.0, yr.abs().max().item())
        assert (dxr - dxm).abs().max() <= tol * max(1.0, dxr.abs().max().item())
        for (n, pr), (_, pm) in zip(ref.named_parameters(), mer.named_parameters()):
            gr, gm = pr.grad.float(), pm.grad.float()
            assert (gr - gm).abs().max() <= tol * max(1.0, gr.abs().max().item()), n
    finally:
        kern.set_compute_bf16(old)
