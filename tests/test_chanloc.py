"""Channel-local fused chains (csrc/chanloc.hip, round 5) against the launch chains they replace.

The fused kernels keep intermediates in fp32 where the unfused chain stores them in the tensors' type, so in bf16 storage the two
differ by bf16 rounding of the intermediates (bounded here in relative L2); in fp32 storage they agree to fp32 round-off.  The
reference-made goldens of the modules (tests/test_modules_parity.py: eucb, cfa_module_*, mca_*) run through the fused path too.
`sim` = the same kernel source on the host SIMT checker (CPU); `hip` = the gfx950 library (marker gpu)."""
import pytest
import torch

from backend import dev  # noqa: F401
from cenet_amd import kern, ops

BF = torch.bfloat16


def _rel(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


def _eucb_chain(x, w, gamma, beta, rm, rv, nbt, fused):
    if fused:
        assert ops.eucb_front_supported(x, True)
        return ops.eucb_front(x, w, gamma, beta, rm, rv, nbt, 1e-5, 0.2, 0.1)
    y = ops.nearest2x(x)
    y = ops.dwconv_nchw(y, w, None, dil=1)
    return ops.batchnorm(y, gamma, beta, rm, rv, nbt, True, 1e-5, "lrelu", 0.2, 0.1)


# (B, C, H, W): the three EUCB levels at reduced batch / channels, odd planes, and a plane set that needs the grouped backward
@pytest.mark.parametrize("shape", [(4, 6, 7, 7), (3, 5, 14, 14), (2, 3, 28, 28), (2, 4, 5, 6), (32, 2, 7, 7), (6, 2, 40, 40)])
@pytest.mark.parametrize("dt", [torch.float32, BF], ids=["f32", "bf16"])
def test_eucb_front_equals_the_launch_chain(dev, shape, dt):
    B, Cn, H, W = shape
    g = torch.Generator().manual_seed(B * 100 + H)
    x0 = torch.randn(B, Cn, H, W, generator=g).to(dt).to(dev)
    cot = torch.randn(B, Cn, 2 * H, 2 * W, generator=g).to(dt).to(dev)
    res = []
    for fused in (False, True):
        x = x0.clone().requires_grad_(True)
        gg = torch.Generator().manual_seed(7)
        w = (0.3 * torch.randn(Cn, 1, 3, 3, generator=gg)).to(dev).requires_grad_(True)
        gamma = (1 + 0.2 * torch.randn(Cn, generator=gg)).to(dev).requires_grad_(True)
        beta = (0.2 * torch.randn(Cn, generator=gg)).to(dev).requires_grad_(True)
        rm, rv = torch.zeros(Cn).to(dev), torch.ones(Cn).to(dev)
        nbt = torch.zeros((), dtype=torch.long).to(dev)
        for p in (w, gamma, beta):
            p.grad = torch.zeros_like(p)
        y = _eucb_chain(x, w, gamma, beta, rm, rv, nbt, fused)
        y.backward(cot)
        ops.wgrad_join() if dev.type != "cpu" else None
        res.append((y.detach(), x.grad, w.grad, gamma.grad, beta.grad, rm, rv, nbt))
    tol = 2e-2 if dt == BF else 2e-5
    names = ("y", "dx", "dw", "dgamma", "dbeta", "running_mean", "running_var")
    for k, name in enumerate(names):
        assert _rel(res[1][k], res[0][k]) < tol, (name, _rel(res[1][k], res[0][k]))
    assert int(res[1][7]) == int(res[0][7]) == 1


def test_eucb_front_is_deterministic(dev):
    """no float atomics: two evaluations are bit-identical (outputs and every gradient)"""
    B, Cn, H, W = 5, 4, 14, 14
    g = torch.Generator().manual_seed(3)
    x0 = torch.randn(B, Cn, H, W, generator=g).to(BF).to(dev)
    cot = torch.randn(B, Cn, 2 * H, 2 * W, generator=g).to(BF).to(dev)
    outs = []
    for _ in range(2):
        x = x0.clone().requires_grad_(True)
        w = (0.3 * torch.ones(Cn, 1, 3, 3)).to(dev).requires_grad_(True)
        gamma, beta = torch.ones(Cn).to(dev).requires_grad_(True), torch.zeros(Cn).to(dev).requires_grad_(True)
        for p in (w, gamma, beta):
            p.grad = torch.zeros_like(p)
        y = ops.eucb_front(x, w, gamma, beta, None, None, None, 1e-5, 0.2, 0.1)
        y.backward(cot)
        outs.append((y.detach(), x.grad, w.grad, gamma.grad, beta.grad))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_eucb_module_takes_the_fused_path_in_training_only(dev, monkeypatch):
    from cenet_amd.networks.cenet.modules.blocks import EUCB
    m = EUCB(8, 4, activation="leakyrelu").to(dev)
    calls = []
    orig = kern.eucb_fwd
    monkeypatch.setattr(kern, "eucb_fwd", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
    x = torch.randn(2, 8, 7, 7).to(dev)
    m.train()
    m(x)
    assert len(calls) == 1
    m.eval()
    with torch.no_grad():
        m(x)
    assert len(calls) == 1


def _run_cfam(dev, dt, dims, hw, B, fused, monkeypatch, rates=(1, 2, 3)):
    """one training forward + backward of a CFAModule; returns outputs, input gradient, every parameter gradient and buffer"""
    from cenet_amd.networks.cenet.modules.cfam import CFAModule
    monkeypatch.setattr(kern, "_NO_CHANLOC", not fused)
    torch.manual_seed(5)
    m = CFAModule(dims, mca_rates=list(rates), init_value=0.5).to(dev)
    g = torch.Generator().manual_seed(9)
    with torch.no_grad():
        for n_, p_ in m.named_parameters():
            if p_.dim() == 1 or "layer_scale" in n_:
                p_.add_(0.1 * torch.randn(p_.shape, generator=g).to(dev))
    m.train()
    from cenet_amd import optim
    arena = optim.ParamArena(m)  # (parameters of equal role back to back: the merged-branch chains run, as in the benched model)
    x = torch.randn(B, dims, hw, hw, generator=g).to(dt).to(dev).requires_grad_(True)
    cot = torch.randn(B, dims, hw, hw, generator=g).to(dt).to(dev)
    old = kern.set_compute_bf16(dt == BF)
    try:
        arena.zero_grad()
        y = m(x)
        y.backward(cot)
        ops.wgrad_join()
    finally:
        kern.set_compute_bf16(old)
    out = {"y": y.detach(), "dx": x.grad}
    out.update({"g." + n_: p_.grad.clone() for n_, p_ in m.named_parameters()})
    out.update({"b." + n_: b_.clone() for n_, b_ in m.named_buffers()})
    return out


@pytest.mark.parametrize("dims,hw,B", [(32, 7, 3), (32, 14, 2), (48, 16, 5)])
@pytest.mark.parametrize("dt", [torch.float32, BF], ids=["f32", "bf16"])
def test_cfam_block_fused_chains_equal_the_launch_chains(dev, dt, dims, hw, B, monkeypatch):
    """the whole CFAM block with every channel-local chain on against the same block on the unfused launch chains"""
    calls = []
    fused_entries = ("cfam_mid_fwd", "cfam_mid_bwd", "dwbn_fwd", "dwbn_bwd", "cfam_front_fwd", "cfam_front_bwd", "dwact_fwd",
                     "dwact_bwd", "pool_branch_fwd", "pool_branch_bwd")
    for name in fused_entries:
        orig = getattr(kern, name)
        monkeypatch.setattr(kern, name, lambda *a, _o=orig, _n=name, **k: (calls.append(_n), _o(*a, **k))[1])
    ref = _run_cfam(dev, dt, dims, hw, B, False, monkeypatch)
    assert not calls
    got = _run_cfam(dev, dt, dims, hw, B, True, monkeypatch)
    assert set(calls) == set(fused_entries), calls
    # bf16 storage: both forms carry rounding noise of their own (the gradients pass ~40 bf16-stored tensors), so the fused form
    # is held to the fp32 evaluation of the launch chains and may be at most 2x (+ a floor of 2 % of the norm) as far from it as the unfused form
    ref32 = _run_cfam(dev, torch.float32, dims, hw, B, False, monkeypatch) if dt == BF else None
    for k in ref:
        if k.endswith(("conv_phi.bias", "conv_out.bias", "conv_g.bias")):
            continue  # mathematically zero (a key bias under a softmax, a bias in front of a BatchNorm): round-off noise only
        if ref[k].dtype in (torch.int64, torch.int32):
            assert torch.equal(ref[k], got[k]), k
            continue
        if dt == BF:
            r32 = ref32[k].float()
            nrm = r32.norm().item()
            floor = 5e-3 * r32.numel() ** 0.5 + 2e-2 * nrm
            e_f, e_u = (got[k].float() - r32).norm().item(), (ref[k].float() - r32).norm().item()
            assert e_f <= 2.0 * e_u + floor, (k, e_f, e_u, nrm)
        else:
            d = (got[k] - ref[k]).norm().item()
            assert d < 1e-4 * ref[k].norm().item() + 5e-5 * ref[k].numel() ** 0.5, (k, d, ref[k].norm().item())


@pytest.mark.parametrize("shape,dil", [((3, 8, 7, 7), 1), ((2, 5, 14, 14), 1), ((5, 3, 9, 6), 2), ((32, 2, 7, 7), 1)])
@pytest.mark.parametrize("dt", [torch.float32, BF], ids=["f32", "bf16"])
def test_dw_gelu_fused_equals_the_launch_chain(dev, shape, dil, dt, monkeypatch):
    """cfam.py:150-151 conv + bias + GELU: one launch per pass against conv / activation-backward / data- and weight-gradient"""
    B, Cn, H, W = shape
    g = torch.Generator().manual_seed(H * 10 + B)
    x0 = torch.randn(B, Cn, H, W, generator=g).to(dt).to(dev)
    cot = torch.randn(B, Cn, H, W, generator=g).to(dt).to(dev)
    res = []
    for fused in (False, True):
        monkeypatch.setattr(kern, "_NO_CHANLOC", not fused)
        x = x0.clone().requires_grad_(True)
        gg = torch.Generator().manual_seed(7)
        w = (0.3 * torch.randn(Cn, 1, 3, 3, generator=gg)).to(dev).requires_grad_(True)
        b = (0.2 * torch.randn(Cn, generator=gg)).to(dev).requires_grad_(True)
        w.grad, b.grad = torch.zeros_like(w), torch.zeros_like(b)
        y = ops.dwconv_nchw(x, w, b, dil=dil, act="gelu")
        assert isinstance(y.grad_fn, torch.autograd.function.BackwardCFunction) and (type(y.grad_fn).__name__.startswith("DWAct") == fused)
        y.backward(cot)
        ops.wgrad_join()
        res.append((y.detach(), x.grad, w.grad, b.grad))
    tol = 2e-2 if dt == BF else 2e-5
    for k, name in enumerate(("y", "dx", "dw", "db")):
        assert _rel(res[1][k], res[0][k]) < tol, (name, _rel(res[1][k], res[0][k]))


@pytest.mark.parametrize("shape", [(3, 10, 7, 7), (2, 12, 14, 14), (2, 8, 24, 40), (32, 6, 7, 7)])
@pytest.mark.parametrize("dt", [torch.float32, BF], ids=["f32", "bf16"])
def test_srm_fused_tail_equals_the_launch_chain(dev, shape, dt, monkeypatch):
    """cfam.py:93-101: conv + GELU + BatchNorm(1) + gate with the statistics folded by the gate kernel (3 + 4 launches) against the
    6 + 6 launches of the chain; running statistics and the batch counter included"""
    from cenet_amd.networks.cenet.modules.cfam import SRM
    B, Cn, H, W = shape
    g = torch.Generator().manual_seed(H + B)
    x0 = torch.randn(B, Cn, H, W, generator=g).to(dt).to(dev)
    cot = torch.randn(B, Cn, H, W, generator=g).to(dt).to(dev)
    res = []
    for fused in (False, True):
        monkeypatch.setattr(kern, "_NO_CHANLOC", not fused)
        torch.manual_seed(1)
        m = SRM().to(dev).train()
        with torch.no_grad():
            m.bn.weight.fill_(1.3), m.bn.bias.fill_(-0.2)
        x = x0.clone().requires_grad_(True)
        for p_ in m.parameters():
            p_.grad = torch.zeros_like(p_)
        calls = []
        orig = kern.srm_conv_gelu_fwd
        monkeypatch.setattr(kern, "srm_conv_gelu_fwd", lambda *a, _o=orig, **k: (calls.append(1), _o(*a, **k))[1])
        y = m(x)
        assert bool(calls) == fused
        y.backward(cot)
        ops.wgrad_join()
        res.append([y.detach(), x.grad] + [p_.grad for p_ in m.parameters()] + [m.bn.running_mean.clone(), m.bn.running_var.clone()])
        assert int(m.bn.num_batches_tracked) == 1
        monkeypatch.setattr(kern, "srm_conv_gelu_fwd", orig)
    tol = 2e-2 if dt == BF else 1e-4
    for a, b in zip(res[1], res[0]):
        assert (a.float() - b.float()).norm().item() <= tol * b.float().norm().item() + 1e-6, (a.shape, _rel(a, b))
