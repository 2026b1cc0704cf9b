"""The fused PVT-MLP kernels (csrc/pvt_mlp.hip) against the chain of launches they replace (LayerNorm -> fc1 -> depthwise
3x3 + GELU -> fc2 + residual with the DropPath scale), on the host SIMT checker and on the GPU."""
import pytest
import torch

from backend import dev  # noqa: F401
from cenet_amd import kern, ops

BF = torch.bfloat16


def _params(C, HD, dev, seed=0):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc)
    p = dict(ln_g=1.0 + 0.2 * r(C), ln_b=0.1 * r(C), w1=r(HD, C, sc=C ** -0.5), b1=0.1 * r(HD), wd=r(HD, 1, 3, 3, sc=0.4),
             bd=0.1 * r(HD), w2=r(C, HD, sc=HD ** -0.5), b2=0.1 * r(C))
    return {k: v.to(dev).requires_grad_(True) for k, v in p.items()}


def _chain(x, p, H, W, bscale):
    y, xr = ops.layernorm_res(x, p["ln_g"], p["ln_b"], 1e-6)
    h = ops.linear(y, p["w1"], p["b1"])
    h = ops.dwconv_tok(h, p["wd"], p["bd"], H, W, act="gelu")
    return ops.linear(h, p["w2"], p["b2"], resid=xr, bscale=bscale)


@pytest.mark.parametrize("C,HD,H,W,B,drop", [(64, 128, 16, 14, 2, True), (64, 64, 7, 14, 1, False), (128, 128, 14, 28, 2, True),
                                              (128, 192, 8, 14, 1, False)])
def test_fused_forward_equals_chain(dev, C, HD, H, W, B, drop):
    p = _params(C, HD, dev)
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(B, H * W, C, generator=g) * 1.5).to(BF).to(dev)
    bscale = torch.tensor([0.0, 1.25][:B] if B > 1 else [1.25], dtype=torch.float32, device=dev) if drop else None
    assert kern.pvt_mlp_supported(x, C, HD, H, W)
    with torch.no_grad():
        ref = _chain(x, p, H, W, bscale)
        y = torch.empty_like(x)
        saved = (torch.empty_like(x), torch.empty(B * H * W, device=dev), torch.empty(B * H * W, device=dev),
                 torch.empty(B, H * W, HD, dtype=BF, device=dev), torch.empty(B, H * W, HD, dtype=BF, device=dev))
        kern.pvt_mlp_fwd(x, p["ln_g"], p["ln_b"], 1e-6, kern.wq(p["w1"], x), p["b1"], p["wd"], p["bd"], kern.wq(p["w2"], x),
                         p["b2"], bscale, y, B, H, W, C, HD, saved)
        y2 = torch.empty_like(x)  # inference form: nothing saved, same result
        kern.pvt_mlp_fwd(x, p["ln_g"], p["ln_b"], 1e-6, kern.wq(p["w1"], x), p["b1"], p["wd"], p["bd"], kern.wq(p["w2"], x),
                         p["b2"], bscale, y2, B, H, W, C, HD)
        assert torch.equal(y, y2)
        # the saved tensors against the chain's own
        xn_ref = torch.empty_like(x)
        mean_ref, rstd_ref = torch.empty(B * H * W, device=dev), torch.empty(B * H * W, device=dev)
        kern.layernorm_fwd(x, p["ln_g"], p["ln_b"], xn_ref, mean_ref, rstd_ref, B * H * W, C, 1e-6)
        h_ref = ops.linear(xn_ref, p["w1"], p["b1"])
        a_ref = ops.dwconv_tok(h_ref, p["wd"], p["bd"], H, W, act="gelu")
        if bscale is not None:  # the saved GELU output carries the DropPath scale (it only feeds the fc2 weight gradient)
            a_ref = (a_ref.float() * bscale.view(-1, 1, 1)).to(BF)
    for got, want, name in ((saved[0], xn_ref, "xn"), (saved[1], mean_ref, "mean"), (saved[2], rstd_ref, "rstd"),
                            (saved[3], h_ref, "h"), (saved[4], a_ref, "a")):
        dd = (got.float() - want.float()).abs()
        assert dd.max().item() <= 0.04 * max(want.float().abs().max().item(), 1e-6), (name, dd.max().item())
        # (a with a DropPath scale: the kernel scales in fp32 and rounds once, the restatement above rounds twice)
        rel = 3e-3 if (name == "a" and bscale is not None) else 1e-3
        assert dd.mean().item() <= rel * max(want.float().abs().mean().item(), 1e-6), (name, dd.mean().item())
    with torch.no_grad():
        pass
    d = (y.float() - ref.float()).abs()
    # identical rounding points; only the order of fp32 additions differs -> a few bf16 ulps on isolated elements
    assert d.max().item() <= 0.05 * ref.float().abs().max().item(), d.max().item()
    assert d.mean().item() <= 2e-3 * ref.float().abs().mean().item(), (d.mean().item(), ref.float().abs().mean().item())


def _grads(fn, x, p, gy):
    xg = x.clone().requires_grad_(True)
    for v in p.values():
        v.grad = None
    fn(xg).backward(gy)
    ops.wgrad_flush()
    return [xg.grad.float()] + [p[k].grad.clone().float() for k in ("ln_g", "ln_b", "w1", "b1", "wd", "bd", "w2", "b2")]


@pytest.mark.parametrize("C,HD,H,W,B,drop", [(64, 128, 16, 14, 2, True), (64, 64, 7, 14, 1, False), (128, 128, 14, 28, 2, True),
                                              (128, 192, 8, 14, 1, False)])
def test_fused_backward_equals_chain(dev, C, HD, H, W, B, drop):
    p = _params(C, HD, dev)
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(B, H * W, C, generator=g) * 1.5).to(BF).to(dev)
    gy = torch.randn(B, H * W, C, generator=g).to(BF).to(dev)
    bscale = torch.tensor([0.0, 1.25][:B] if B > 1 else [1.25], dtype=torch.float32, device=dev) if drop else None
    ref = _grads(lambda t: _chain(t, p, H, W, bscale), x, p, gy)
    got = _grads(lambda t: ops.pvt_mlp(t, H, W, p["ln_g"], p["ln_b"], 1e-6, p["w1"], p["b1"], p["wd"], p["bd"], p["w2"], p["b2"],
                                       bscale), x, p, gy)
    names = ["dx", "ln_g", "ln_b", "w1", "b1", "wd", "bd", "w2", "b2"]
    for n, r, o in zip(names, ref, got):
        assert o.shape == r.shape, n
        err = (o - r).norm().item() / max(r.norm().item(), 1e-12)
        assert err < 1e-2, (n, err)
