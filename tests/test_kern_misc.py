"""Kernel-level checks of the bandwidth-bound helpers (16-byte and scalar forms) against plain PyTorch fp32:
LayerNorm fwd/bwd, column sums, BatchNorm statistics / apply / backward, depthwise-conv weight gradients."""
import pytest
import torch
import torch.nn.functional as F

from backend import dev  # noqa: F401
from cenet_amd import kern


# C = 50 exercises the scalar fallback; 320 and 512 the two-pass form; rows not a multiple of the workgroup step
@pytest.mark.parametrize("rows,C", [(37, 64), (70, 128), (33, 256), (45, 320), (19, 512), (21, 50)])
def test_layernorm_fwd_bwd(dev, rows, C):
    g = torch.Generator().manual_seed(rows * 1000 + C)
    x = torch.randn(rows, C, generator=g) * 2 + 0.5
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
    dy = torch.randn(rows, C, generator=g)
    xr = x.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ref = F.layer_norm(xr, (C,), gr, br, 1e-6)
    ref.backward(dy)
    xd, gd, bd, dyd = (t.to(dev) for t in (x, gamma, beta, dy))
    y, mean, rstd = torch.empty_like(xd), torch.empty(rows, device=dev), torch.empty(rows, device=dev)
    kern.layernorm_fwd(xd, gd, bd, y, mean, rstd, rows, C, 1e-6)
    torch.testing.assert_close(y.cpu(), ref.detach(), rtol=1e-5, atol=2e-5)
    dx, dg, db = torch.empty_like(xd), torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    kern.layernorm_bwd(dyd, xd, gd, mean, rstd, dx, dg, db, rows, C)
    torch.testing.assert_close(dx.cpu(), xr.grad, rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(dg.cpu(), gr.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(db.cpu(), br.grad, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("R,C", [(300, 64), (1000, 128), (129, 320), (77, 2048), (513, 30), (5, 4)])
def test_col_sum_accumulates(dev, R, C):
    g = torch.Generator().manual_seed(R + C)
    a = torch.randn(R, C, generator=g)
    out0 = torch.randn(C, generator=g)
    out = out0.clone().to(dev)
    kern.col_sum(a.to(dev), out, R, C)
    torch.testing.assert_close(out.cpu(), out0 + a.sum(0), rtol=1e-4, atol=1e-4)


# HW = 196 / 784: 16-byte kernels (64- and 256-thread plane workgroups); HW = 49: scalar fallback; Ctot > C: the input is a
# channel slice of a wider tensor (batch stride != C*HW)
@pytest.mark.parametrize("B,C,HW,Ctot,act", [(3, 5, 196, 5, "none"), (2, 4, 784, 6, "relu"), (4, 3, 49, 3, "relu"),
                                             (2, 2, 2500, 2, "lrelu")])
def test_batchnorm_train_fwd_bwd(dev, B, C, HW, Ctot, act):
    g = torch.Generator().manual_seed(B * 100 + HW)
    xfull = torch.randn(B, Ctot, HW, generator=g) * 1.5 + 0.3
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    dy = torch.randn(B, C, HW, generator=g)
    slope = 0.2
    xr = xfull[:, :C].clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    pre = F.batch_norm(xr, None, None, gr, br, True, 0.1, 1e-5)
    ref = {"none": pre, "relu": F.relu(pre), "lrelu": F.leaky_relu(pre, slope)}[act]
    ref.backward(dy)
    xd = xfull.to(dev)
    sxb = Ctot * HW
    mean, var, ws = (torch.empty(n, device=dev) for n in (C, C, 2 * C * 256))
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    nbt = torch.zeros(1, dtype=torch.int64, device=dev)
    kern.bn_stats(xd, sxb, B, C, HW, ws, mean, var, rm, rv, 0.1, nbt)
    torch.testing.assert_close(mean.cpu(), xr.detach().transpose(0, 1).reshape(C, -1).mean(1), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(var.cpu(), xr.detach().transpose(0, 1).reshape(C, -1).var(1, unbiased=False), rtol=1e-4, atol=1e-5)
    assert int(nbt.item()) == 1
    y = torch.empty(B, C, HW, device=dev)
    kern.bn_apply(xd, sxb, y, C * HW, mean, var, 1e-5, gamma.to(dev), beta.to(dev), act, slope, B, C, HW)
    torch.testing.assert_close(y.cpu(), ref.detach(), rtol=1e-4, atol=1e-5)
    dx = torch.empty(B, C, HW, device=dev)
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    kern.bn_bwd(dy.to(dev), C * HW, xd, sxb, dx, C * HW, mean, var, 1e-5, gamma.to(dev), beta.to(dev), act, slope, B, C, HW,
                ws, dg, db)
    torch.testing.assert_close(dx.cpu(), xr.grad, rtol=1e-3, atol=2e-5)
    torch.testing.assert_close(dg.cpu(), gr.grad, rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(db.cpu(), br.grad, rtol=1e-3, atol=1e-4)


def _dw_ref(x, w, b, dil):
    return F.conv2d(x, w.view(-1, 1, 3, 3), b, padding=dil, dilation=dil, groups=x.shape[1])


# W % 4 == 0: 16-byte NCHW kernels (rows of 8 / 12 pixels, dilation 1..3 reaching past both row ends); W = 7: scalar form
@pytest.mark.parametrize("B,C,H,W,dil", [(2, 3, 6, 8, 1), (1, 2, 9, 12, 3), (2, 2, 5, 8, 2), (2, 3, 7, 7, 1)])
def test_dwconv_nchw_fwd_dgrad_wgrad(dev, B, C, H, W, dil):
    g = torch.Generator().manual_seed(B + C + H + W + dil)
    x = torch.randn(B, C, H, W, generator=g)
    w, b = torch.randn(C, 9, generator=g) * 0.3, torch.randn(C, generator=g)
    dy = torch.randn(B, C, H, W, generator=g)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    pre = _dw_ref(xr, wr, br, dil)
    F.gelu(pre).backward(dy)
    xd, wd, bd = x.to(dev), w.to(dev), b.to(dev)
    y, a = torch.empty_like(xd), torch.empty_like(xd)
    kern.dw_nchw(xd, C * H * W, wd, bd, y, C * H * W, a, C * H * W, B, C, H, W, dil, False, act="gelu")
    torch.testing.assert_close(y.cpu(), pre.detach(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(a.cpu(), F.gelu(pre.detach()), rtol=1e-4, atol=1e-5)
    # gradient w.r.t. the pre-activation, then the data / weight gradients of the convolution itself
    p2 = pre.detach().clone().requires_grad_(True)
    F.gelu(p2).backward(dy)
    gp = p2.grad
    dx = torch.empty_like(xd)
    kern.dw_nchw(gp.to(dev), C * H * W, wd, None, dx, C * H * W, None, 0, B, C, H, W, dil, True)
    torch.testing.assert_close(dx.cpu(), xr.grad, rtol=1e-4, atol=1e-5)
    dw, db = torch.zeros(C, 9, device=dev), torch.zeros(C, device=dev)
    kern.dw_wgrad_nchw(xd, C * H * W, gp.to(dev), C * H * W, dw, db, B, C, H, W, dil)
    torch.testing.assert_close(dw.cpu(), wr.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(db.cpu(), br.grad, rtol=1e-4, atol=1e-4)


# C % 4 == 0: 16-byte token-layout kernels (C = 8: one partly filled wave; C = 260: two workgroup slabs); C = 6: scalar form
# H = 30: the four-rows-per-thread forward (H >= 28), ragged in both directions
@pytest.mark.parametrize("B,C,H,W", [(2, 8, 5, 9), (1, 260, 3, 10), (2, 6, 4, 5), (1, 8, 30, 11)])
def test_dwconv_tok_fwd_dgrad_wgrad(dev, B, C, H, W):
    g = torch.Generator().manual_seed(B + C + H + W)
    xt = torch.randn(B, H * W, C, generator=g)
    w, b = torch.randn(C, 9, generator=g) * 0.3, torch.randn(C, generator=g)
    dyt = torch.randn(B, H * W, C, generator=g)
    to_nchw = lambda t: t.transpose(1, 2).reshape(B, C, H, W)
    to_tok = lambda t: t.reshape(B, C, H * W).transpose(1, 2)
    xr, wr, br = to_nchw(xt).clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    pre = _dw_ref(xr, wr, br, 1)
    pre.backward(to_nchw(dyt))
    xd, wd, bd, gd = xt.to(dev), w.to(dev), b.to(dev), dyt.contiguous().to(dev)
    y, a = torch.empty_like(xd), torch.empty_like(xd)
    kern.dw_tok(xd, wd, bd, y, a, B, C, H, W, False, act="gelu")
    torch.testing.assert_close(y.cpu(), to_tok(pre.detach()), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(a.cpu(), F.gelu(to_tok(pre.detach())), rtol=1e-4, atol=1e-5)
    dx = torch.empty_like(xd)
    kern.dw_tok(gd, wd, None, dx, None, B, C, H, W, True)
    torch.testing.assert_close(dx.cpu(), to_tok(xr.grad), rtol=1e-4, atol=1e-5)
    dw, db = torch.zeros(C, 9, device=dev), torch.zeros(C, device=dev)
    kern.dw_wgrad_tok(xd, gd, dw, db, B, C, H, W)
    torch.testing.assert_close(dw.cpu(), wr.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(db.cpu(), br.grad, rtol=1e-4, atol=1e-4)


# up x2 / x4, down x0.5, odd target sizes, both corner conventions: forward and the gather-form backward
@pytest.mark.parametrize("Hi,Wi,kw", [(6, 7, dict(scale_factor=2.0)), (5, 4, dict(scale_factor=4.0)),
                                      (8, 10, dict(scale_factor=0.5)), (7, 5, dict(size=(11, 13))),
                                      (6, 6, dict(size=(12, 12), align_corners=True)), (9, 8, dict(size=(4, 3), align_corners=True)),
                                      (3, 3, dict(size=(1, 1)))])
def test_bilinear_fwd_bwd(dev, Hi, Wi, kw):
    from cenet_amd import ops
    g = torch.Generator().manual_seed(Hi * 10 + Wi)
    x = torch.randn(2, 3, Hi, Wi, generator=g)
    xr = x.clone().requires_grad_(True)
    ref = F.interpolate(xr, mode="bilinear", **kw)
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)
    xd = x.to(dev).requires_grad_(True)
    y = ops.interpolate_bilinear(xd, size=kw.get("size"), scale_factor=kw.get("scale_factor"),
                                 align_corners=kw.get("align_corners", False))
    y.backward(dy.to(dev))
    torch.testing.assert_close(y.detach().cpu(), ref.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(xd.grad.cpu(), xr.grad, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("B,C,HW,with_b", [(3, 5, 196, True), (2, 4, 1028, False), (4, 3, 49, True)])
def test_chan_dot_accumulates(dev, B, C, HW, with_b):
    g = torch.Generator().manual_seed(B * C + HW)
    a, b = torch.randn(B, C, HW, generator=g), torch.randn(B, C, HW, generator=g)
    out0 = torch.randn(C, generator=g)
    out = out0.clone().to(dev)
    kern.chan_dot(a.to(dev), C * HW, b.to(dev) if with_b else None, C * HW if with_b else 0, out, B, C, HW)
    ref = out0 + ((a * b) if with_b else a).sum((0, 2))
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-4, atol=1e-4)


def test_zero_std_planes_give_finite_gradients(dev):
    """A constant plane / constant pixel has std == 0; aten::std_backward masks the 0/0 to 0, and so must the CCU and SRM
    backward kernels (a whole-model training step at 32x32 — 1x1 maps in the last decoder stage — turned every backbone
    gradient into NaN before they did)."""
    from cenet_amd.networks.cenet.modules.cfam import CCU, SRM
    from oracle import cenet_oracle as O
    g = torch.Generator().manual_seed(9)
    torch.manual_seed(9)  # module initialisation
    # CCU: per-(sample, channel) statistics over the plane; sample 0 / channel 1 is constant, and a 1x1 map is all-constant
    for shape in [(2, 4, 3, 3), (2, 4, 1, 1)]:
        x = torch.randn(*shape, generator=g)
        if shape[2] > 1:
            x[0, 1] = 0.75  # exactly representable mean: std is exactly 0 in both implementations
        mod = CCU(shape[1]).train()
        sd = {"m." + k: v.detach().clone() for k, v in mod.state_dict().items()}
        xr = x.clone().requires_grad_(True)
        ref = O.ccu(sd, "m", xr, True)
        go = torch.randn(ref.shape, generator=g)
        ref.backward(go)
        xd = x.to(dev).requires_grad_(True)
        out = mod.to(dev)(xd)
        out.backward(go.to(dev))
        assert torch.isfinite(xd.grad).all()
        torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(xd.grad.cpu(), xr.grad, rtol=2e-3, atol=2e-4)
    # SRM: per-pixel statistics over channels; one pixel has identical channels
    x = torch.randn(2, 5, 4, 4, generator=g)
    x[1, :, 2, 3] = -0.25
    mod = SRM().train()
    sd = {"m." + k: v.detach().clone() for k, v in mod.state_dict().items()}
    xr = x.clone().requires_grad_(True)
    ref = O.srm(sd, "m", xr, True)
    go = torch.randn(ref.shape, generator=g)
    ref.backward(go)
    xd = x.to(dev).requires_grad_(True)
    out = mod.to(dev)(xd)
    out.backward(go.to(dev))
    assert torch.isfinite(xd.grad).all()
    torch.testing.assert_close(xd.grad.cpu(), xr.grad, rtol=2e-3, atol=2e-4)


# kernel == stride spatial-reduction conv on a token map (pvtv2.py:93-95): patch gather + dense GEMM against F.conv2d,
# forward, data gradient (inverse scatter) and weight / bias gradients; s = 2, 4, 8; C not a multiple of 64; Cout ragged
@pytest.mark.parametrize("B,C,Cout,H,W,s", [(2, 8, 12, 4, 6, 2), (1, 20, 8, 8, 4, 4), (2, 4, 6, 8, 16, 8), (3, 36, 36, 6, 6, 2)])
def test_sr_conv_tok_patch_path(dev, B, C, Cout, H, W, s):
    from cenet_amd import ops
    g = torch.Generator().manual_seed(B + C + H + s)
    xt = torch.randn(B, H * W, C, generator=g)
    w = torch.randn(Cout, C, s, s, generator=g) * 0.2
    b = torch.randn(Cout, generator=g)
    Ho, Wo = H // s, W // s
    dy = torch.randn(B, Ho * Wo, Cout, generator=g)
    xr, wr, br = xt.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.conv2d(xr.transpose(1, 2).reshape(B, C, H, W), wr, br, stride=s).reshape(B, Cout, Ho * Wo).transpose(1, 2)
    ref.backward(dy)
    # the gather itself, both directions: a bijection between the token map and the patch rows
    xd = xt.to(dev)
    xp = torch.empty(B, Ho * Wo, C * s * s, device=dev)
    kern.patch_tok(xd, xp, B, Ho, Wo, C, s)
    want = xt.reshape(B, Ho, s, Wo, s, C).permute(0, 1, 3, 5, 2, 4).reshape(B, Ho * Wo, C * s * s)
    assert torch.equal(xp.cpu(), want)
    back = torch.empty_like(xd)
    kern.patch_tok(xp, back, B, Ho, Wo, C, s, inverse=True)
    assert torch.equal(back.cpu(), xt)
    # the op as the model calls it
    xq, wq, bq = xt.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    y = ops.conv2d_tok(xq, H, W, wq, bq, stride=s, pad=0, out_layout="tok")
    torch.testing.assert_close(y.detach().cpu(), ref.detach(), rtol=1e-4, atol=1e-4)
    y.backward(dy.to(dev))
    torch.testing.assert_close(xq.grad.cpu(), xr.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(wq.grad.cpu(), wr.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(bq.grad.cpu(), br.grad, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("n,off", [(7, 0), (9, 1), (1, 0), (1031, 3), (4096, 0)])
def test_zero_fill_of_odd_sized_bf16_buffers(dev, n, off):
    """kern.zero_ on bf16 tensors with an odd element count / starting on an odd element (a channel-split branch without a
    consumer, a non-overlapping strided-conv gradient): whole words by the grid, head / tail bytes by one workgroup, nothing
    outside the range touched"""
    buf = torch.full((n + off + 5,), 3.0, dtype=torch.bfloat16, device=dev)
    view = buf[off:off + n]
    assert view.is_contiguous()
    kern.zero_(view)
    out = buf.float().cpu()
    assert torch.all(out[off:off + n] == 0)
    assert torch.all(out[:off] == 3.0) and torch.all(out[off + n:] == 3.0)


def test_dpp_wave_reductions(dev):
    """common.h wave_sum_dpp / wave_max_dpp / wave_min_i_dpp (round 5: the wave reductions of every kernel): against numpy on
    random, all-negative and single-outlier waves; on the host checker the same entry runs the shuffle forms."""
    import ctypes as C
    import numpy as np
    from cenet_amd import _lib, kern
    g = torch.Generator().manual_seed(4)
    nw = 37
    x = torch.randn(nw, 64, generator=g)
    x[1] = -x[1].abs() - 1.0          # all negative: a zero-filled (bound_ctrl) DPP source would corrupt the maximum
    x[2] = 0.0
    x[2, 63] = 5.0                    # the result lane itself
    x[3] = 0.0
    x[3, 0] = -7.0
    xd = x.to(dev)
    sums, maxs = torch.empty(2 * nw).to(dev), torch.empty(2 * nw).to(dev)
    mins = torch.empty(2 * nw, dtype=torch.int32).to(dev)
    rc = _lib.lib().cenet_selftest_wave_reduce(kern.P(xd), kern.P(sums), kern.P(maxs), kern.P(mins), C.c_int(nw), kern.stream())
    assert rc == 0
    if dev.type != "cpu":
        torch.cuda.synchronize()
    np.testing.assert_allclose(sums.cpu().numpy(), x.double().sum(1).repeat(2).numpy(), rtol=1e-5, atol=1e-5)
    assert torch.equal(sums[:nw], sums[nw:])  # (lane 0 and lane 37 hold the same bits)
    assert torch.equal(maxs.cpu(), x.max(1).values.repeat(2))
    want = ((x * 1024.0).to(torch.int32) + torch.arange(64, dtype=torch.int32)).min(1).values
    assert torch.equal(mins.cpu(), want.repeat(2))
