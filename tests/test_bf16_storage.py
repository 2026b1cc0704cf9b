"""Throughput-mode storage: every operator of cenet_amd.ops run on bf16 tensors (the `_bf16` twins of the C ABI) against the
SAME operator on fp32 tensors holding the same bf16-rounded values (the parity-mode path, itself pinned on the reference
goldens).  Forward outputs, input gradients and parameter gradients must agree to bf16 rounding of the results (a few
2^-8 of the tensor's range; the parameter gradients are fp32 sums in both modes).  Runs on the host SIMT checker (CPU) and,
with -m gpu, through libcenet_hip.so.  Shapes cover the vector (multiples of 4 / 8) and the scalar fallback forms."""
import pytest
import torch

from backend import dev  # noqa: F401
from cenet_amd import kern, ops

BF = torch.bfloat16


def _r(t):
    return t.bfloat16().float()


def _rel(a, b):
    return (a.float() - b.float()).abs().max().item() / (b.float().abs().max().item() + 1e-12)


def compare(fn, acts, params=(), tol=2.5e-2, ptol=3e-2, dev="cpu", out_index=None, extra=()):
    """fn(*acts, *params, *extra) -> tensor (or tuple; out_index picks).  acts: fp32 activation tensors (rounded to bf16
    values here); params: fp32 parameter tensors (kept fp32 in both runs)."""
    acts = [_r(a) for a in acts]
    res = {}
    for mode in ("f32", "bf16"):
        xs = [(a.detach().clone().to(dev).to(BF) if mode == "bf16" else a.detach().clone().to(dev)).requires_grad_(True)
              for a in acts]
        ps = [torch.nn.Parameter(p.clone().to(dev)) for p in params]
        y = fn(*xs, *ps, *extra)
        if out_index is not None:
            y = y[out_index]
        if mode == "f32":
            go = _r(torch.randn(y.shape, generator=torch.Generator().manual_seed(11)))
        assert y.dtype == (BF if mode == "bf16" else torch.float32), (mode, y.dtype)
        y.backward(go.to(dev).to(y.dtype))
        res[mode] = (y.detach().float().cpu(), [x.grad.float().cpu() if x.grad is not None else None for x in xs],
                     [p.grad.float().cpu() if p.grad is not None else None for p in ps])
    (y0, gx0, gp0), (y1, gx1, gp1) = res["f32"], res["bf16"]
    assert _rel(y1, y0) < tol, ("output", _rel(y1, y0))
    for i, (a, b) in enumerate(zip(gx1, gx0)):
        if b is not None:
            assert a is not None and a.dtype == torch.float32
            assert _rel(a, b) < tol * 1.6, ("input grad", i, _rel(a, b))
    for i, (a, b) in enumerate(zip(gp1, gp0)):
        if b is not None:
            assert _rel(a, b) < ptol, ("param grad", i, _rel(a, b))


def G(seed):
    return torch.Generator().manual_seed(seed)


@pytest.mark.parametrize("R,K,N,bias,resid,bscale", [(70, 64, 40, True, True, False), (24, 50, 33, True, False, False),
                                                     (2 * 30, 128, 64, True, True, True), (150, 192, 70, False, False, False)])
def test_linear(dev, R, K, N, bias, resid, bscale):
    g = G(R + K)
    B = 2
    x = torch.randn(B, R // B, K, generator=g)
    W, b = torch.randn(N, K, generator=g) * 0.1, torch.randn(N, generator=g)
    r = torch.randn(B, R // B, N, generator=g)
    bs = torch.tensor([0.0, 1.25]).to(dev) if bscale else None

    def fn(x, r, W, b):
        return ops.linear(x, W, b if bias else None, resid=r if resid else None, bscale=bs)
    compare(fn, [x, r], [W, b], dev=dev)


@pytest.mark.parametrize("B,Cin,Cout,H,W", [(2, 40, 24, 8, 8), (1, 64, 130, 7, 7), (2, 136, 72, 10, 10)])
def test_conv1x1(dev, B, Cin, Cout, H, W):
    g = G(Cin + Cout)
    x, r = torch.randn(B, Cin, H, W, generator=g), torch.randn(B, Cout, H, W, generator=g)
    Wt, b = torch.randn(Cout, Cin, 1, 1, generator=g) * 0.1, torch.randn(Cout, generator=g)
    compare(lambda x, r, Wt, b: ops.conv1x1(x, Wt, b, resid=r), [x, r], [Wt, b], dev=dev)


# grouped pointwise convs (ops.grouped_conv1x1: the three dilated SepConvBN branches in one launch): G = 20 / 40 channels
# per group take the thread-per-pixel kernel (conv_c1.hip, pw_small_kernel) with bf16 tensors, 100 the batched ring GEMM;
# even and odd plane sizes.  The fp32 run (batched GEMM) is itself held against F.conv2d(groups=3).
@pytest.mark.parametrize("Gc,H,W", [(20, 8, 8), (40, 7, 7), (20, 5, 6), (100, 7, 7), (32, 6, 6)])
def test_grouped_conv1x1(dev, Gc, H, W):
    g = G(Gc + H)
    x = torch.randn(2, 3 * Gc, H, W, generator=g)
    Wt = torch.randn(3, Gc, Gc, 1, 1, generator=g) * 0.2
    compare(lambda x, Wt: ops.grouped_conv1x1(x, Wt), [x], [Wt], dev=dev)
    xr = _r(x)
    y = ops.grouped_conv1x1(xr.to(dev), Wt.to(dev)).cpu()
    ref = torch.nn.functional.conv2d(xr, Wt.reshape(3 * Gc, Gc, 1, 1), groups=3)
    assert _rel(y, ref) < 1e-5


# square bias-free 1x1 convs over a few channels (the pooled branch of MultiOrderDWConv at 7x7): pw_small_kernel with bf16
@pytest.mark.parametrize("Cn,H,W", [(4, 7, 7), (8, 7, 7), (20, 7, 7), (32, 4, 4)])
def test_conv1x1_few_channels(dev, Cn, H, W):
    g = G(Cn)
    x = torch.randn(3, Cn, H, W, generator=g)
    Wt = torch.randn(Cn, Cn, 1, 1, generator=g) * 0.3
    compare(lambda x, Wt: ops.conv1x1(x, Wt), [x], [Wt], dev=dev)


# 64 -> num_classes with bias (the head's last layer): pw_fewout kernels with bf16 tensors; 4 / 9 / 2 classes have the
# dedicated weight-gradient kernel, 3 takes the GEMM one; odd plane size
@pytest.mark.parametrize("Co,H,W", [(4, 12, 12), (9, 10, 14), (2, 40, 40), (3, 8, 8), (4, 7, 7)])
def test_conv1x1_few_outputs(dev, Co, H, W):
    g = G(Co)
    x = torch.randn(2, 64, H, W, generator=g)
    Wt, b = torch.randn(Co, 64, 1, 1, generator=g) * 0.2, torch.randn(Co, generator=g)
    compare(lambda x, Wt, b: ops.conv1x1(x, Wt, b), [x], [Wt, b], dev=dev)


def test_conv1x1_one_channel_input(dev):
    """the shortcut 1x1 convolution of the one-channel network input (unet.py conv3): stencil kernels in bf16"""
    g = G(3)
    x = torch.randn(2, 1, 9, 140, generator=g)
    Wt = torch.randn(32, 1, 1, 1, generator=g) * 0.5
    compare(lambda x, Wt: ops.conv1x1(x, Wt), [x], [Wt], dev=dev)


@pytest.mark.parametrize("Cin,Cout,k,s,H,expand,layout", [(1, 16, 7, 4, 32, 3, "tok"), (16, 24, 3, 2, 16, 0, "tok"),
                                                          (8, 12, 3, 1, 9, 0, "nchw"), (4, 6, 1, 1, 7, 0, "nchw")])
def test_conv2d_implicit_gemm(dev, Cin, Cout, k, s, H, expand, layout):
    g = G(Cin * 7 + k)
    x = torch.randn(2, Cin, H, H, generator=g)
    Wt = torch.randn(Cout, expand or Cin, k, k, generator=g) * 0.1
    b = torch.randn(Cout, generator=g)
    compare(lambda x, Wt, b: ops.conv2d_nchw(x, Wt, b, stride=s, pad=k // 2, out_layout=layout, expand_channels=expand),
            [x], [Wt, b], dev=dev)


# the one-channel stencils of conv_c1.hip (bf16 only; fp32 takes the implicit GEMM): k = 5 / 3 / 1, full 128-column tiles with
# a ragged last one (W = 140), a width that is not a multiple of 4 (scalar stores), several row groups per workgroup.  The
# input needs no gradient in the network; here x does, which exercises the generic data-gradient path behind the special forward.
# W % 32 == 0 with k = 5: the matrix-core kernels (taps as the contraction / an output dimension); 160 columns = a full and a
# quarter tile, 19 rows = ragged row groups, 24 channels = a partly empty second channel fragment
@pytest.mark.parametrize("Cout,k,H,W", [(32, 5, 20, 140), (32, 1, 9, 132), (12, 3, 11, 30), (32, 5, 7, 9), (32, 5, 19, 160),
                                        (24, 5, 8, 32), (32, 5, 30, 224)])
def test_conv_one_channel_input(dev, Cout, k, H, W):
    g = G(Cout + k + W)
    x = torch.randn(2, 1, H, W, generator=g)
    Wt = torch.randn(Cout, 1, k, k, generator=g) * 0.2
    compare(lambda x, Wt: ops.conv2d_nchw(x, Wt, None, stride=1, pad=k // 2), [x], [Wt], dev=dev)


# 3x3 patch-embedding convs on tokens: bf16 materialises the overlapping patch rows (im2col_tok) and runs plain GEMMs, its
# data gradient is the gathering transpose; fp32 is the implicit GEMM.  Stride 2 (odd and even maps) and stride 1.
@pytest.mark.parametrize("B,C,Cout,H,W,s", [(2, 16, 24, 8, 8, 2), (1, 8, 12, 7, 9, 2), (1, 8, 16, 6, 6, 1)])
def test_conv3x3_tok_overlapping_patches(dev, B, C, Cout, H, W, s):
    g = G(C + H + s)
    x = torch.randn(B, H * W, C, generator=g)
    Wt, b = torch.randn(Cout, C, 3, 3, generator=g) * 0.1, torch.randn(Cout, generator=g)
    compare(lambda x, Wt, b: ops.conv2d_tok(x, H, W, Wt, b, stride=s, pad=1, out_layout="tok"), [x], [Wt, b], dev=dev)


@pytest.mark.parametrize("B,C,Cout,H,W,s", [(2, 16, 24, 8, 8, 2), (1, 64, 40, 8, 8, 4)])
def test_sr_conv_tok(dev, B, C, Cout, H, W, s):
    g = G(C + s)
    x = torch.randn(B, H * W, C, generator=g)
    Wt, b = torch.randn(Cout, C, s, s, generator=g) * 0.05, torch.randn(Cout, generator=g)
    compare(lambda x, Wt, b: ops.conv2d_tok(x, H, W, Wt, b, stride=s, pad=0, out_layout="tok"), [x], [Wt, b], dev=dev)


# (C % 8 == 0: the 16-byte backward kernel, one instantiation per width class 64 / 128 / 256 / 512; C = 50: quad-free path)
@pytest.mark.parametrize("rows,C", [(37, 64), (33, 320), (21, 50), (70, 128), (19, 512), (45, 200)])
def test_layernorm_and_residual_form(dev, rows, C):
    g = G(rows + C)
    x = torch.randn(rows, C, generator=g) * 2 + 0.5
    gm, bt = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    compare(lambda x, gm, bt: ops.layernorm(x, gm, bt, 1e-6), [x], [gm, bt], dev=dev)

    def res(x, gm, bt):
        y, xr = ops.layernorm_res(x, gm, bt, 1e-6)
        return y * 1.0 + xr * 0.5  # torch elementwise glue on the host checker / GPU: both outputs carry gradient
    compare(res, [x], [gm, bt], dev=dev)


# bf16: channels of up to 2048 elements (B*H*W) run the one-kernel forms; beyond that planes of up to 1024 pixels take the flat
# small-plane kernels (12 x 28x28, 12 x 7x9) and larger planes the plane-per-workgroup ones (6 x 40x40)
@pytest.mark.parametrize("B,C,H,W,act", [(3, 5, 14, 14, "none"), (2, 4, 28, 28, "relu"), (4, 3, 7, 7, "lrelu"),
                                         (2, 4, 40, 40, "relu"), (2, 3, 9, 5, "none"), (12, 3, 28, 28, "relu"),
                                         (6, 2, 40, 40, "lrelu"), (140, 2, 7, 9, "none")])
def test_batchnorm(dev, B, C, H, W, act):
    g = G(B + C + H)
    x = torch.randn(B, C, H, W, generator=g) * 1.5 + 0.3
    gm, bt = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2

    def fn(x, gm, bt):
        rm, rv, nbt = torch.zeros(C, device=x.device), torch.ones(C, device=x.device), torch.zeros((), dtype=torch.long, device=x.device)
        return ops.batchnorm(x, gm, bt, rm, rv, nbt, True, 1e-5, act, 0.2, 0.1)
    compare(fn, [x], [gm, bt], dev=dev, tol=3e-2)


# C % 8 == 0 takes the LDS-tiled kernels (dwconv.hip): the three tile shapes (W % 14 == 0; maps up to 8x8; 8x16), ragged
# tiles, two channel slabs with a partial one (C = 136), several tiles per workgroup in the fused backward (28x28)
@pytest.mark.parametrize("B,C,H,W,act", [(2, 8, 6, 6, "gelu"), (1, 6, 30, 9, "none"), (1, 5, 5, 5, "gelu"),
                                         (2, 16, 7, 7, "gelu"), (1, 136, 14, 14, "gelu"), (1, 8, 11, 19, "gelu"),
                                         (1, 8, 28, 28, "gelu"), (1, 8, 9, 14, "none"),
                                         # C % 64 == 0 at 14 x 14 / 7 x 7 (PVT stages 3 / 4): whole-plane kernels, thread = channel
                                         # pair x image row (dw3x3_tok_plane_kernel: forward, activation backward + weight
                                         # gradient, data gradient); two slabs, two images
                                         (2, 128, 14, 14, "gelu"), (2, 64, 7, 7, "gelu"), (1, 192, 14, 14, "none")])
def test_dwconv_tok(dev, B, C, H, W, act):
    g = G(C + H)
    x = torch.randn(B, H * W, C, generator=g)
    w, b = torch.randn(C, 1, 3, 3, generator=g) * 0.3, torch.randn(C, generator=g) * 0.1
    compare(lambda x, w, b: ops.dwconv_tok(x, w, b, H, W, act), [x], [w, b], dev=dev)


# bf16 with H*W % 4 == 0 takes the plane-in-LDS kernels (several planes per workgroup; 14x14: quads straddle rows; dilation 5;
# 7x7 keeps the scalar kernels); the 2 x 70 x 8 x 8 case makes a workgroup's planes span two images
@pytest.mark.parametrize("B,C,H,W,dil,act", [(2, 3, 8, 8, 1, "gelu"), (1, 2, 12, 12, 2, "none"), (2, 3, 7, 7, 3, "none"),
                                             (2, 5, 14, 14, 1, "gelu"), (1, 3, 20, 28, 5, "none"), (2, 70, 8, 8, 2, "none"),
                                             # one plane per workgroup (owned quads; the weight gradient's workgroup reduction)
                                             (1, 2, 48, 48, 1, "gelu"), (2, 2, 56, 56, 2, "none"), (1, 3, 56, 56, 1, "none"),
                                             # batch >= 4 with W % 4 == 0: the weight gradient takes ONE CHANNEL ACROSS IMAGES per
                                             # workgroup (dw3x3_wgrad_nchw_chan_kernel): one pass of 5 planes; passes of 5 + 4
                                             # planes (28 x 28); one plane per pass over 4 images (56 x 56); dilation 2
                                             (5, 6, 8, 8, 1, "gelu"), (9, 3, 28, 28, 1, "none"), (4, 2, 56, 56, 1, "none"),
                                             (6, 4, 12, 12, 2, "none")])
def test_dwconv_nchw(dev, B, C, H, W, dil, act):
    g = G(C + H + dil)
    x = torch.randn(B, C, H, W, generator=g)
    w, b = torch.randn(C, 1, 3, 3, generator=g) * 0.3, torch.randn(C, generator=g) * 0.1
    compare(lambda x, w, b: ops.dwconv_nchw(x, w, b, dil=dil, act=act), [x], [w, b], dev=dev)


# (<= 64 keys with 64-dim heads: the fused backward kernel sra_bwd_kernel — 49 keys incl. a second, ragged key tile, 20 keys =
# one key tile, 64 keys, query counts with partial last tiles / idle waves, several tiles per wave)
@pytest.mark.parametrize("B,N,Nk,C,heads", [(2, 70, 49, 128, 2), (1, 130, 130, 64, 1), (1, 200, 20, 64, 1), (2, 49, 49, 192, 3),
                                            (1, 300, 64, 64, 1)])
def test_sr_attention(dev, B, N, Nk, C, heads):
    g = G(N)
    q, kv = torch.randn(B, N, C, generator=g), torch.randn(B, Nk, 2 * C, generator=g)
    compare(lambda q, kv: ops.sr_attention(q, kv, heads), [q, kv], dev=dev, tol=3e-2)


# C = 64 with N >= 256 takes the single-softmax form of the pair kernels (attn_diff.hip) on token-major copies; N = 300 has a
# ragged last key tile and a partial last query wave
# C = 128 (the 28x28 level): the same form with 64-wide halves
@pytest.mark.parametrize("B,C,N", [(2, 64, 100), (1, 32, 49), (2, 64, 256), (1, 64, 300), (1, 128, 260), (2, 128, 256)])
def test_nonlocal_attention(dev, B, C, N):
    g = G(C + N)
    th, ph, gx = (torch.randn(B, C, N, generator=g) for _ in range(3))
    compare(ops.nonlocal_attention, [th, ph, gx], dev=dev, tol=3e-2)


# hd 16 / 8: tiled kernels (two softmax heads add into one value head: fp32 accumulators); hd 80: materialised path (fp32 scores)
# (N % 4 == 0 and hd in {8, 16, 32, 64}: the pair kernels of attn_diff.hip, incl. ragged last tiles and the 64-query form at 1024+)
@pytest.mark.parametrize("B,N,H,hd", [(2, 96, 2, 16), (1, 70, 2, 8), (1, 49, 1, 80), (1, 132, 2, 16), (2, 72, 1, 8),
                                      (1, 100, 2, 32), (1, 1028, 1, 16), (1, 72, 2, 64), (2, 45, 1, 64)])
def test_diff_attention_heads_and_combine(dev, B, N, H, hd):
    g = G(N + hd)
    E = 2 * H * hd
    q, k, v = (torch.randn(B, N, E, generator=g) for _ in range(3))
    compare(lambda q, k, v: ops.diff_attention_heads(q, k, v, H), [q, k, v], dev=dev, tol=3e-2)
    U = torch.randn(B, 2 * H, N, 2 * hd, generator=g)
    lams = [torch.randn(hd, generator=g) * 0.1 for _ in range(4)]
    compare(lambda U, a, b, c, d: ops.diff_attention_combine(U, a, b, c, d, 0.5), [U], lams, dev=dev)


def test_layout_and_glue(dev):
    g = G(5)
    x = torch.randn(2, 6 * 6, 10, generator=g)
    compare(lambda x: ops.tok_to_nchw(x, 6, 6), [x], dev=dev, tol=1e-6)
    x7 = torch.randn(2, 7 * 7, 5, generator=g)  # odd extents: scalar transpose
    compare(lambda x: ops.tok_to_nchw(x, 7, 7), [x7], dev=dev, tol=1e-6)
    a, b = torch.randn(2, 3, 4, 4, generator=g), torch.randn(2, 5, 4, 4, generator=g)
    compare(ops.concat2, [a, b], dev=dev, tol=1e-6)
    # n-way concat in one launch (cenet_cat_channels): 16-byte, 8-byte and scalar part lengths, 2..4 parts, unequal channels
    for hw, cs in ((8, (8, 8, 8, 4)), (4, (5, 5, 5, 1)), (7, (3, 2)), (3, (1, 2, 3))):
        parts = [torch.randn(2, c, hw, hw, generator=g) for c in cs]
        compare(lambda *xs: ops.concat(list(xs)), parts, dev=dev, tol=1e-6)
    compare(lambda x: ops.split_channels(x, [2, 3])[1], [b], dev=dev, tol=1e-6)
    c = torch.randn(2, 3, 4, 4, generator=g)
    compare(lambda a, c: ops.add_act(a, c, "lrelu", 0.01), [a, c], dev=dev)
    compare(ops.silu_mul, [a, c], dev=dev)
    compare(lambda a, c, w: ops.mix(a, c, w), [a, c], [torch.tensor(0.4)], dev=dev)
    compare(lambda a, c, ls: ops.scale_residual(a, c, ls), [a, c], [torch.rand(1, 3, 1, 1, generator=g)], dev=dev)
    a7 = torch.randn(2, 3, 7, 7, generator=g)
    compare(lambda a, c, ls: ops.scale_residual(a, c, ls), [a7, a7 * 0.5 + 1], [torch.rand(1, 3, 1, 1, generator=g)], dev=dev)


# (exact x2 / x0.5 with align_corners=False take the fixed-tap backward kernels in bf16, strong up-sampling (x7, cfam.py:231-236;
# x4.5-5) the row-parallel one; the fp32 run is the general gather)
@pytest.mark.parametrize("kw", [dict(scale_factor=2, align_corners=True), dict(scale_factor=0.5, align_corners=False),
                                dict(size=(9, 11), align_corners=False), dict(scale_factor=2, align_corners=False),
                                dict(size=(16, 20), align_corners=False), dict(scale_factor=7, align_corners=True),
                                dict(size=(40, 45), align_corners=False)])
def test_resampling(dev, kw):
    g = G(9)
    x = torch.randn(2, 3, 8, 10, generator=g)
    compare(lambda x: ops.interpolate_bilinear(x, **kw), [x], dev=dev)
    compare(ops.nearest2x, [x], dev=dev)
    compare(lambda x: ops.adaptive_avgpool(x, 3, 3), [x], dev=dev)
    compare(lambda x, w: ops.maxpool2_scale(x, w), [x], [torch.rand(1, 3, 1, 1, generator=g) + 0.5], dev=dev)


# (C = 256 on a 7x7 map: the channel-split form of the SRM gate reduction, bf16 only)
@pytest.mark.parametrize("B,C,H,W", [(2, 6, 8, 8), (3, 4, 7, 7), (1, 4, 8, 8), (2, 256, 7, 7)])
def test_ccu_and_srm(dev, B, C, H, W):
    g = G(B + C + H)
    x = torch.randn(B, C, H, W, generator=g) + 0.2
    fc1, fc2 = torch.randn(3 * C, 1, 3, generator=g) * 0.5, torch.randn(C, 3, 1, generator=g) * 0.5
    bw, bb = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1

    def ccu(x, fc1, fc2, bw, bb):
        rm, rv, nbt = torch.zeros(C, device=x.device), torch.ones(C, device=x.device), torch.zeros((), dtype=torch.long, device=x.device)
        return ops.ccu(x, fc1, fc2, bw, bb, rm, rv, nbt, True)
    compare(ccu, [x], [fc1, fc2, bw, bb], dev=dev, tol=3e-2, ptol=6e-2)
    pwc, dwc = torch.randn(1, 3, 1, 1, generator=g) * 0.5, torch.randn(1, 3, 3, 3, generator=g) * 0.3
    sw, sb = torch.rand(1, generator=g) + 0.5, torch.randn(1, generator=g) * 0.1

    def srm(x, pwc, dwc, sw, sb):
        rm, rv, nbt = torch.zeros(1, device=x.device), torch.ones(1, device=x.device), torch.zeros((), dtype=torch.long, device=x.device)
        return ops.srm(x, pwc, dwc, sw, sb, rm, rv, nbt, True)
    compare(srm, [x], [pwc, dwc, sw, sb], dev=dev, tol=3e-2, ptol=6e-2)


@pytest.mark.parametrize("n,HW", [(2, (8, 8)), (3, (7, 7))])
def test_dseb_combine(dev, n, HW):
    g = G(n)
    y = torch.randn(2, 4, *HW, generator=g)
    diff = torch.randn(2, 4, *HW, generator=g)
    recs = [y + 0.3 * torch.randn(2, 4, *HW, generator=g) for _ in range(n - 1)]
    w = torch.randn(1, 4, 1, 1, generator=g)

    def fn(y, diff, *rest):
        rs, w = rest[:-1], rest[-1]
        return ops.dseb_combine(y, w, diff, [None] + list(rs))
    compare(fn, [y, diff] + recs, [w], dev=dev)


@pytest.mark.parametrize("K", [4, 9])
def test_seg_loss(dev, K):
    g = G(K)
    logits = torch.randn(2, K, 12, 12, generator=g) * 2
    labels = torch.randint(0, K, (2, 12, 12), generator=g).float().to(dev)
    res = {}
    for mode in ("f32", "bf16"):
        lg = (_r(logits).to(dev).to(BF) if mode == "bf16" else _r(logits).to(dev)).requires_grad_(True)
        loss = ops.dice_ce_loss(lg, labels, 0.4, 0.3, 0.3)
        loss.backward()
        res[mode] = (loss.item(), lg.grad.float().cpu())
    assert abs(res["f32"][0] - res["bf16"][0]) < 1e-5 * max(1.0, abs(res["f32"][0]))  # same inputs, fp32 arithmetic
    assert _rel(res["bf16"][1], res["f32"][1]) < 1e-2


def test_weight_shadow_follows_in_place_updates(dev):
    """kern.wq: the bf16 shadow of a weight is re-cast when the tensor changed through torch, kept when it did not"""
    W = torch.nn.Parameter(torch.randn(8, 8, generator=G(0)).to(dev))
    x = torch.zeros(1, 8, device=dev, dtype=BF)
    s1 = kern.wq(W, x)
    assert s1.dtype == BF and kern.wq(W, x) is s1
    torch.testing.assert_close(s1.float().cpu(), W.detach().cpu().bfloat16().float())
    with torch.no_grad():
        W.mul_(2.0)
    s2 = kern.wq(W, x)
    torch.testing.assert_close(s2.float().cpu(), W.detach().cpu().bfloat16().float())
    assert kern.wq(W, torch.zeros(1, device=dev)) is W  # fp32 activations read the master weight itself


def test_arena_shadow_is_refreshed_by_the_fused_sgd(dev):
    from cenet_amd import optim
    m = torch.nn.Module()
    m.w = torch.nn.Parameter(torch.randn(70, generator=G(1)).to(dev))
    arena = optim.ParamArena(m)
    opt = optim.FusedSGD(arena, lr=0.1, momentum=0.0, weight_decay=0.0)
    sh = kern.wq(m.w, torch.zeros(1, device=dev, dtype=BF))
    assert sh.data_ptr() == arena.shadow.data_ptr()
    opt.zero_grad()
    m.w.grad.fill_(1.0)
    opt.step()
    torch.testing.assert_close(kern.wq(m.w, sh).float().cpu(), m.w.detach().cpu().bfloat16().float())
    torch.testing.assert_close(arena.shadow[:70].float().cpu(), (m.w.detach()).cpu().bfloat16().float())
