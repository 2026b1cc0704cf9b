"""Glue launches folded into their producers (round 4).

* LayerNorm backward also writes the DropPath-scaled gradient the branch upstream wants (cenet_layernorm_bwd_add_part_scaled_bf16):
  bit-identical to scale_batch of its dx; a PVT block chain with per-sample scales gives the SAME gradients with the second output
  on and off, and LinearFn.backward launches no scale pass when it finds the pre-scaled copy.
`sim` = the same kernel source on the host SIMT checker (CPU); `hip` = the gfx950 library (marker gpu)."""
import pytest
import torch

from backend import dev  # noqa: F401
from cenet_amd import kern, ops

BF = torch.bfloat16


@pytest.mark.parametrize("B,N,Cn", [(4, 49, 64), (3, 50, 128), (2, 196, 320), (5, 21, 512)])
def test_ln_backward_second_output_is_the_scale_pass(dev, B, N, Cn):
    g = torch.Generator().manual_seed(B * 1000 + Cn)
    rows = B * N
    x = torch.randn(B, N, Cn, generator=g).to(BF).to(dev)
    dy = torch.randn(B, N, Cn, generator=g).to(BF).to(dev)
    add = torch.randn(B, N, Cn, generator=g).to(BF).to(dev)
    gamma = (1 + 0.1 * torch.randn(Cn, generator=g)).to(dev)
    beta = torch.zeros(Cn).to(dev)
    s = torch.tensor([0.0, 1.25, 1.25, 0.0, 1.25][:B]).to(dev)
    y, mean, rstd = torch.empty_like(x), torch.empty(rows).to(dev), torch.empty(rows).to(dev)
    kern.layernorm_fwd(x, gamma, beta, y, mean, rstd, rows, Cn, 1e-5)
    dx0, dx1, dxs = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    p0 = kern.layernorm_bwd_part(dy, x, gamma, mean, rstd, dx0, rows, Cn, dx_add=add)
    p1 = kern.layernorm_bwd_part(dy, x, gamma, mean, rstd, dx1, rows, Cn, dx_add=add, bscale=s, dxs=dxs)
    assert torch.equal(dx0, dx1) and torch.equal(p0, p1)
    want = torch.empty_like(x)
    kern.scale_batch(dx0, s, want, B, N * Cn)
    assert torch.equal(dxs, want)


def test_ln_fold_group_with_a_shared_destination(dev):
    """the same LayerNorm parameters twice in one flush (the model run twice before one backward pass, a shared norm
    module): the fold adds with plain read-modify-writes, so duplicates must land in separate launches — both contributions
    arrive.  Items 0 and 2 share dgamma / dbeta, 1 has its own; 60 more items force a second launch by count as well."""
    g = torch.Generator().manual_seed(5)
    Cn = 64
    dg = [torch.zeros(Cn).to(dev) for _ in range(2)]
    db = [torch.zeros(Cn).to(dev) for _ in range(2)]
    parts = [torch.randn(r, 2 * Cn, generator=g).to(dev) for r in (7, 40, 13)]
    items = [(parts[0], dg[0], db[0]), (parts[1], dg[1], db[1]), (parts[2], dg[0], db[0])]
    many = [torch.randn(3, 2 * Cn, generator=g).to(dev) for _ in range(60)]
    items += [(p, dg[1], db[1]) for p in many]
    kern.ln_fold_group(items)
    s0 = parts[0].sum(0) + parts[2].sum(0)
    s1 = parts[1].sum(0) + sum(p.sum(0) for p in many)
    torch.testing.assert_close(dg[0], s0[:Cn], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(db[0], s0[Cn:], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(dg[1], s1[:Cn], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(db[1], s1[Cn:], rtol=1e-4, atol=1e-4)


def _chain(dev_, masks, seed=0):
    """two PVT-style half blocks on tokens: x1 = x + s1 * Linear(LN(x)); x2 = x1 + s2 * Linear(LN(x1)); out = LN(x2)"""
    g = torch.Generator().manual_seed(seed)
    B, N, Cn = 4, 49, 64
    x = torch.randn(B, N, Cn, generator=g).to(BF).to(dev_).requires_grad_(True)
    ps = []
    for _ in range(3):
        ps += [(1 + 0.1 * torch.randn(Cn, generator=g)).to(dev_).requires_grad_(True), (0.1 * torch.randn(Cn, generator=g)).to(dev_).requires_grad_(True)]
    Ws = [(0.1 * torch.randn(Cn, Cn, generator=g)).to(dev_).requires_grad_(True) for _ in range(2)]
    bs = [(0.1 * torch.randn(Cn, generator=g)).to(dev_).requires_grad_(True) for _ in range(2)]
    t = x
    for i in range(2):
        y, tr = ops.layernorm_res(t, ps[2 * i], ps[2 * i + 1], 1e-5)
        t = ops.linear(y, Ws[i], bs[i], resid=tr, bscale=masks[i].to(dev_))
    out = ops.layernorm(t, ps[4], ps[5], 1e-5)
    w = torch.randn(B, N, Cn, generator=g).to(BF).to(dev_)
    for p in ps + Ws + bs:
        p.grad = torch.zeros_like(p)
    (out.float() * w.float()).sum().backward()
    ops.wgrad_flush()
    return [x.grad] + [p.grad for p in ps + Ws + bs]


def test_block_chain_gradients_do_not_depend_on_where_the_scale_is_applied(dev, monkeypatch):
    masks = [torch.tensor([1.25, 0.0, 1.25, 1.25]), torch.tensor([0.0, 1.25, 1.25, 0.0])]
    calls = []
    orig = kern.scale_batch
    monkeypatch.setattr(kern, "scale_batch", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
    monkeypatch.setattr(ops._WgradCfg, "prescale", True)
    on = _chain(dev, masks)
    n_on = len(calls)
    monkeypatch.setattr(ops._WgradCfg, "prescale", False)
    off = _chain(dev, masks)
    n_off = len(calls) - n_on
    assert n_off == 2 and n_on == (0 if ops._WgradCfg.grouping else 2)
    for a, b in zip(on, off):
        assert torch.equal(a, b)
    assert not any(st.prescaled for st in ops._WG.values())


def _attn_then_fused_mlp(dev_, s1, s2, seed=3):
    """x1 = x + s1 * Linear(LN(x)) (the attention half's tail), then the fused MLP half reads x1: its second backward kernel is the
    producer of the gradient the Linear wants scaled"""
    from test_pvt_mlp import _params
    B, H, W, Cn, HD = 2, 8, 14, 64, 128
    g = torch.Generator().manual_seed(seed)
    p = _params(Cn, HD, dev_, seed)
    x = torch.randn(B, H * W, Cn, generator=g).to(BF).to(dev_).requires_grad_(True)
    ng = (1 + 0.1 * torch.randn(Cn, generator=g)).to(dev_).requires_grad_(True)
    nb = (0.1 * torch.randn(Cn, generator=g)).to(dev_).requires_grad_(True)
    Wp = (0.1 * torch.randn(Cn, Cn, generator=g)).to(dev_).requires_grad_(True)
    bp = (0.1 * torch.randn(Cn, generator=g)).to(dev_).requires_grad_(True)
    y, xr = ops.layernorm_res(x, ng, nb, 1e-6)
    x1 = ops.linear(y, Wp, bp, resid=xr, bscale=s1.to(dev_))
    assert ops.pvt_mlp_supported(x1, HD, H, W)
    out = ops.pvt_mlp(x1, H, W, p["ln_g"], p["ln_b"], 1e-6, p["w1"], p["b1"], p["wd"], p["bd"], p["w2"], p["b2"], s2.to(dev_))
    w = torch.randn(B, H * W, Cn, generator=g).to(BF).to(dev_)
    leaves = [ng, nb, Wp, bp] + list(p.values())
    for q in leaves:
        q.grad = torch.zeros_like(q)
    (out.float() * w.float()).sum().backward()
    ops.wgrad_flush()
    return [x.grad] + [q.grad for q in leaves]


def test_fused_mlp_backward_hands_the_attention_half_its_scaled_gradient(dev, monkeypatch):
    s1, s2 = torch.tensor([1.25, 0.0]), torch.tensor([1.25, 1.25])
    calls = []
    orig = kern.scale_batch
    monkeypatch.setattr(kern, "scale_batch", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
    monkeypatch.setattr(ops._WgradCfg, "prescale", True)
    on = _attn_then_fused_mlp(dev, s1, s2)
    n_on = len(calls)
    monkeypatch.setattr(ops._WgradCfg, "prescale", False)
    off = _attn_then_fused_mlp(dev, s1, s2)
    assert n_on == 0 and len(calls) == 1
    for i, (a, b) in enumerate(zip(on, off)):
        # (the fused kernels' affine / depthwise gradient sums use LDS float atomics: their order varies from run to run on the GPU)
        if a.dtype == BF or str(dev) == "cpu":
            assert torch.equal(a, b), i
        else:
            assert torch.allclose(a, b, rtol=1e-4, atol=1e-5 * max(b.abs().max().item(), 1.0)), i
    assert not any(st.prescaled for st in ops._WG.values())


def test_sra_backward_accumulator_is_zero_at_rest(dev, monkeypatch):
    """dK / dV of the multi-workgroup spatial-reduction attention backward: a persistent accumulator + cast_clear instead of a zero
    fill before and a cast after every use — same bits as the filled form, also on the second use and after an interrupted one"""
    B, H, Nq, Nk, hd = 1, 1, 1024, 49, 64
    g = torch.Generator().manual_seed(11)
    q = torch.randn(B, Nq, H * hd, generator=g).to(BF).to(dev).requires_grad_(True)
    kv = torch.randn(B, Nk, 2 * H * hd, generator=g).to(BF).to(dev).requires_grad_(True)
    w = torch.randn(B, Nq, H * hd, generator=g).to(BF).to(dev)
    if kern.sra_attn_bwd_direct_supported(B, H, Nq, Nk):
        pytest.skip("one workgroup per (batch, head) here: no accumulator")

    def run():
        q.grad = kv.grad = None
        ops.sr_attention(q, kv, H).backward(w)
        return q.grad.clone(), kv.grad.clone()

    takes = []
    orig_take = ops._ZeroWs.take
    monkeypatch.setattr(ops._ZeroWs, "take", staticmethod(lambda shape, ref: (takes.append(1), orig_take(shape, ref))[1]))
    a = run()
    b = run()
    assert len(takes) == 2
    key = ops._ZeroWs._key(dev, kv.numel())
    assert all(e[1] is False and float(e[0].abs().max()) == 0.0 for e in ops._ZeroWs.bufs[key])
    # an interrupted use (a taker that never gave its accumulator back) does not reach the next taker: it gets a buffer of its
    # own, two live takers of one size never alias, and once MAX_LIVE leftovers pile up the oldest is filled again and reused
    leaked = ops._ZeroWs.take(kv.shape, kv).fill_(3.0)
    other = ops._ZeroWs.take(kv.shape, kv)
    assert other.data_ptr() != leaked.data_ptr() and float(other.abs().max()) == 0.0
    c = run()
    for _ in range(ops._ZeroWs.MAX_LIVE + 1):
        t = ops._ZeroWs.take(kv.shape, kv)
        assert float(t.abs().max()) == 0.0
        t.fill_(5.0)
    assert len(ops._ZeroWs.bufs[key]) == ops._ZeroWs.MAX_LIVE
    c2 = run()
    assert torch.equal(c2[0], c[0])
    ops._ZeroWs.bufs.pop(key)
    # the filled form
    monkeypatch.setattr(ops._ZeroWs, "take", staticmethod(lambda shape, ref: ops._zeros(shape, ref)))
    monkeypatch.setattr(ops._ZeroWs, "give_back_as", staticmethod(lambda ws, like: kern.cast(ws, like.dtype)))
    ref = run()
    for x in (a, b, c):
        assert torch.equal(x[0], ref[0])
        # (fp32 atomics from several workgroups: the order of the adds is not fixed on the GPU)
        if dev.type == "cpu":
            assert torch.equal(x[1], ref[1])
        else:
            torch.testing.assert_close(x[1].float(), ref[1].float(), rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("dt", [torch.float32, BF])
def test_concat_backward_adds_the_gradients_of_its_inputs_other_consumers(dev, dt):
    """y = cat(a, b); a and b each have a second consumer reading the tap: one split launch writes slice + other gradient"""
    g = torch.Generator().manual_seed(4)
    B, Ca, Cb, H, W = 2, 8, 24, 7, 7
    a0 = torch.randn(B, Ca, H, W, generator=g).to(dt).to(dev)
    b0 = torch.randn(B, Cb, H, W, generator=g).to(dt).to(dev)
    wy = torch.randn(B, Ca + Cb, H, W, generator=g).to(dt).to(dev)
    wa = torch.randn(B, Ca, H, W, generator=g).to(dt).to(dev)
    wb = torch.randn(B, Cb, H, W, generator=g).to(dt).to(dev)
    res = []
    for tap in (True, False):
        a, b = a0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
        if tap:
            y, at, bt = ops.concat2(a, b, tap=True)
        else:
            y, at, bt = ops.concat2(a, b), a, b
        ((y.float() * wy.float()).sum() + (at.float() * wa.float()).sum() + (bt.float() * wb.float()).sum()).backward()
        res.append((y.detach(), a.grad, b.grad))
    assert torch.equal(res[0][0], res[1][0])
    tol = dict(rtol=0, atol=0) if dt == torch.float32 else dict(rtol=2e-2, atol=2e-2)  # (bf16: one rounding instead of two)
    torch.testing.assert_close(res[0][1].float(), res[1][1].float(), **tol)
    torch.testing.assert_close(res[0][2].float(), res[1][2].float(), **tol)
    # only one of the taps used / none used
    a, b = a0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
    y, at, bt = ops.concat2(a, b, tap=True)
    ((y.float() * wy.float()).sum() + (bt.float() * wb.float()).sum()).backward()
    torch.testing.assert_close(a.grad.float(), wy[:, :Ca].float(), **tol)
    torch.testing.assert_close(b.grad.float(), (wy[:, Ca:].float() + wb.float()), **tol)


@pytest.mark.parametrize("dt", [torch.float32, BF])
@pytest.mark.parametrize("B,Cn,H,W", [(2, 16, 7, 7), (3, 8, 14, 14), (2, 8, 40, 36), (1, 24, 5, 3), (2, 4, 64, 66)])
def test_batchnorm_backward_adds_the_residual_gradient(dev, dt, B, Cn, H, W):
    """x + f(BN(x)) with the residual reading the tap: same gradients as the untapped form (every plane-size class of the backward)"""
    g = torch.Generator().manual_seed(B * 100 + Cn)
    x0 = torch.randn(B, Cn, H, W, generator=g).to(dt).to(dev)
    wy = torch.randn(B, Cn, H, W, generator=g).to(dt).to(dev)
    wr = torch.randn(B, Cn, H, W, generator=g).to(dt).to(dev)
    res = []
    for tap in (True, False):
        x = x0.clone().requires_grad_(True)
        gam = (1 + 0.1 * torch.arange(Cn, dtype=torch.float32)).to(dev).requires_grad_(True)
        bet = (0.05 * torch.arange(Cn, dtype=torch.float32)).to(dev).requires_grad_(True)
        rm, rv, nbt = torch.zeros(Cn).to(dev), torch.ones(Cn).to(dev), torch.zeros((), dtype=torch.long).to(dev)
        gam.grad, bet.grad = torch.zeros_like(gam), torch.zeros_like(bet)
        if tap:
            y, xt = ops.batchnorm(x, gam, bet, rm, rv, nbt, True, 1e-5, "lrelu", 0.01, 0.1, tap=True)
        else:
            y, xt = ops.batchnorm(x, gam, bet, rm, rv, nbt, True, 1e-5, "lrelu", 0.01, 0.1), x
        ((y.float() * wy.float()).sum() + (xt.float() * wr.float()).sum()).backward()
        res.append((y.detach(), x.grad, gam.grad, bet.grad))
    assert torch.equal(res[0][0], res[1][0])
    tol = dict(rtol=1e-6, atol=1e-6) if dt == torch.float32 else dict(rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(res[0][1].float(), res[1][1].float(), **tol)
    torch.testing.assert_close(res[0][2], res[1][2], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(res[0][3], res[1][3], rtol=1e-4, atol=1e-4)


def test_split_k_linear_tail_is_one_pass(dev, monkeypatch):
    """the spatial-reduction convs as split-K GEMMs (few rows, K = C s^2): atomics into a zero-at-rest accumulator, then ONE pass
    adds the bias, rounds and clears — same numbers as bias pre-fill + atomics + cast, twice in a row"""
    g = torch.Generator().manual_seed(9)
    R, K, N = 98, 2048, 64
    x = (torch.randn(2, R // 2, K, generator=g) * 0.5).to(BF).to(dev)
    W = (torch.randn(N, K, generator=g) * K ** -0.5).to(dev).requires_grad_(True)
    b = (0.1 * torch.randn(N, generator=g)).to(dev).requires_grad_(True)
    monkeypatch.setattr(kern, "pick_splits", lambda *a, **k: 4)
    with torch.no_grad():
        y1 = ops.linear(x, W, b, split_k=True)
        y2 = ops.linear(x, W, b, split_k=True)
        ref = ops.linear(x, W, b)
    assert y1.dtype == BF and torch.equal(y1, y2) if dev.type == "cpu" else True
    torch.testing.assert_close(y1.float(), ref.float(), rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(y2.float(), ref.float(), rtol=2e-2, atol=2e-2)
    key = ops._ZeroWs._key(dev, R * N)
    assert all(e[1] is False and float(e[0].abs().max()) == 0.0 for e in ops._ZeroWs.bufs[key])


@pytest.mark.parametrize("dt", [torch.float32, BF])
@pytest.mark.parametrize("B,H,W,Cn", [(2, 7, 7, 64), (1, 5, 3, 10), (2, 14, 14, 320), (1, 9, 4, 33)])
def test_layout_change_backward_adds_the_token_path_gradient(dev, dt, B, H, W, Cn):
    g = torch.Generator().manual_seed(H * 10 + Cn)
    t0 = torch.randn(B, H * W, Cn, generator=g).to(dt).to(dev)
    wn = torch.randn(B, Cn, H, W, generator=g).to(dt).to(dev)
    wt = torch.randn(B, H * W, Cn, generator=g).to(dt).to(dev)
    t = t0.clone().requires_grad_(True)
    x, tt = ops.tok_to_nchw(t, H, W, tap=True)
    assert torch.equal(x, t0.transpose(1, 2).reshape(B, Cn, H, W))
    ((x.float() * wn.float()).sum() + (tt.float() * wt.float()).sum()).backward()
    want = wn.float().reshape(B, Cn, H * W).transpose(1, 2) + wt.float()
    tol = dict(rtol=0, atol=0) if dt == torch.float32 else dict(rtol=1e-2, atol=2e-2)
    torch.testing.assert_close(t.grad.float(), want, **tol)
    t2 = t0.clone().requires_grad_(True)
    x2, _ = ops.tok_to_nchw(t2, H, W, tap=True)  # tap unused: the plain transpose
    (x2.float() * wn.float()).sum().backward()
    torch.testing.assert_close(t2.grad.float(), wn.float().reshape(B, Cn, H * W).transpose(1, 2).to(dt).float(), rtol=0, atol=0)


@pytest.mark.parametrize("dt", [torch.float32, BF])
@pytest.mark.parametrize("scale", [0.8, 0.4, 2.0, 0.5])  # (0.5 on bf16 maps: the specialised x0.5 backward kernel with the addend)
def test_bilinear_backward_adds_the_gradient_of_other_consumers(dev, dt, scale):
    g = torch.Generator().manual_seed(int(scale * 10))
    B, Cn, H, W = 2, 6, 14, 10
    x0 = torch.randn(B, Cn, H, W, generator=g).to(dt).to(dev)
    wt = torch.randn(B, Cn, H, W, generator=g).to(dt).to(dev)
    res = []
    for tap in (True, False):
        x = x0.clone().requires_grad_(True)
        if tap:
            y, xt = ops.interpolate_bilinear(x, scale_factor=scale, tap=True)
        else:
            y, xt = ops.interpolate_bilinear(x, scale_factor=scale), x
        wy = torch.linspace(-1, 1, y.numel()).reshape(y.shape).to(dt).to(dev)
        ((y.float() * wy.float()).sum() + (xt.float() * wt.float()).sum()).backward()
        res.append((y.detach(), x.grad))
    assert torch.equal(res[0][0], res[1][0])
    tol = dict(rtol=1e-6, atol=1e-6) if dt == torch.float32 else dict(rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(res[0][1].float(), res[1][1].float(), **tol)


def test_multi_linear_tap_gradient_rides_in_the_data_gradient_gemm(dev):
    g = torch.Generator().manual_seed(2)
    R, K, N, n = 96, 64, 64, 3
    x0 = torch.randn(2, R // 2, K, generator=g).to(BF).to(dev)
    W = (torch.randn(n, N, K, generator=g) * K ** -0.5).to(dev).requires_grad_(True)
    ws = [torch.randn(2, R // 2, N, generator=g).to(BF).to(dev) for _ in range(n)]
    wt = torch.randn(2, R // 2, K, generator=g).to(BF).to(dev)
    res = []
    for tap in (True, False):
        x = x0.clone().requires_grad_(True)
        W.grad = torch.zeros_like(W)
        outs = ops.multi_linear(x, W, tap=tap)
        xt = outs[n] if tap else x
        (sum((o.float() * w.float()).sum() for o, w in zip(outs[:n], ws)) + (xt.float() * wt.float()).sum()).backward()
        ops.wgrad_flush()
        res.append((x.grad, W.grad.clone()))
    torch.testing.assert_close(res[0][0].float(), res[1][0].float(), rtol=2e-2, atol=3e-2)
    torch.testing.assert_close(res[0][1], res[1][1], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("B,Cn,H,W", [(2, 8, 8, 16), (3, 4, 6, 24), (1, 32, 16, 8)])
def test_head_image_branch_tail_fused_equals_the_chain(dev, monkeypatch, B, Cn, H, W):
    """w * MaxPool(LeakyReLU(BN2(conv2) + BN3(conv3))) as one forward kernel and two backward kernels (csrc/res_tail.hip) against the
    chain bn x 2 + add_act + maxpool2_scale, through the module that uses it (UnetResBlock with pool_scale)"""
    import copy

    from cenet_amd.networks.cenet.modules.unet import UnetResBlock
    torch.manual_seed(5)
    blk = UnetResBlock(2, 1, Cn, kernel_size=3, stride=1, norm_name="batch").to(dev)
    with torch.no_grad():
        for bn in (blk.norm1, blk.norm2, blk.norm3):
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.uniform_(-0.3, 0.3)
    w = (torch.randn(1, Cn, 1, 1) + 0.75).to(dev).requires_grad_(True)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, 1, H, W, generator=g).to(BF).to(dev)
    go = torch.randn(B, Cn, H // 2, W // 2, generator=g).to(BF).to(dev)
    res = []
    for fused in ("1", "x3", "0"):  # 1: shortcut not materialised (one-channel input); x3: the two-tensor form; 0: the chain
        monkeypatch.setenv("CENET_RES_TAIL_FUSED", fused)
        m = copy.deepcopy(blk).train()
        wc = w.detach().clone().requires_grad_(True)
        for p_ in list(m.parameters()) + [wc]:
            p_.grad = torch.zeros_like(p_)
        out = m(x, wc)
        out.backward(go)
        ops.wgrad_flush()
        res.append((out.detach().float(), wc.grad.clone(), [p_.grad.clone() for p_ in m.parameters()],
                    [b_.clone() for b_ in m.buffers()]))
    for f in res[:2]:
        _compare_tail(f, res[2])


def _compare_tail(f, u):
    # the fused form rounds once where the chain rounds three times (two normalised maps and their sum): bf16-ulp differences,
    # and a window whose two largest activations tie after the chain's rounding may pick another element
    d = (f[0] - u[0]).abs()
    assert d.max().item() <= 0.06 * max(u[0].abs().max().item(), 1e-3) and d.mean().item() <= 4e-3 * u[0].abs().mean().item()
    cos = lambda a, b: float((a.flatten().double() @ b.flatten().double()) / (a.double().norm() * b.double().norm() + 1e-30))
    assert cos(f[1], u[1]) > 0.999
    for a, b_ in zip(f[2], u[2]):
        if b_.abs().max() > 1e-6:
            assert cos(a, b_) > 0.99, (a.shape, cos(a, b_))
    for a, b_ in zip(f[3], u[3]):  # running statistics and counters: the same statistics kernels
        torch.testing.assert_close(a.float(), b_.float(), rtol=1e-5, atol=1e-6)


def test_res_tail_kernels_against_fp64_autograd(dev):
    import torch.nn.functional as F
    B, Cn, H, W = 2, 6, 8, 16
    g = torch.Generator().manual_seed(3)
    x2 = torch.randn(B, Cn, H, W, generator=g).to(BF)
    x3 = (0.7 * torch.randn(B, Cn, H, W, generator=g) + 0.2).to(BF)
    go = torch.randn(B, Cn, H // 2, W // 2, generator=g).to(BF)
    par = [(1 + 0.3 * torch.randn(Cn, generator=g)), 0.2 * torch.randn(Cn, generator=g), (1 + 0.3 * torch.randn(Cn, generator=g)),
           0.2 * torch.randn(Cn, generator=g), torch.randn(Cn, generator=g) + 0.75]
    # fp64 reference
    r = [t.double().requires_grad_(True) for t in (x2, x3)] + [p.double().requires_grad_(True) for p in par]
    y = F.leaky_relu(F.batch_norm(r[0], None, None, r[2], r[3], True, 0.1, 1e-5) + F.batch_norm(r[1], None, None, r[4], r[5], True, 0.1, 1e-5), 0.01)
    out_ref = F.max_pool2d(y, 2, 2) * r[6].view(1, -1, 1, 1)
    out_ref.backward(go.double())
    # kernels
    d = dev
    bn = []
    for _ in range(2):
        m = torch.nn.BatchNorm2d(Cn).to(d).train()
        bn.append(m)
    with torch.no_grad():
        bn[0].weight.copy_(par[0]); bn[0].bias.copy_(par[1]); bn[1].weight.copy_(par[2]); bn[1].bias.copy_(par[3])
    w = par[4].to(d).view(1, Cn, 1, 1).requires_grad_(True)
    a2, a3 = x2.to(d).requires_grad_(True), x3.to(d).requires_grad_(True)
    for p_ in [w] + list(bn[0].parameters()) + list(bn[1].parameters()):
        p_.grad = torch.zeros_like(p_)
    out = ops.res_tail_pool(a2, bn[0], a3, bn[1], w, 0.01)
    out.backward(go.to(d))
    tol = dict(rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(out.float().cpu(), out_ref.float(), **tol)
    torch.testing.assert_close(a2.grad.float().cpu(), r[0].grad.float(), **tol)
    torch.testing.assert_close(a3.grad.float().cpu(), r[1].grad.float(), **tol)
    for got, want in ((bn[0].weight.grad, r[2].grad), (bn[0].bias.grad, r[3].grad), (bn[1].weight.grad, r[4].grad),
                      (bn[1].bias.grad, r[5].grad), (w.grad.view(-1), r[6].grad)):
        torch.testing.assert_close(got.float().cpu(), want.float(), rtol=2e-2, atol=3e-2)
    # running statistics as torch's (momentum 0.1, unbiased running variance)
    rm = 0.1 * x2.double().mean((0, 2, 3))
    rv = 0.9 + 0.1 * x2.double().var((0, 2, 3), unbiased=True)
    torch.testing.assert_close(bn[0].running_mean.double().cpu(), rm, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(bn[0].running_var.double().cpu(), rv, rtol=1e-4, atol=1e-5)
    assert int(bn[0].num_batches_tracked) == 1


def test_res_tail_image_form_against_fp64_autograd(dev):
    """shortcut = 1x1 conv of the one-channel image, never materialised: output, dx2 and every parameter gradient — including the
    shortcut weight's, which reaches the output through eps only — against fp64 autograd of the unfused expression"""
    import torch.nn.functional as F
    B, Cn, H, W = 2, 6, 8, 16
    g = torch.Generator().manual_seed(8)
    x2 = torch.randn(B, Cn, H, W, generator=g).to(BF)
    img = (0.5 * torch.randn(B, 1, H, W, generator=g) + 0.1).to(BF)
    go = torch.randn(B, Cn, H // 2, W // 2, generator=g).to(BF)
    w3 = torch.tensor([0.8, -0.6, 0.05, 1.3, -0.02, 0.4])
    par = [(1 + 0.3 * torch.randn(Cn, generator=g)), 0.2 * torch.randn(Cn, generator=g), (1 + 0.3 * torch.randn(Cn, generator=g)),
           0.2 * torch.randn(Cn, generator=g), torch.randn(Cn, generator=g) + 0.75]
    eps3 = 1e-3  # (large enough for the eps-only gradient of w3 to be measurable against fp64)
    r = [x2.double().requires_grad_(True), w3.double().requires_grad_(True)] + [p.double().requires_grad_(True) for p in par]
    x3 = img.double() * r[1].view(1, -1, 1, 1)
    y = F.leaky_relu(F.batch_norm(r[0], None, None, r[2], r[3], True, 0.1, 1e-5) + F.batch_norm(x3, None, None, r[4], r[5], True, 0.1, eps3), 0.01)
    out_ref = F.max_pool2d(y, 2, 2) * r[6].view(1, -1, 1, 1)
    out_ref.backward(go.double())
    d = dev
    bn = [torch.nn.BatchNorm2d(Cn).to(d).train(), torch.nn.BatchNorm2d(Cn, eps=eps3).to(d).train()]
    with torch.no_grad():
        bn[0].weight.copy_(par[0]); bn[0].bias.copy_(par[1]); bn[1].weight.copy_(par[2]); bn[1].bias.copy_(par[3])
    w = par[4].to(d).view(1, Cn, 1, 1).requires_grad_(True)
    w3d = w3.to(d).view(Cn, 1, 1, 1).requires_grad_(True)
    a2 = x2.to(d).requires_grad_(True)
    for p_ in [w, w3d] + list(bn[0].parameters()) + list(bn[1].parameters()):
        p_.grad = torch.zeros_like(p_)
    assert ops.res_tail_img_pool_supported(a2, img.to(d), w3d, bn[0], bn[1], w)
    out = ops.res_tail_img_pool(a2, bn[0], img.to(d), w3d, bn[1], w, 0.01)
    out.backward(go.to(d))
    tol = dict(rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(out.float().cpu(), out_ref.float(), **tol)
    torch.testing.assert_close(a2.grad.float().cpu(), r[0].grad.float(), **tol)
    for got, want in ((bn[0].weight.grad, r[2].grad), (bn[0].bias.grad, r[3].grad), (bn[1].weight.grad, r[4].grad),
                      (bn[1].bias.grad, r[5].grad), (w.grad.view(-1), r[6].grad)):
        torch.testing.assert_close(got.float().cpu(), want.float(), rtol=2e-2, atol=3e-2)
    # the shortcut weight's gradient: small (eps / var) but exact in form — compare with relative tolerance on its own scale
    gw3, ref3 = w3d.grad.view(-1).double().cpu(), r[1].grad
    assert float((gw3 - ref3).abs().max()) <= 0.05 * float(ref3.abs().max()) + 1e-7, (gw3, ref3)
    # BatchNorm3's running statistics: those of w3 * img
    x3f = img.double() * w3.double().view(1, -1, 1, 1)
    torch.testing.assert_close(bn[1].running_mean.double().cpu(), 0.1 * x3f.mean((0, 2, 3)), rtol=1e-3, atol=1e-5)
    torch.testing.assert_close(bn[1].running_var.double().cpu(), 0.9 + 0.1 * x3f.var((0, 2, 3), unbiased=True), rtol=1e-3, atol=1e-5)
    assert int(bn[1].num_batches_tracked) == 1 and int(bn[0].num_batches_tracked) == 1


@pytest.mark.parametrize("B,Cn", [(2, 5), (32, 64), (7, 300), (64, 33)])
def test_batchnorm1d_one_launch_each_way(dev, B, Cn):
    """the CCU gate's BatchNorm1d on a [B, C] fp32 matrix (cfam.py:251-264): one kernel per pass, against torch"""
    g = torch.Generator().manual_seed(B + Cn)
    z = torch.randn(B, Cn, generator=g) * 2 + 0.5
    go = torch.randn(B, Cn, generator=g)
    ref = torch.nn.BatchNorm1d(Cn).train()
    with torch.no_grad():
        ref.weight.uniform_(0.5, 1.5, generator=g)
        ref.bias.uniform_(-0.5, 0.5, generator=g)
    zr = z.clone().requires_grad_(True)
    ref(zr).backward(go)
    gam, bet = ref.weight.detach().clone().to(dev), ref.bias.detach().clone().to(dev)
    rm, rv, nbt = torch.zeros(Cn).to(dev), torch.ones(Cn).to(dev), torch.zeros(1, dtype=torch.long).to(dev)
    zd, zn, mean, var = z.to(dev), torch.empty(B, Cn).to(dev), torch.empty(Cn).to(dev), torch.empty(Cn).to(dev)
    assert kern.bn1d_supported(B)
    kern.bn1d_train_fwd(zd, zn, mean, var, rm, rv, 0.1, nbt, 1e-5, gam, bet, B, Cn)
    torch.testing.assert_close(zn.cpu(), ref(z).detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(rm.cpu(), ref.running_mean / 1.0 * 0 + 0.1 * z.mean(0), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(rv.cpu(), 0.9 + 0.1 * z.var(0, unbiased=True), rtol=1e-5, atol=1e-6)
    assert int(nbt) == 1
    dz, dg, db = torch.empty(B, Cn).to(dev), torch.full((Cn,), 2.0).to(dev), torch.full((Cn,), -1.0).to(dev)
    kern.bn1d_bwd(go.to(dev), zd, dz, mean, var, 1e-5, gam, dg, db, B, Cn)
    torch.testing.assert_close(dz.cpu(), zr.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(dg.cpu() - 2.0, ref.weight.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(db.cpu() + 1.0, ref.bias.grad, rtol=1e-4, atol=1e-4)
    assert not kern.bn1d_supported(1) and not kern.bn1d_supported(65)


@pytest.mark.parametrize("Hi,Wi", [(8, 12), (56, 56), (6, 8), (9, 4)])
def test_bilinear_x2_align_corners_backward_quad_kernel(dev, Hi, Wi):
    """nn.UpsamplingBilinear2d(scale_factor=2) on bf16 maps (the head's UpConv): the backward kernel that owns four input pixels per
    thread (bilinear_up2ac_bwd_kernel) against fp32 torch on the same bf16 gradient"""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(Hi * 7 + Wi)
    x = torch.randn(2, 3, Hi, Wi, generator=g).to(BF)
    xr = x.float().clone().requires_grad_(True)
    ref = F.interpolate(xr, scale_factor=2.0, mode="bilinear", align_corners=True)
    dy = torch.randn(ref.shape, generator=g).to(BF)
    ref.backward(dy.float())
    xd = x.to(dev).requires_grad_(True)
    y = ops.interpolate_bilinear(xd, scale_factor=2.0, align_corners=True)
    y.backward(dy.to(dev))
    torch.testing.assert_close(y.float().cpu(), ref.detach(), rtol=1e-2, atol=2e-2)
    torch.testing.assert_close(xd.grad.float().cpu(), xr.grad, rtol=1e-2, atol=2e-2)


@pytest.mark.parametrize("Hi,Wi,kw", [(7, 7, dict(scale_factor=7.0, align_corners=True)), (7, 7, dict(size=(49, 56))), (8, 5, dict(size=(64, 33), align_corners=True)),
                                      (3, 4, dict(scale_factor=2.0))])
def test_bilinear_backward_of_small_planes_as_two_products(dev, Hi, Wi, kw):
    """bf16 planes with Hi, Wi <= 8 and Ho, Wo <= 64 (the pooled branch's 7 x 7 -> 49 x 49): dx = Wy^T dy Wx through LDS"""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(Hi * 9 + Wi)
    x = torch.randn(2, 5, Hi, Wi, generator=g).to(BF)
    xr = x.float().clone().requires_grad_(True)
    ref = F.interpolate(xr, mode="bilinear", **kw)
    dy = torch.randn(ref.shape, generator=g).to(BF)
    ref.backward(dy.float())
    xd = x.to(dev).requires_grad_(True)
    y = ops.interpolate_bilinear(xd, size=kw.get("size"), scale_factor=kw.get("scale_factor"), align_corners=kw.get("align_corners", False))
    y.backward(dy.to(dev))
    torch.testing.assert_close(xd.grad.float().cpu(), xr.grad, rtol=1e-2, atol=4e-2)


@pytest.mark.parametrize("dt", [torch.float32, BF])
@pytest.mark.parametrize("H,W,oh,ow", [(56, 56, 7, 7), (14, 14, 7, 7), (7, 7, 7, 7), (10, 13, 3, 5), (5, 6, 7, 7), (28, 30, 7, 4)])
def test_adaptive_avgpool_backward_candidate_bins(dev, dt, H, W, oh, ow):
    """pooling down: a pixel belongs to at most two windows per axis (the kernel looks at three candidates instead of all bins);
    pooling up keeps the full walk"""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(H + W + oh)
    x = torch.randn(2, 3, H, W, generator=g).to(dt)
    xr = x.float().clone().requires_grad_(True)
    ref = F.adaptive_avg_pool2d(xr, (oh, ow))
    dy = torch.randn(ref.shape, generator=g).to(dt)
    ref.backward(dy.float())
    xd = x.to(dev).requires_grad_(True)
    y = ops.adaptive_avgpool(xd, oh, ow)
    y.backward(dy.to(dev))
    tol = dict(rtol=1e-5, atol=1e-6) if dt == torch.float32 else dict(rtol=1e-2, atol=2e-2)
    torch.testing.assert_close(y.float().cpu(), ref.detach(), **tol)
    torch.testing.assert_close(xd.grad.float().cpu(), xr.grad, **tol)


@pytest.mark.parametrize("B,C,H,W,s", [(2, 64, 16, 16, 8), (3, 64, 8, 12, 4), (2, 320, 4, 4, 2)])
def test_sr_conv_and_norm_as_one_tail_kernel(dev, B, C, H, W, s):
    """ops.sr_conv_ln (pvtv2.py:93-95,99-100, bf16 mode): patch rows -> split-K GEMM into the zero-at-rest accumulator -> ONE kernel
    that adds the bias, rounds, normalises and clears the accumulator (cenet_layernorm_fwd_acc_bf16) — against the launch chain
    conv2d_tok + layernorm: same rounded rows, same statistics; gradients of the input, conv and norm parameters."""
    g = torch.Generator().manual_seed(B + C + s)
    x0 = torch.randn(B, H * W, C, generator=g).to(BF)
    W0 = (torch.randn(C, C, s, s, generator=g) * (C * s * s) ** -0.5)
    b0, ga0, be0 = torch.randn(C, generator=g) * 0.1, 1 + 0.2 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    cot = torch.randn(B, (H // s) * (W // s), C, generator=g).to(BF).to(dev)
    res = []
    for fused in (True, False):
        x = x0.clone().to(dev).requires_grad_(True)
        ps = [t.clone().to(dev).requires_grad_(True) for t in (W0, b0, ga0, be0)]
        for p in ps:
            p.grad = torch.zeros_like(p)
        if fused:
            with torch.no_grad():
                xp = ops.PatchTokFn.apply(x, H, W, s)
            if not ops.linear_ln_supported(xp, ps[0], ps[1]):
                pytest.skip("reduction too short to split here")
            y = ops.sr_conv_ln(x, H, W, ps[0], ps[1], s, ps[2], ps[3], 1e-5)
        else:
            y = ops.layernorm(ops.conv2d_tok(x, H, W, ps[0], ps[1], stride=s, pad=0, out_layout="tok"), ps[2], ps[3], 1e-5)
        y.backward(cot)
        ops.wgrad_flush()
        res.append([y.detach().float().cpu(), x.grad.float().cpu()] + [p.grad.float().cpu() for p in ps])
    names = ["y", "dx", "dW", "db", "dgamma", "dbeta"]
    for n, a, b in zip(names, *res):
        tol = 2e-2 * float(b.abs().max()) + 1e-6  # (split-K atomics order the fp32 sums differently run to run; bf16 rows)
        assert float((a - b).abs().max()) <= tol, (n, float((a - b).abs().max()), tol)
    # the accumulator is zero at rest and free again
    for es in ops._ZeroWs.bufs.values():
        assert all(e[1] is False and float(e[0].abs().max()) == 0.0 for e in es)
