"""GEMM / implicit-GEMM kernel vs plain PyTorch fp32 (sim on CPU here, real kernels with -m gpu)."""
import pytest
import torch
import torch.nn.functional as F

from backend import dev  # noqa: F401
from cenet_amd import kern


def rnd(*s, dev, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*s, generator=g).to(dev)


@pytest.mark.parametrize("M,N,K", [(64, 64, 32), (70, 50, 45), (130, 33, 7)])
def test_gemm_nt_bias(dev, M, N, K):
    x, w, b = rnd(M, K, dev=dev), rnd(N, K, dev=dev, seed=1), rnd(N, dev=dev, seed=2)
    y = torch.empty(M, N, device=dev)
    kern.gemm(kern.mat_plain(x, K, 1, kfast=1), kern.mat_plain(w, 1, K, kfast=1), y, M, N, K, scr=N, scc=1, bias=b)
    torch.testing.assert_close(y.cpu(), F.linear(x, w, b).cpu(), rtol=1e-4, atol=1e-4)


def test_gemm_batched_nn_epilogue(dev):
    Bt, M, N, K = 3, 20, 70, 24
    w, x = rnd(M, K, dev=dev), rnd(Bt, K, N, dev=dev, seed=1)
    bias, R, bs = rnd(M, dev=dev, seed=2), rnd(Bt, M, N, dev=dev, seed=3), rnd(Bt, dev=dev, seed=4)
    y = torch.empty(Bt, M, N, device=dev)
    kern.gemm(kern.mat_plain(w, K, 1, kfast=1), kern.mat_plain(x, N, 1, sb=K * N), y, M, N, K, scr=N, scc=1,
              scb=M * N, nbatch=Bt, bias=bias, bias_on_row=True, act="gelu", bscale=bs, R=R, srb=M * N, srr=N, src=1)
    ref = F.gelu(torch.matmul(w, x) + bias[None, :, None]) * bs[:, None, None] + R
    torch.testing.assert_close(y.cpu(), ref.cpu(), rtol=1e-4, atol=1e-4)


# per-sample scale looked up by row in ONE flat GEMM (DropPath of pvtv2.py:117-118 on [B, n, K] activations): samples that
# straddle tile boundaries (n = 50, 100), interior tiles (lean epilogue) and ragged edge tiles, with and without bias / residual,
# and through the activation (generic) epilogue
@pytest.mark.parametrize("Bt,n,N,K,bias,resid,act", [(3, 50, 70, 24, True, True, "none"), (2, 100, 128, 40, False, True, "none"),
                                                      (4, 64, 64, 32, True, False, "none"), (3, 50, 36, 20, True, True, "gelu")])
def test_gemm_flat_per_row_sample_scale(dev, Bt, n, N, K, bias, resid, act):
    x, w = rnd(Bt, n, K, dev=dev), rnd(N, K, dev=dev, seed=1)
    b = rnd(N, dev=dev, seed=2) if bias else None
    R = rnd(Bt, n, N, dev=dev, seed=3) if resid else None
    bs = rnd(Bt, dev=dev, seed=4)
    y = torch.empty(Bt, n, N, device=dev)
    kern.gemm(kern.mat_plain(x, K, 1, kfast=1), kern.mat_plain(w, 1, K, kfast=1), y, Bt * n, N, K, scr=N, scc=1, bias=b,
              act=act, bscale=bs, bscale_rows=n, R=R, srr=N, src=1)
    ref = F.linear(x, w, b)
    if act == "gelu":
        ref = F.gelu(ref)
    ref = ref * bs[:, None, None]
    if resid:
        ref = ref + R
    torch.testing.assert_close(y.cpu(), ref.cpu(), rtol=1e-4, atol=1e-4)


def test_gemm_splitk_atomic_kbatch(dev):
    Bt, Co, Ci, HW = 4, 10, 12, 50
    dy, x = rnd(Bt, Co, HW, dev=dev), rnd(Bt, Ci, HW, dev=dev, seed=1)
    dw = torch.zeros(Co, Ci, device=dev)
    db = torch.zeros(Co, device=dev)
    kern.gemm(kern.mat_plain(dy, HW, 1, skb=Co * HW, kfast=1), kern.mat_plain(x, 1, HW, skb=Ci * HW, kfast=1), dw,
              Co, Ci, HW, scr=Ci, scc=1, nkb=Bt, splits=3, atomic=True, asum=db)
    ref = torch.einsum("bop,bip->oi", dy, x)
    torch.testing.assert_close(dw.cpu(), ref.cpu(), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(db.cpu(), dy.sum((0, 2)).cpu(), rtol=1e-4, atol=1e-4)  # asum through the fallback reduction
    dbt = torch.zeros(Ci, device=dev)  # and with a row-contiguous A
    xt = x.permute(0, 2, 1).contiguous()  # [Bt, HW, Ci]
    d2 = torch.zeros(Ci, Co, device=dev)
    kern.gemm(kern.mat_plain(xt, 1, Ci, skb=HW * Ci, kfast=0), kern.mat_plain(dy, 1, HW, skb=Co * HW, kfast=1), d2,
              Ci, Co, HW, scr=Co, scc=1, nkb=Bt, splits=2, atomic=True, asum=dbt)
    torch.testing.assert_close(d2.cpu(), ref.t().cpu(), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(dbt.cpu(), x.sum((0, 2)).cpu(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("k,stride,pad,dil", [(3, 1, 1, 1), (7, 4, 3, 1), (3, 2, 1, 1), (2, 2, 0, 1), (5, 1, 2, 1)])
def test_conv_igemm_fwd_dgrad_wgrad(dev, k, stride, pad, dil):
    Bt, Ci, Co, H, W = 2, 5, 6, 13, 11
    x, w = rnd(Bt, Ci, H, W, dev=dev), rnd(Co, Ci, k, k, dev=dev, seed=1)
    Ho = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    ref = F.conv2d(xr, wr, None, stride=stride, padding=pad, dilation=dil)
    gy = rnd(Bt, Co, Ho, Wo, dev=dev, seed=2)
    ref.backward(gy)
    # forward: y[b] = W[Co, Ci*k*k] x im2col(x[b])
    y = torch.empty(Bt, Co, Ho, Wo, device=dev)
    Bm = kern.mat_im2col(x, sb=Ci * H * W, skb=0, sci=H * W, sy=W, sx=1, KH=k, KW=k, Pw=Wo, Hs=H, Ws=W,
                         stride=stride, pad=pad, dil=dil, patch_is_row=1, transposed=0, kfast=0)
    kern.gemm(kern.mat_plain(w, Ci * k * k, 1, kfast=1), Bm, y, Co, Ho * Wo, Ci * k * k, scr=Ho * Wo, scc=1,
              scb=Co * Ho * Wo, nbatch=Bt)
    torch.testing.assert_close(y.cpu(), ref.detach().cpu(), rtol=1e-4, atol=1e-4)
    # dgrad: dx[b] = Wt[Ci, Co*k*k] x gather(dy[b])
    wt = w.permute(1, 0, 2, 3).contiguous()
    dx = torch.empty_like(x)
    Bm = kern.mat_im2col(gy, sb=Co * Ho * Wo, skb=0, sci=Ho * Wo, sy=Wo, sx=1, KH=k, KW=k, Pw=W, Hs=Ho, Ws=Wo,
                         stride=stride, pad=pad, dil=dil, patch_is_row=1, transposed=1, kfast=0)
    kern.gemm(kern.mat_plain(wt, Co * k * k, 1, kfast=1), Bm, dx, Ci, H * W, Co * k * k, scr=H * W, scc=1,
              scb=Ci * H * W, nbatch=Bt)
    torch.testing.assert_close(dx.cpu(), xr.grad.cpu(), rtol=1e-4, atol=1e-4)
    # wgrad: dW[Co, Ci*k*k] += sum_b dy[b] x im2col(x[b])^T
    dw = torch.zeros_like(w)
    Bm = kern.mat_im2col(x, sb=0, skb=Ci * H * W, sci=H * W, sy=W, sx=1, KH=k, KW=k, Pw=Wo, Hs=H, Ws=W,
                         stride=stride, pad=pad, dil=dil, patch_is_row=0, transposed=0, kfast=1)
    kern.gemm(kern.mat_plain(gy, Ho * Wo, 1, skb=Co * Ho * Wo, kfast=1), Bm, dw, Co, Ci * k * k, Ho * Wo,
              scr=Ci * k * k, scc=1, nkb=Bt, splits=2, atomic=True)
    torch.testing.assert_close(dw.cpu(), wr.grad.cpu(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("M,N,K", [(200, 150, 100), (32, 300, 70), (130, 64, 129), (65, 33, 160)])
def test_gemm_tiles_fp32_and_bf16(dev, M, N, K):
    """larger shapes exercise the 128x128 / 32x256 tiles; bf16-operand mode is checked against bf16-rounded inputs."""
    x, w, b = rnd(M, K, dev=dev), rnd(N, K, dev=dev, seed=1), rnd(N, dev=dev, seed=2)
    y = torch.empty(M, N, device=dev)
    kern.gemm(kern.mat_plain(x, K, 1, kfast=1), kern.mat_plain(w, 1, K, kfast=1), y, M, N, K, scr=N, scc=1, bias=b)
    torch.testing.assert_close(y.cpu(), F.linear(x, w, b).cpu(), rtol=1e-4, atol=1e-4)
    # transposed operands (mfast staging): y2 = x^T-view GEMM
    xt = x.t().contiguous()  # [K, M]
    kern.gemm(kern.mat_plain(xt, 1, M, kfast=0), kern.mat_plain(w, 1, K, kfast=1), y, M, N, K, scr=N, scc=1, bias=b)
    torch.testing.assert_close(y.cpu(), F.linear(x, w, b).cpu(), rtol=1e-4, atol=1e-4)
    # bf16 tensors (throughput mode): k-contiguous quads, the row-pair form of a row-contiguous operand (M even) and its
    # scalar fallback (M odd), bf16 output with bias; then an atomic (fp32) epilogue
    xb, wb = x.bfloat16(), w.bfloat16()
    ref = F.linear(xb.float(), wb.float(), b).cpu()
    yb = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    kern.gemm(kern.mat_plain(xb, K, 1, kfast=1), kern.mat_plain(wb, 1, K, kfast=1), yb, M, N, K, scr=N, scc=1, bias=b)
    torch.testing.assert_close(yb.float().cpu(), ref, rtol=1e-2, atol=1e-2)
    xtb = xb.t().contiguous()
    kern.gemm(kern.mat_plain(xtb, 1, M, kfast=0), kern.mat_plain(wb, 1, K, kfast=1), yb, M, N, K, scr=N, scc=1, bias=b)
    torch.testing.assert_close(yb.float().cpu(), ref, rtol=1e-2, atol=1e-2)
    wtb = wb.t().contiguous()  # [K, N]: B row-contiguous
    kern.gemm(kern.mat_plain(xb, K, 1, kfast=1), kern.mat_plain(wtb, N, 1, kfast=0), yb, M, N, K, scr=N, scc=1, bias=b)
    torch.testing.assert_close(yb.float().cpu(), ref, rtol=1e-2, atol=1e-2)
    acc = torch.zeros(M, N, device=dev)
    kern.gemm(kern.mat_plain(xtb, 1, M, kfast=0), kern.mat_plain(wtb, N, 1, kfast=0), acc, M, N, K, scr=N, scc=1, splits=2,
              atomic=True)
    torch.testing.assert_close(acc.cpu(), F.linear(xb.float(), wb.float()).cpu(), rtol=1e-3, atol=1e-3)
    with pytest.raises(TypeError):  # mixed operand types are refused
        kern.gemm(kern.mat_plain(xb, K, 1, kfast=1), kern.mat_plain(w, 1, K, kfast=1), yb, M, N, K, scr=N, scc=1)


# LDS-DMA ring kernel (gemm_ring.h): taken for bf16 operands whose contiguous axis is 16-byte aligned (extents % 8 == 0)
# and M, N >= 48.  All four operand orientations; ragged tiles in M and N; K tails (K % 64 != 0), single K-step and chains
# longer than the ring; row-major bf16 store with bias / residual / per-row sample scale, and the fp32 atomic split-K form.
# (the last four: pitches / extents / K that are NOT multiples of 8 elements — 2-byte aligned LDS-DMA chunks, K tails
# inside a chunk masked at the fragment, the operand's final chunk fetched by hand instead of reading past its end)
@pytest.mark.parametrize("M,N,K", [(64, 64, 64), (200, 152, 72), (136, 264, 320), (56, 48, 8), (264, 136, 456),
                                   (100, 49, 49), (137, 196, 200), (50, 61, 20), (49, 49, 515), (56, 200, 72)])
@pytest.mark.parametrize("akf,bkf", [(1, 1), (1, 0), (0, 1), (0, 0)])
def test_gemm_ring_orientations(dev, M, N, K, akf, bkf):
    x, w, b = rnd(M, K, dev=dev).bfloat16(), rnd(N, K, dev=dev, seed=1).bfloat16(), rnd(N, dev=dev, seed=2)
    R = rnd(M, N, dev=dev, seed=3).bfloat16()
    xt = x.t().contiguous()
    wt = w.t().contiguous()
    A = kern.mat_plain(x, K, 1, kfast=1) if akf else kern.mat_plain(xt, 1, M, kfast=0)
    Bm = kern.mat_plain(w, 1, K, kfast=1) if bkf else kern.mat_plain(wt, N, 1, kfast=0)
    ref = F.linear(x.float(), w.float(), b) + R.float()
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    kern.gemm(A, Bm, y, M, N, K, scr=N, scc=1, bias=b, R=R, srr=N, src=1)
    torch.testing.assert_close(y.float().cpu(), ref.cpu(), rtol=1e-2, atol=2e-2)
    # transposed store (column-major C: the un-swapped accumulator layout), bias along rows
    yt = torch.empty(N, M, device=dev, dtype=torch.bfloat16)
    kern.gemm(A, Bm, yt, M, N, K, scr=1, scc=M)
    torch.testing.assert_close(yt.float().cpu(), F.linear(x.float(), w.float()).t().cpu(), rtol=1e-2, atol=2e-2)
    acc = torch.zeros(M, N, device=dev)
    rs = torch.zeros(M, device=dev)  # asum: row sums of A (the bias gradient of a weight-gradient contraction)
    kern.gemm(A, Bm, acc, M, N, K, scr=N, scc=1, splits=3, atomic=True, asum=rs)
    torch.testing.assert_close(acc.cpu(), F.linear(x.float(), w.float()).cpu(), rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(rs.cpu(), x.float().sum(1).cpu(), rtol=1e-4, atol=1e-3)
    # atomic=2: the unsplit product STORED as fp32 (no zero fill: the destination starts as garbage), row-major and strided C
    st = torch.full((M, N), 7.5e8, device=dev)
    kern.gemm(A, Bm, st, M, N, K, scr=N, scc=1, alpha=0.5, atomic=2)
    torch.testing.assert_close(st.cpu(), 0.5 * F.linear(x.float(), w.float()).cpu(), rtol=1e-3, atol=1e-3)
    stt = torch.full((N, M), -3.0e9, device=dev)
    kern.gemm(A, Bm, stt, M, N, K, scr=1, scc=M, atomic=2)
    torch.testing.assert_close(stt.cpu(), F.linear(x.float(), w.float()).t().cpu(), rtol=1e-3, atol=1e-3)
    with pytest.raises(Exception):
        kern.gemm(A, Bm, st, M, N, K, scr=N, scc=1, splits=2, atomic=2)


@pytest.mark.parametrize("Co,Ci,HW", [(64, 72, 80), (64, 72, 49), (56, 50, 196)])
def test_gemm_ring_batched_kbatch(dev, Co, Ci, HW):
    """per-image batches (grid z) with a per-sample scale, and the K-batch walk of the flat weight gradient; 7x7 / 14x14
    planes (49 / 196 pixels) are the unaligned NCHW operands of the two deepest decoder levels"""
    Bt = 3
    w = rnd(Co, Ci, dev=dev).bfloat16()
    x = rnd(Bt, Ci, HW, dev=dev, seed=1).bfloat16()
    bs = rnd(Bt, dev=dev, seed=2)
    y = torch.empty(Bt, Co, HW, device=dev, dtype=torch.bfloat16)
    kern.gemm(kern.mat_plain(w, Ci, 1, kfast=1), kern.mat_plain(x, HW, 1, sb=Ci * HW), y, Co, HW, Ci, scr=HW, scc=1,
              scb=Co * HW, nbatch=Bt, bscale=bs)
    ref = torch.matmul(w.float(), x.float()) * bs[:, None, None]
    torch.testing.assert_close(y.float().cpu(), ref.cpu(), rtol=1e-2, atol=2e-2)
    dy = rnd(Bt, Co, HW, dev=dev, seed=3).bfloat16()
    dw, db = torch.zeros(Co, Ci, device=dev), torch.zeros(Co, device=dev)
    kern.gemm(kern.mat_plain(dy, HW, 1, skb=Co * HW, kfast=1), kern.mat_plain(x, 1, HW, skb=Ci * HW, kfast=1), dw,
              Co, Ci, HW, scr=Ci, scc=1, nkb=Bt, splits=2, atomic=True, asum=db)
    torch.testing.assert_close(dw.cpu(), torch.einsum("bop,bip->oi", dy.float(), x.float()).cpu(), rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(db.cpu(), dy.float().sum((0, 2)).cpu(), rtol=1e-4, atol=1e-3)


# K = 64 / 128 with >= 768 output tiles: the persistent streaming variant of the ring GEMM (gemm_ring.h, round 4) — all four
# operand orientations, ragged edge tiles, the per-image batched form with bias / residual, and the flat form with the
# per-row DropPath scale
@pytest.mark.parametrize("akf,bkf", [(1, 1), (1, 0), (0, 1), (0, 0)])
@pytest.mark.parametrize("K", [64, 128])
def test_ring_stream_orientations(dev, akf, bkf, K):
    BF = torch.bfloat16
    M, N = 1656, 1912  # 26 x 30 = 780 tiles, both edges ragged, every pitch a multiple of 8
    a = rnd(M, K, dev=dev).to(BF)
    b = rnd(K, N, dev=dev, seed=1).to(BF)
    at, bt = a.t().contiguous(), b.t().contiguous()  # (kept alive: a MatT holds a raw pointer)
    A = kern.mat_plain(a, K, 1, kfast=1) if akf else kern.mat_plain(at, 1, M, kfast=0)
    B = kern.mat_plain(bt, 1, K, kfast=1) if bkf else kern.mat_plain(b, N, 1, kfast=0)
    y = torch.empty(M, N, device=dev, dtype=BF)
    kern.gemm(A, B, y, M, N, K, scr=N, scc=1)
    if not kern._lib.is_hostsim():
        assert kern.last_gemm_kernel().startswith("gemm_ring_stream_kernel"), kern.last_gemm_kernel()
    ref = a.float() @ b.float()
    torch.testing.assert_close(y.float().cpu(), ref.cpu(), rtol=2e-2, atol=2e-2 * ref.abs().max().item())
    assert (y.float() - ref).abs().mean().item() < 4e-3 * ref.abs().mean().item()


def test_ring_stream_batched_and_scaled_epilogues(dev):
    BF = torch.bfloat16
    # per-image 1x1 conv: W [M, K] . X_b [K, HW], bias on rows, residual
    nb, M, K, HW = 16, 64, 64, 3136
    w, x = rnd(M, K, dev=dev).to(BF), rnd(nb, K, HW, dev=dev, seed=1).to(BF)
    bias, R = rnd(M, dev=dev, seed=2), rnd(nb, M, HW, dev=dev, seed=3).to(BF)
    y = torch.empty(nb, M, HW, device=dev, dtype=BF)
    kern.gemm(kern.mat_plain(w, K, 1, kfast=1), kern.mat_plain(x, HW, 1, sb=K * HW), y, M, HW, K, scr=HW, scc=1, scb=M * HW,
              nbatch=nb, bias=bias, bias_on_row=True, R=R, srb=M * HW, srr=HW, src=1)
    ref = torch.matmul(w.float(), x.float()) + bias[None, :, None] + R.float()
    assert (y.float() - ref).abs().mean().item() < 4e-3 * ref.abs().mean().item()
    # flat Linear with the per-sample scale looked up by row
    Bt, n, N2, K2 = 8, 784, 128, 128
    xt, w2 = rnd(Bt, n, K2, dev=dev).to(BF), rnd(N2, K2, dev=dev, seed=1).to(BF)
    b2, R2, bs = rnd(N2, dev=dev, seed=2), rnd(Bt, n, N2, dev=dev, seed=3).to(BF), rnd(Bt, dev=dev, seed=4)
    y2 = torch.empty(Bt, n, N2, device=dev, dtype=BF)
    kern.gemm(kern.mat_plain(xt, K2, 1, kfast=1), kern.mat_plain(w2, 1, K2, kfast=1), y2, Bt * n, N2, K2, scr=N2, scc=1, bias=b2,
              bscale=bs, bscale_rows=n, R=R2, srr=N2, src=1)
    ref2 = (F.linear(xt.float(), w2.float(), b2)) * bs[:, None, None] + R2.float()
    assert (y2.float() - ref2).abs().mean().item() < 4e-3 * ref2.abs().mean().item()
