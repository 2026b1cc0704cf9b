"""Attention kernels of the BENCHED (bf16) mode at every head dimension and token count the three reference presets
produce, against plain PyTorch fp32 of the same formula on the same bf16-rounded inputs (forward and all three input
gradients).  Runs on the GPU only (full token counts).

  differential attention (multihead_diffattn.py:83-109), embed = 2C, hd = embed / H / 2:
      ACDC    heads 4/4/4   (acdc.sh:41-77):    hd 80 @196, 32 @784, 16 @3136
      Synapse heads 16/8/8  (synapse.sh:42-81): hd 20 @196, 16 @784,  8 @3136
      HAM     heads 2/2/2   (skin.sh:45-100):   hd 160 @196 (1024 @512^2), 64 @784, 32 @3136
  spatial-reduction attention (pvtv2.py:88-109): hd 64, 49 keys, 1/2/5/8 heads at 3136/784/196/49 queries
  non-local attention (nlb.py:117-138): channel-major, C = 64 @3136 (pair kernels), 128 @784 (materialised path)

Tolerances: inputs are exact bf16 values, the kernels compute in fp32 from bf16 operands and store bf16, so the error
budget is a few bf16 ulps of the result scale: relative L2 error < 2e-2 (outputs) / 3e-2 (gradients)."""
import pytest
import torch

from cenet_amd import kern, ops

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a.detach().float() - b.detach().float()).norm() / (b.detach().float().norm() + 1e-12))


def _bf(*shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g).bfloat16()


DIFF = [(196, 4, 80), (784, 4, 32), (3136, 4, 16), (196, 16, 20), (784, 8, 16), (3136, 8, 8), (196, 2, 160), (784, 2, 64),
        (3136, 2, 32), (1024, 2, 160), (4096, 2, 64)]


@pytest.mark.parametrize("N,H,hd", DIFF)
def test_diff_attention_heads_every_preset_head_dim(N, H, hd):
    dev = torch.device("cuda:0")
    B, E = 2, 2 * H * hd
    q, k, v = (_bf(B, N, E, seed=s + N + hd).to(dev) for s in range(3))
    go = _bf(B, 2 * H, N, 2 * hd, seed=7).to(dev)
    qs, ks, vs = (t.clone().requires_grad_(True) for t in (q, k, v))
    U = ops.diff_attention_heads(qs, ks, vs, H)
    assert U.dtype == torch.bfloat16
    U.backward(go)
    qf, kf, vf = (t.float().requires_grad_(True) for t in (q, k, v))
    qh = qf.view(B, N, 2 * H, hd).transpose(1, 2)
    kh = kf.view(B, N, 2 * H, hd).transpose(1, 2)
    vh = vf.view(B, N, H, 2 * hd).transpose(1, 2).repeat_interleave(2, dim=1)
    ref = torch.softmax(qh @ kh.transpose(-1, -2) * hd ** -0.5, dim=-1) @ vh
    ref.backward(go.float())
    assert _rel(U, ref) < 2e-2
    for a, b in ((qs.grad, qf.grad), (ks.grad, kf.grad), (vs.grad, vf.grad)):
        assert _rel(a, b) < 3e-2


@pytest.mark.parametrize("N,heads,Nk", [(3136, 1, 49), (784, 2, 49), (196, 5, 49), (49, 8, 49), (16384, 1, 256)])
def test_sr_attention_every_stage(N, heads, Nk):
    dev = torch.device("cuda:0")
    B, Cn = 2, 64 * heads
    q, kv = _bf(B, N, Cn, seed=N).to(dev), _bf(B, Nk, 2 * Cn, seed=N + 1).to(dev)
    go = _bf(B, N, Cn, seed=3).to(dev)
    qs, kvs = q.clone().requires_grad_(True), kv.clone().requires_grad_(True)
    o = ops.sr_attention(qs, kvs, heads)
    o.backward(go)
    qf, kvf = q.float().requires_grad_(True), kv.float().requires_grad_(True)
    qh = qf.view(B, N, heads, 64).transpose(1, 2)
    kh = kvf[..., :Cn].reshape(B, Nk, heads, 64).transpose(1, 2)
    vh = kvf[..., Cn:].reshape(B, Nk, heads, 64).transpose(1, 2)
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) * 64 ** -0.5, dim=-1) @ vh).transpose(1, 2).reshape(B, N, Cn)
    ref.backward(go.float())
    assert _rel(o, ref) < 2e-2
    assert _rel(qs.grad, qf.grad) < 3e-2 and _rel(kvs.grad, kvf.grad) < 3e-2


@pytest.mark.parametrize("Cn,N", [(64, 3136), (128, 784), (320, 196), (512, 49)])
def test_nonlocal_attention_every_level(Cn, N):
    dev = torch.device("cuda:0")
    B = 2
    th, ph, gx = (_bf(B, Cn, N, seed=s + Cn).to(dev) for s in range(3))
    go = _bf(B, Cn, N, seed=9).to(dev)
    ts, ps, gs = (t.clone().requires_grad_(True) for t in (th, ph, gx))
    y = ops.nonlocal_attention(ts, ps, gs)
    y.backward(go)
    tf, pf, gf = (t.float().requires_grad_(True) for t in (th, ph, gx))
    att = torch.softmax(tf.transpose(1, 2) @ pf * Cn ** -0.5, dim=-1)  # [B, i, j]
    ref = (att @ gf.transpose(1, 2)).transpose(1, 2)                  # y[b, c, i] = sum_j att[i, j] g[c, j]
    ref.backward(go.float())
    assert _rel(y, ref) < 2e-2
    for a, b in ((ts.grad, tf.grad), (ps.grad, pf.grad), (gs.grad, gf.grad)):
        assert _rel(a, b) < 3e-2


def test_pair_kernels_cover_the_benched_preset():
    """the ACDC / Synapse problems with hd in {8, 16, 32} run on attn_diff.hip, not on the fallback"""
    for hd, N in ((16, 3136), (32, 784), (8, 3136), (16, 784)):
        assert kern.diffattn_heads_supported(hd, N)
