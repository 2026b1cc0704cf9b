"""PVTv2 encoder on HIP kernels — mirrors reference src/networks/cenet/pvtv2.py (module names, ctor args,
state-dict keys).  torch.nn layers are used ONLY as parameter containers (same keys / init / deepcopy
behaviour as the reference); every forward runs on cenet_amd.ops."""
from __future__ import annotations

import math
from functools import partial

import torch
import torch.nn as nn

from ... import ops


def _init_weights(m):
    """Same distributions as reference pvtv2.py:24-38."""
    if isinstance(m, nn.Linear):
        nn.init.trunc_normal_(m.weight, std=.02)
        if m.bias is not None:
            nn.init.zeros_(m.bias)
    elif isinstance(m, nn.LayerNorm):
        nn.init.ones_(m.weight)
        nn.init.zeros_(m.bias)
    elif isinstance(m, nn.Conv2d):
        fan_out = m.kernel_size[0] * m.kernel_size[1] * m.out_channels // m.groups
        nn.init.normal_(m.weight, 0.0, math.sqrt(2.0 / fan_out))
        if m.bias is not None:
            nn.init.zeros_(m.bias)


class DropPath(nn.Module):
    """Stochastic depth (timm semantics). Produces the per-sample scale consumed by the GEMM epilogue."""

    def __init__(self, drop_prob: float = 0.0):
        super().__init__()
        self.drop_prob = drop_prob

    def sample_scale(self, batch: int, device):
        """[B] tensor keep_mask/keep_prob, or None when inactive (eval / p == 0)."""
        if not self.training or self.drop_prob == 0.0:
            return None
        keep = 1.0 - self.drop_prob
        return torch.empty(batch, device=device, dtype=torch.float32).bernoulli_(keep).div_(keep)


class DWConv(nn.Module):
    """pvtv2.py:359-370 (3x3 depthwise on tokens)."""

    def __init__(self, dim=768):
        super().__init__()
        self.dwconv = nn.Conv2d(dim, dim, 3, 1, 1, bias=True, groups=dim)


class Mlp(nn.Module):
    """pvtv2.py:12-47: fc1 -> DW3x3 -> GELU -> fc2."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.dwconv = DWConv(hidden_features)
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.apply(_init_weights)

    def forward(self, x, H, W, resid=None, bscale=None):
        h = ops.linear(x, self.fc1.weight, self.fc1.bias)
        h = ops.dwconv_tok(h, self.dwconv.dwconv.weight, self.dwconv.dwconv.bias, H, W, act="gelu")
        return ops.linear(h, self.fc2.weight, self.fc2.bias, resid=resid, bscale=bscale)


class Attention(nn.Module):
    """pvtv2.py:50-109: spatial-reduction attention."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0., sr_ratio=1):
        super().__init__()
        assert dim % num_heads == 0, f"dim {dim} should be divided by num_heads {num_heads}."
        self.dim, self.num_heads, self.sr_ratio = dim, num_heads, sr_ratio
        if qk_scale is not None:
            raise NotImplementedError("qk_scale override is not used by CENet")
        self.q = nn.Linear(dim, dim, bias=qkv_bias)
        self.kv = nn.Linear(dim, dim * 2, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        if sr_ratio > 1:
            self.sr = nn.Conv2d(dim, dim, kernel_size=sr_ratio, stride=sr_ratio)
            self.norm = nn.LayerNorm(dim)
        self.apply(_init_weights)

    def forward(self, x, H, W, resid=None, bscale=None):
        q, x = ops.linear(x, self.q.weight, self.q.bias, tap=True)  # the kv branch's gradient joins q's inside the dgrad GEMM
        if self.sr_ratio > 1:
            x_ = ops.sr_conv_ln(x, H, W, self.sr.weight, self.sr.bias, self.sr_ratio, self.norm.weight, self.norm.bias, self.norm.eps)
        else:
            x_ = x
        kv = ops.linear(x_, self.kv.weight, self.kv.bias)
        o = ops.sr_attention(q, kv, self.num_heads)
        return ops.linear(o, self.proj.weight, self.proj.bias, resid=resid, bscale=bscale)


class Block(nn.Module):
    """pvtv2.py:112-149: x += DropPath(Attn(LN(x))); x += DropPath(Mlp(LN(x))) — the residual add and the per-sample
    DropPath scale are fused into the proj / fc2 GEMM epilogues."""

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0.,
                 drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm, sr_ratio=1):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, sr_ratio=sr_ratio)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio))
        self.apply(_init_weights)

    def _scale(self, x):
        pend = getattr(self, "_dp_pending", None)
        if pend:  # sampled for the whole encoder in one shot (PyramidVisionTransformerImpr.forward_features)
            return pend.pop()
        if isinstance(self.drop_path, DropPath):
            return self.drop_path.sample_scale(x.shape[0], x.device)
        return None

    def forward(self, x, H, W):
        # layernorm_res hands x back for the skip connection so that both gradients meet inside the LN backward kernel
        y, xr = ops.layernorm_res(x, self.norm1.weight, self.norm1.bias, self.norm1.eps)
        x = self.attn(y, H, W, resid=xr, bscale=self._scale(x))
        m = self.mlp
        if ops.pvt_mlp_supported(x, m.fc1.out_features, H, W):
            # bf16 tokens at 56x56 / 28x28 (C = 64 / 128): norm2 + Mlp + DropPath + residual as ONE forward kernel (csrc/pvt_mlp.hip)
            return ops.pvt_mlp(x, H, W, self.norm2.weight, self.norm2.bias, self.norm2.eps, m.fc1.weight, m.fc1.bias,
                               m.dwconv.dwconv.weight, m.dwconv.dwconv.bias, m.fc2.weight, m.fc2.bias, self._scale(x))
        y, xr = ops.layernorm_res(x, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        x = self.mlp(y, H, W, resid=xr, bscale=self._scale(x))
        return x


class OverlapPatchEmbed(nn.Module):
    """pvtv2.py:152-191: strided conv (implicit GEMM writing token layout directly) + LayerNorm."""

    def __init__(self, img_size=224, patch_size=7, stride=4, in_chans=3, embed_dim=768):
        super().__init__()
        self.patch_size, self.stride = patch_size, stride
        self.in_chans = in_chans
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=stride, padding=patch_size // 2)
        self.norm = nn.LayerNorm(embed_dim)
        self.apply(_init_weights)

    def forward(self, x, tokens=None):
        """tokens: the same map in token layout [B, H*W, C] (the previous stage's normed output, of which x is the NCHW
        copy made for the decoder): bf16 3x3 embeddings read it instead, as patch rows + plain GEMMs (ops.conv2d_tok)."""
        H, W = x.shape[2:]
        k, s, p = self.patch_size, self.stride, self.patch_size // 2
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        if tokens is not None and k == 3 and tokens.dtype == torch.bfloat16:
            t = ops.conv2d_tok(tokens, H, W, self.proj.weight, self.proj.bias, stride=s, pad=p, out_layout="tok")
            t = ops.layernorm(t, self.norm.weight, self.norm.bias, self.norm.eps)
            return t, Ho, Wo
        expand = self.in_chans if (x.shape[1] == 1 and self.in_chans > 1) else 0  # net.py:55 without materialising cat
        if expand and x.dtype == torch.bfloat16:
            # throughput mode: a conv over `expand` identical copies of one channel is the conv of that channel with the
            # channel-summed weight — a third of the reduction length (K = 49 instead of 147 for the 7x7 stem); the weight
            # gradient is added back into all three channel slices by ops.ChanSumWeightFn.  (fp32 parity mode keeps the reference's operation order.)
            t = ops.conv2d_nchw(x, ops.chan_sum_weight(self.proj.weight), self.proj.bias, stride=s, pad=p, out_layout="tok")
        else:
            t = ops.conv2d_nchw(x, self.proj.weight, self.proj.bias, stride=s, pad=p, out_layout="tok", expand_channels=expand)
        t = ops.layernorm(t, self.norm.weight, self.norm.bias, self.norm.eps)
        return t, Ho, Wo


class PyramidVisionTransformerImpr(nn.Module):
    """pvtv2.py:194-356."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dims=[64, 128, 256, 512],
                 num_heads=[1, 2, 4, 8], mlp_ratios=[4, 4, 4, 4], qkv_bias=False, qk_scale=None, drop_rate=0.,
                 attn_drop_rate=0., drop_path_rate=0., norm_layer=nn.LayerNorm, depths=[3, 4, 6, 3],
                 sr_ratios=[8, 4, 2, 1]):
        super().__init__()
        self.depths = depths
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, sum(depths))]
        cur = 0
        for s in range(4):
            pe = OverlapPatchEmbed(img_size=img_size // (1 if s == 0 else 2 ** (s + 1)), patch_size=7 if s == 0 else 3,
                                   stride=4 if s == 0 else 2, in_chans=in_chans if s == 0 else embed_dims[s - 1],
                                   embed_dim=embed_dims[s])
            setattr(self, f"patch_embed{s + 1}", pe)
        for s in range(4):
            blocks = nn.ModuleList([
                Block(dim=embed_dims[s], num_heads=num_heads[s], mlp_ratio=mlp_ratios[s], qkv_bias=qkv_bias,
                      qk_scale=qk_scale, drop_path=dpr[cur + i], norm_layer=norm_layer, sr_ratio=sr_ratios[s])
                for i in range(depths[s])])
            setattr(self, f"block{s + 1}", blocks)
            setattr(self, f"norm{s + 1}", norm_layer(embed_dims[s]))
            cur += depths[s]
        self.apply(_init_weights)

    def reset_drop_path(self, drop_path_rate):
        """pvtv2.py:272-288."""
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, sum(self.depths))]
        cur = 0
        for s in range(4):
            for i, blk in enumerate(getattr(self, f"block{s + 1}")):
                if isinstance(blk.drop_path, DropPath):
                    blk.drop_path.drop_prob = dpr[cur + i]
                elif dpr[cur + i] > 0:
                    blk.drop_path = DropPath(dpr[cur + i])
            cur += self.depths[s]

    def set_drop_path_masks(self, masks):
        """Inject the stochastic-depth draws of the NEXT training forward: {(stage, i): (keep_attn[B], keep_mlp[B])} of 0/1
        keep masks (what `bernoulli_(keep_prob)` returns inside timm's DropPath, pvtv2.py:123,145-149); they are divided by
        keep_prob here as DropPath does.  Blocks without an entry (rate 0 -> Identity) are left alone.  Used by the parity
        tests (tests/test_model_parity.py) — a training run samples its own masks."""
        self._dp_injected = masks

    def _sample_drop_path(self, batch, device):
        """All stochastic-depth scales of one forward pass from ONE uniform draw (two per block: attention and MLP branch)
        instead of two tiny RNG launches per use; same distribution as DropPath.sample_scale."""
        inj = getattr(self, "_dp_injected", None)
        if inj is not None:
            self._dp_injected = None
            for (s, i), (ma, mm) in inj.items():
                b = getattr(self, f"block{s + 1}")[i]
                keep = 1.0 - b.drop_path.drop_prob
                b._dp_pending = [(mm.to(device=device, dtype=torch.float32) / keep).contiguous(),
                                 (ma.to(device=device, dtype=torch.float32) / keep).contiguous()]
            return
        blocks = [b for s in range(4) for b in getattr(self, f"block{s + 1}")
                  if isinstance(b.drop_path, DropPath) and b.drop_path.training and b.drop_path.drop_prob > 0.0]
        if not blocks:
            return
        probs = tuple(1.0 - b.drop_path.drop_prob for b in blocks for _ in range(2))
        cache = getattr(self, "_dp_keep", None)
        if cache is None or cache[0] != probs or cache[1].device != device:  # uploaded once, not per step
            cache = self._dp_keep = (probs, torch.tensor(probs, device=device))
        keep = cache[1]
        scale = (torch.rand(keep.numel(), batch, device=device) < keep[:, None]).float() / keep[:, None]
        for i, b in enumerate(blocks):
            b._dp_pending = [scale[2 * i + 1], scale[2 * i]]  # popped from the end: attention first, then MLP

    def forward_features(self, x):
        outs = []
        if self.training:
            self._sample_drop_path(x.shape[0], x.device)
        cuts = None
        if getattr(self, "segment_cuts", None) is not None and self.training and torch.is_grad_enabled():
            cuts = self.segment_cuts
            del cuts[:]
        t = None
        for s in range(4):
            t, H, W = getattr(self, f"patch_embed{s + 1}")(x, t)  # (positional: module backward hooks only see positional inputs)
            for blk in getattr(self, f"block{s + 1}"):
                t = blk(t, H, W)
            n = getattr(self, f"norm{s + 1}")
            t = ops.layernorm(t, n.weight, n.bias, n.eps)
            if cuts is not None and t.requires_grad:
                # segmented backward (cenet_amd.graph.SegmentedStep): the stage's token output is cut out of the autograd graph;
                # its consumers (the decoder's NCHW copy, the next stage's patch embedding) read a leaf whose .grad collects
                # their gradients, and the stage itself is differentiated later by autograd.backward([t], [leaf.grad])
                leaf = t.detach().requires_grad_(True)
                cuts.append((t, leaf))
                t = leaf
            # (t has a second consumer, the next stage's patch embedding: it reads the tap so that its gradient joins the decoder's
            # inside the transpose of this node's backward)
            x, t = ops.tok_to_nchw(t, H, W, tap=True)
            outs.append(x)
        return outs

    def forward(self, x):
        return self.forward_features(x)


def _pvt(depths, embed_dims=(64, 128, 320, 512), mlp_ratios=(8, 8, 4, 4)):
    return PyramidVisionTransformerImpr(patch_size=4, embed_dims=list(embed_dims), num_heads=[1, 2, 5, 8],
                                        mlp_ratios=list(mlp_ratios), qkv_bias=True,
                                        norm_layer=partial(nn.LayerNorm, eps=1e-6), depths=list(depths),
                                        sr_ratios=[8, 4, 2, 1], drop_rate=0.0, drop_path_rate=0.1)


def pvt_v2_b0(**kw):
    return _pvt([2, 2, 2, 2], embed_dims=(32, 64, 160, 256))


def pvt_v2_b1(**kw):
    return _pvt([2, 2, 2, 2])


def pvt_v2_b2(**kw):
    """pvtv2.py:401-406."""
    return _pvt([3, 4, 6, 3])


def pvt_v2_b3(**kw):
    return _pvt([3, 4, 18, 3])


def pvt_v2_b4(**kw):
    return _pvt([3, 8, 27, 3])


def pvt_v2_b5(**kw):
    return _pvt([3, 6, 40, 3], mlp_ratios=(4, 4, 4, 4))
