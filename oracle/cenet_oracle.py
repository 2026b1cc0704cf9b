"""CPU ORACLE (TEST INFRASTRUCTURE — not product code).

A functional, plain-PyTorch fp32 restatement of the reference CENet hot path
(forward + Dice/CE loss; backward comes from torch autograd over these functions).
It operates directly on a *reference-keyed state dict* (the 801 keys of
`networks.CENet.state_dict()`), so the same weights can be
fed to the reference, to this oracle and to the HIP product path.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import
this file, and only as the checker / reported CPU baseline — never as the shipped path.

Parity pinning: `tests/test_oracle_golden.py` checks every function here against
golden vectors produced by the *imported, unmodified reference* in the build
container (`oracle/gen_golden.py`, fixtures under `tests/golden/`).
Third-party arithmetic not under /root/reference: medpy==0.5.2 `binary.dc`, `binary.jc`,
`binary.hd95`, `binary.assd` (restated in `dice_metric`, `jaccard_metric`, `surface_distances`,
`hd95_metric`, `assd_metric` from medpy's published algorithm on the scipy.ndimage calls it is
built on; parity unpinned — medpy is absent and the reference holds no test for it); timm==1.0.16 DropPath (restated in `_drop_path`).

All file:line citations are relative to /root/reference/src/.
"""
from __future__ import annotations

import math

import numpy as np
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# --------------------------------------------------------------------------------------
# configuration (networks/cenet/net.py:9-22, pvtv2.py:401-406, decoders.py:36-88)
# --------------------------------------------------------------------------------------
@dataclass
class CENetConfig:
    input_channels: int = 1
    num_classes: int = 4
    scale_factors: Sequence[float] = (1.0, 0.5)
    diffatt_num_heads: Sequence[int] = (4, 4, 4)
    # PVTv2-b2 (pvtv2.py:401-406)
    embed_dims: Sequence[int] = (64, 128, 320, 512)
    num_heads: Sequence[int] = (1, 2, 5, 8)
    mlp_ratios: Sequence[int] = (8, 8, 4, 4)
    depths: Sequence[int] = (3, 4, 6, 3)
    sr_ratios: Sequence[int] = (8, 4, 2, 1)
    drop_path_rate: float = 0.1
    # decoder (decoders.py:64,78,82,86): rates listed for levels 4,3,2,1 ; diff-attn depths for 3,2,1
    mca_rates: Dict[int, Sequence[int]] = field(
        default_factory=lambda: {4: (1, 2, 2), 3: (1, 2, 3), 2: (1, 2, 4), 1: (2, 3, 5)})
    dseb_depth: Dict[int, int] = field(default_factory=lambda: {3: 4, 2: 3, 1: 2})


def drop_path_rates(cfg: CENetConfig) -> List[float]:
    """pvtv2.py:214 — linspace(0, rate, sum(depths))."""
    n = sum(cfg.depths)
    return [x.item() for x in torch.linspace(0, cfg.drop_path_rate, n)]


# --------------------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------------------
def _bn(sd: SD, p: str, x: Tensor, training: bool, eps: float = 1e-5) -> Tensor:
    """nn.BatchNorm{1,2}d forward; in training updates running stats in `sd` in place (momentum 0.1)."""
    if training:
        sd[p + ".num_batches_tracked"] += 1
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        training, 0.1, eps)


def _drop_path(x: Tensor, mask: Optional[Tensor], rate: float) -> Tensor:
    """timm DropPath: per-sample keep mask scaled by 1/keep. `mask` is the [B] 0/1 keep tensor (None = identity)."""
    if mask is None or rate == 0.0:
        return x
    keep = 1.0 - rate
    return x * (mask.view(-1, *([1] * (x.ndim - 1))) / keep)


# --------------------------------------------------------------------------------------
# PVTv2 encoder (networks/cenet/pvtv2.py)
# --------------------------------------------------------------------------------------
def overlap_patch_embed(sd: SD, p: str, x: Tensor, k: int, stride: int):
    """pvtv2.py:185-191 — conv(k, stride, pad k//2) -> flatten -> LayerNorm(eps 1e-5)."""
    x = F.conv2d(x, sd[p + ".proj.weight"], sd[p + ".proj.bias"], stride=stride, padding=k // 2)
    H, W = x.shape[2:]
    x = x.flatten(2).transpose(1, 2)
    x = F.layer_norm(x, (x.shape[-1],), sd[p + ".norm.weight"], sd[p + ".norm.bias"], 1e-5)
    return x, H, W


def sr_attention(sd: SD, p: str, x: Tensor, H: int, W: int, heads: int, sr: int) -> Tensor:
    """pvtv2.py:88-109 — spatial-reduction attention."""
    B, N, C = x.shape
    hd = C // heads
    q = F.linear(x, sd[p + ".q.weight"], sd[p + ".q.bias"]).reshape(B, N, heads, hd).permute(0, 2, 1, 3)
    if sr > 1:
        x_ = x.permute(0, 2, 1).reshape(B, C, H, W)
        x_ = F.conv2d(x_, sd[p + ".sr.weight"], sd[p + ".sr.bias"], stride=sr).reshape(B, C, -1).permute(0, 2, 1)
        x_ = F.layer_norm(x_, (C,), sd[p + ".norm.weight"], sd[p + ".norm.bias"], 1e-5)
    else:
        x_ = x
    kv = F.linear(x_, sd[p + ".kv.weight"], sd[p + ".kv.bias"]).reshape(B, -1, 2, heads, hd).permute(2, 0, 3, 1, 4)
    k, v = kv[0], kv[1]
    attn = (q @ k.transpose(-2, -1)) * (hd ** -0.5)
    attn = attn.softmax(dim=-1)
    x = (attn @ v).transpose(1, 2).reshape(B, N, C)
    return F.linear(x, sd[p + ".proj.weight"], sd[p + ".proj.bias"])


def pvt_mlp(sd: SD, p: str, x: Tensor, H: int, W: int) -> Tensor:
    """pvtv2.py:40-47,364-370 — fc1 -> DW3x3(+bias) -> GELU(erf) -> fc2."""
    B, N, _ = x.shape
    x = F.linear(x, sd[p + ".fc1.weight"], sd[p + ".fc1.bias"])
    Ch = x.shape[-1]
    x = x.transpose(1, 2).reshape(B, Ch, H, W)
    x = F.conv2d(x, sd[p + ".dwconv.dwconv.weight"], sd[p + ".dwconv.dwconv.bias"], padding=1, groups=Ch)
    x = x.flatten(2).transpose(1, 2)
    x = F.gelu(x)
    return F.linear(x, sd[p + ".fc2.weight"], sd[p + ".fc2.bias"])


def pvt_block(sd: SD, p: str, x: Tensor, H: int, W: int, heads: int, sr: int,
              dp_rate: float, dp_masks) -> Tensor:
    """pvtv2.py:145-149 — pre-LN residual x2 (LN eps 1e-6, pvtv2.py:405) with DropPath."""
    C = x.shape[-1]
    m1, m2 = dp_masks if dp_masks is not None else (None, None)
    y = F.layer_norm(x, (C,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], 1e-6)
    x = x + _drop_path(sr_attention(sd, p + ".attn", y, H, W, heads, sr), m1, dp_rate)
    y = F.layer_norm(x, (C,), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], 1e-6)
    x = x + _drop_path(pvt_mlp(sd, p + ".mlp", y, H, W), m2, dp_rate)
    return x


def pvt_backbone(sd: SD, p: str, x: Tensor, cfg: CENetConfig, drop_masks=None) -> List[Tensor]:
    """pvtv2.py:312-348 — 4 stages; returns NCHW maps x1..x4.

    drop_masks: None (no stochastic depth; == eval or reset_drop_path(0)) or a dict
    {(stage, i): (mask_attn[B], mask_mlp[B])} of 0/1 keep tensors (training with injected masks).
    """
    B = x.shape[0]
    rates = drop_path_rates(cfg)
    outs = []
    cur = 0
    for s in range(4):
        k, stride = (7, 4) if s == 0 else (3, 2)
        x, H, W = overlap_patch_embed(sd, f"{p}.patch_embed{s + 1}", x, k, stride)
        for i in range(cfg.depths[s]):
            masks = None if drop_masks is None else drop_masks.get((s, i))
            x = pvt_block(sd, f"{p}.block{s + 1}.{i}", x, H, W, cfg.num_heads[s], cfg.sr_ratios[s],
                          rates[cur + i], masks)
        cur += cfg.depths[s]
        C = x.shape[-1]
        x = F.layer_norm(x, (C,), sd[f"{p}.norm{s + 1}.weight"], sd[f"{p}.norm{s + 1}.bias"], 1e-6)
        x = x.reshape(B, H, W, C).permute(0, 3, 1, 2).contiguous()
        outs.append(x)
    return outs


# --------------------------------------------------------------------------------------
# CFAM decoder block (networks/cenet/modules/cfam.py, blocks.py:131-185, nlb.py:102-148)
# --------------------------------------------------------------------------------------
def ccu(sd: SD, p: str, x: Tensor, training: bool) -> Tensor:
    """cfam.py:251-264 — per-(b,c) [max, mean, std(biased)] -> grouped conv1d k3 -> ReLU -> k1 -> BN1d(if B>1) -> sigmoid."""
    b, c = x.shape[:2]
    flat = x.reshape(b, c, -1)
    u = torch.stack([flat.max(dim=2)[0], flat.mean(dim=2), flat.std(dim=2, unbiased=False)], dim=-1)
    z = F.conv1d(u, sd[p + ".fc1.weight"], None, groups=c)
    z = F.conv1d(F.relu(z), sd[p + ".fc2.weight"], None, groups=c).view(b, c)
    if b > 1:
        z = _bn(sd, p + ".bn", z, training)
    return x * torch.sigmoid(z).reshape(b, c, 1, 1)


def sep_conv_bn(sd: SD, p: str, x: Tensor, rate: int, training: bool) -> Tensor:
    """blocks.py:169-185 with depth_activation=True, eps=1e-5 (cfam.py:197-207)."""
    c = x.shape[1]
    x = F.conv2d(x, sd[p + ".depthwise.weight"], None, padding=rate, dilation=rate, groups=c)
    x = F.relu(_bn(sd, p + ".depthwise_bn", x, training))
    x = F.conv2d(x, sd[p + ".pointwise.weight"], None)
    return F.relu(_bn(sd, p + ".pointwise_bn", x, training))


def multi_order_dwconv(sd: SD, p: str, x: Tensor, rates: Sequence[int], training: bool) -> Tensor:
    """cfam.py:227-241 — 5/16,5/16,5/16,1/16 channel split (cfam.py:178-190)."""
    C, H, W = x.shape[1:]
    g, q = int(5 / 16 * C), int(1 / 16 * C)
    bounds = [(0, g), (g, 2 * g), (2 * g, 3 * g), (3 * g, 3 * g + q)]
    ys = []
    for j, r in enumerate(rates):
        lo, hi = bounds[j]
        ys.append(sep_conv_bn(sd, f"{p}.dlps.{j}", x[:, lo:hi], r, training))
    lo, hi = bounds[3]
    y = F.adaptive_avg_pool2d(x[:, lo:hi], (7, 7))
    y = F.conv2d(y, sd[f"{p}.dlps.3.1.weight"], None)
    y = F.leaky_relu(_bn(sd, f"{p}.dlps.3.2", y, training), 0.01)
    y = F.interpolate(y, scale_factor=7, mode="bilinear", align_corners=True)  # nn.UpsamplingBilinear2d
    if y.shape[2] != H or y.shape[3] != W:
        y = F.interpolate(y, size=(H, W), mode="bilinear", align_corners=False)
    ys.append(y)
    x = torch.cat(ys, dim=1)
    return F.conv2d(x, sd[p + ".PW_conv.weight"], sd[p + ".PW_conv.bias"])


def _query_rows(n: int) -> int:
    """Attention maps are materialised as the reference does (multihead_diffattn.py:96-116, nlb.py:121-138) up to 4 096
    positions — every size the reference itself can run.  Beyond (512x512 inputs: 16 384 positions, SURVEY.md §7: the
    reference fails there) the QUERY rows are processed 2 048 at a time: softmax is row-wise, so each row sees exactly the
    reference's arithmetic while the host needs 0.5 GB instead of 17 GB."""
    return n if n <= 4096 else 2048


def nonlocal_block(sd: SD, p: str, x: Tensor, training: bool) -> Tensor:
    """nlb.py:102-148 — softmax(theta^T phi / sqrt(C)) g ; conv_out ; BN ; (1-w)x + w p."""
    N, C, H, W = x.shape
    theta = F.conv2d(x, sd[p + ".conv_theta.weight"], sd[p + ".conv_theta.bias"]).view(N, C, -1)
    phi = F.conv2d(x, sd[p + ".conv_phi.weight"], sd[p + ".conv_phi.bias"]).view(N, C, -1)
    g = F.conv2d(x, sd[p + ".conv_g.weight"], sd[p + ".conv_g.bias"]).view(N, C, -1)
    HW = H * W
    ys = []
    for h0 in range(0, HW, _query_rows(HW)):  # (one chunk below 4 096 positions: the reference's single product)
        a = torch.einsum("nch,ncp->nhp", theta[:, :, h0:h0 + _query_rows(HW)], phi) * (C ** -0.5)
        a = a.softmax(dim=2)
        ys.append(torch.einsum("nhg,ncg->nch", a, g))
    y = (ys[0] if len(ys) == 1 else torch.cat(ys, dim=2)).view(N, C, H, W)
    pout = F.conv2d(y, sd[p + ".conv_out.weight"], sd[p + ".conv_out.bias"])
    pout = _bn(sd, p + ".bn", pout, training)
    w = sd[p + ".w"]
    return (1 - w) * x + w * pout


def mca(sd: SD, p: str, x: Tensor, rates: Sequence[int], training: bool) -> Tensor:
    """cfam.py:298-306."""
    shortcut = x
    x = ccu(sd, p + ".ccu", x, training)
    g = F.conv2d(x, sd[p + ".gate.weight"], sd[p + ".gate.bias"])
    v = multi_order_dwconv(sd, p + ".value", x, rates, training)
    x = F.conv2d(F.silu(g) * F.silu(v), sd[p + ".proj_2.weight"], sd[p + ".proj_2.bias"])
    x = x + shortcut
    return nonlocal_block(sd, p + ".denoising_module", x, training)


def srm(sd: SD, p: str, x: Tensor, training: bool) -> Tensor:
    """cfam.py:93-101 — channel-wise [max, mean, std(unbiased)] -> 1x1 + 3x3 -> GELU -> BN(1) -> sigmoid gate."""
    u = torch.cat([x.max(1, keepdim=True)[0], x.mean(1, keepdim=True), x.std(1, keepdim=True)], dim=1)
    f = F.gelu(F.conv2d(u, sd[p + ".pwc.weight"]) + F.conv2d(u, sd[p + ".dwc.weight"], padding=1))
    f = _bn(sd, p + ".bn", f, training)
    return x * torch.sigmoid(f)


def cfam_mlp(sd: SD, p: str, x: Tensor, training: bool) -> Tensor:
    """cfam.py:149-159."""
    x = F.conv2d(x, sd[p + ".fc1.weight"], sd[p + ".fc1.bias"])
    x = F.conv2d(x, sd[p + ".dwconv.weight"], sd[p + ".dwconv.bias"], padding=1, groups=x.shape[1])
    x = F.gelu(x)
    x = srm(sd, p + ".srm", x, training)
    return F.conv2d(x, sd[p + ".fc2.weight"], sd[p + ".fc2.bias"])


def cfa_module(sd: SD, p: str, x: Tensor, rates: Sequence[int], training: bool) -> Tensor:
    """cfam.py:365-374 (drop_path is Identity: rate 0, decoders.py:67)."""
    x = x + sd[p + ".layer_scale_1"] * mca(sd, p + ".mca", _bn(sd, p + ".norm1", x, training), rates, training)
    x = x + sd[p + ".layer_scale_2"] * cfam_mlp(sd, p + ".mlp", _bn(sd, p + ".norm2", x, training), training)
    return x


# --------------------------------------------------------------------------------------
# EUCB / UpConv / DSEB (blocks.py, dseb.py, multihead_diffattn.py, rms_norm.py)
# --------------------------------------------------------------------------------------
def eucb(sd: SD, p: str, x: Tensor, training: bool) -> Tensor:
    """blocks.py:317-321 — nearest x2 -> DW3x3 -> BN -> LeakyReLU(0.2) -> (shuffle == identity) -> 1x1+bias."""
    c = x.shape[1]
    x = F.interpolate(x, scale_factor=2, mode="nearest")
    x = F.conv2d(x, sd[p + ".up_dwc.1.weight"], None, padding=1, groups=c)
    x = F.leaky_relu(_bn(sd, p + ".up_dwc.2", x, training), 0.2)
    return F.conv2d(x, sd[p + ".pwc.0.weight"], sd[p + ".pwc.0.bias"])


def up_conv(sd: SD, p: str, x: Tensor, training: bool) -> Tensor:
    """blocks.py:220-221 — bilinear(align_corners=True) x2 -> 3x3 -> BN -> LeakyReLU(0.2)."""
    x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    x = F.conv2d(x, sd[p + ".up.1.weight"], None, padding=1)
    return F.leaky_relu(_bn(sd, p + ".up.2", x, training), 0.2)


def fea(sd: SD, p: str, x: Tensor, scales: Sequence[float]) -> Tensor:
    """dseb.py:63-76,40-50 — e_s = |x - up(down_s(x))| ; edge = mean_{i<j} |e_i - e_j| ; x + w*edge."""
    H, W = x.shape[2:]
    edges = []
    for s in scales:
        d = F.interpolate(x, scale_factor=s, mode="bilinear")
        edges.append(torch.abs(x - F.interpolate(d, size=(H, W), mode="bilinear")))
    n = len(scales)
    m = n * (n - 1) // 2
    edge = 0
    for i in range(n):
        for j in range(i + 1, n):
            edge = edge + (1.0 / m) * torch.abs(edges[i] - edges[j])
    return x + sd[p + ".w"] * edge


def lambda_init_fn(depth: int) -> float:
    """multihead_diffattn.py:28-29."""
    return 0.8 - 0.6 * math.exp(-0.3 * depth)


def multihead_diff_attn(sd: SD, p: str, x: Tensor, num_heads: int, depth: int) -> Tensor:
    """multihead_diffattn.py:70-129 (no mask, no rotary, n_rep=1); RMSNorm fallback rms_norm.py:15-22."""
    B, N, E = x.shape
    hd = E // num_heads // 2
    lam0 = lambda_init_fn(depth)
    q = F.linear(x, sd[p + ".q_proj.weight"]).view(B, N, 2 * num_heads, hd).transpose(1, 2)
    k = F.linear(x, sd[p + ".k_proj.weight"]).view(B, N, 2 * num_heads, hd).transpose(1, 2)
    v = F.linear(x, sd[p + ".v_proj.weight"]).view(B, N, num_heads, 2 * hd).transpose(1, 2)
    q = q * hd ** -0.5
    l1 = torch.exp(torch.sum(sd[p + ".lambda_q1"] * sd[p + ".lambda_k1"], dim=-1).float())
    l2 = torch.exp(torch.sum(sd[p + ".lambda_q2"] * sd[p + ".lambda_k2"], dim=-1).float())
    lam = l1 - l2 + lam0
    os_ = []
    for n0 in range(0, N, _query_rows(N)):  # (one chunk below 4 096 tokens: the reference's single product)
        qc = q[:, :, n0:n0 + _query_rows(N)]
        a = torch.nan_to_num(qc @ k.transpose(-1, -2))
        a = F.softmax(a, dim=-1, dtype=torch.float32).type_as(a)
        a = a.view(B, num_heads, 2, qc.shape[2], N)
        a = a[:, :, 0] - lam * a[:, :, 1]
        os_.append(a @ v)
    o = os_[0] if len(os_) == 1 else torch.cat(os_, dim=2)
    o = o * torch.rsqrt(o.pow(2).mean(-1, keepdim=True) + 1e-5)  # RMSNorm(2hd, eps 1e-5, no affine)
    o = o * (1 - lam0)
    o = o.transpose(1, 2).reshape(B, N, E)
    return F.linear(o, sd[p + ".out_proj.weight"])


def dse_block(sd: SD, p: str, skip: Tensor, dec: Tensor, scales, heads: int, depth: int) -> Tensor:
    """dseb.py:153-165 with mode='cat', use_command='dat-fea'; apply_diffattn dseb.py:114-118.

    The `view`s reinterpret the flat NCHW buffer; the reference divides by the hard-wired
    `input_size` — identical to view(B, 2C, H, W) at 224x224."""
    y = torch.cat([dec, skip], dim=1).contiguous()
    B, C2, H, W = y.shape
    x_fea = fea(sd, p + ".boundary", y, scales) + y
    tok = y.view(B, -1, C2)
    diff = multihead_diff_attn(sd, p + ".diffattn", tok, heads, depth).view(B, C2, H, W)
    z = x_fea + diff * y
    return F.conv2d(z, sd[p + ".mixer.weight"], None) + skip


def decoder(sd: SD, p: str, x4: Tensor, skips: Sequence[Tensor], cfg: CENetConfig, training: bool) -> Tensor:
    """decoders.py:90-105."""
    d = cfa_module(sd, f"{p}.dec4", x4, cfg.mca_rates[4], training)
    for lvl, skip, head in zip((3, 2, 1), skips, cfg.diffatt_num_heads):
        d = eucb(sd, f"{p}.up{lvl}", d, training)
        s = dse_block(sd, f"{p}.skip_enhancer{lvl}", skip, d, cfg.scale_factors, head, cfg.dseb_depth[lvl])
        d = cfa_module(sd, f"{p}.dec{lvl}", d + s, cfg.mca_rates[lvl], training)
    return d


# --------------------------------------------------------------------------------------
# OutHead (out.py:69-75, unet.py:201-214,380-381)
# --------------------------------------------------------------------------------------
def unet_res_block(sd: SD, p: str, x: Tensor, k: int, training: bool) -> Tensor:
    """unet.py:201-214 — conv-BN-LReLU(.01)-conv-BN (+ 1x1 conv+BN residual if channels differ) -> add -> LReLU."""
    pad = k // 2  # monai get_padding: (k - 1 + 1) // 2
    out = F.conv2d(x, sd[p + ".conv1.conv.weight"], None, padding=pad)
    out = F.leaky_relu(_bn(sd, p + ".norm1", out, training), 0.01)
    out = F.conv2d(out, sd[p + ".conv2.conv.weight"], None, padding=pad)
    out = _bn(sd, p + ".norm2", out, training)
    res = x
    if (p + ".conv3.conv.weight") in sd:
        res = _bn(sd, p + ".norm3", F.conv2d(x, sd[p + ".conv3.conv.weight"], None), training)
    return F.leaky_relu(out + res, 0.01)


def out_head(sd: SD, p: str, dec: Tensor, x: Tensor, training: bool) -> Tensor:
    """out.py:69-75 with merge_mode='cat', up_block='upcn'."""
    rb = F.max_pool2d(unet_res_block(sd, p + ".rb.0", x, 5, training), 2, 2)
    rb = sd[p + ".w"] * rb
    d = up_conv(sd, p + ".up", dec, training)
    z = torch.cat([d, rb], dim=1)
    y = unet_res_block(sd, p + ".out.0", z, 3, training)
    y = F.conv2d(y, sd[p + ".out.1.conv.conv.weight"], sd[p + ".out.1.conv.conv.bias"])
    return F.interpolate(y, scale_factor=2, mode="bilinear")


# --------------------------------------------------------------------------------------
# whole network (net.py:53-64)
# --------------------------------------------------------------------------------------
def cenet_forward(sd: SD, x: Tensor, cfg: CENetConfig, training: bool = False, drop_masks=None) -> Tensor:
    """CENet.forward. In training mode BN running stats inside `sd` are updated in place."""
    y = torch.cat([x, x, x], dim=1) if x.shape[1] == 1 else x
    x1, x2, x3, x4 = pvt_backbone(sd, "backbone", y, cfg, drop_masks if training else None)
    dec = decoder(sd, "decoder", x4, [x3, x2, x1], cfg, training)
    return out_head(sd, "out", dec, x, training)


# --------------------------------------------------------------------------------------
# losses (utils/core.py:44-80,161-188) and Dice metric (medpy 0.5.2 binary.dc; metrics_eval.py:10-21)
# --------------------------------------------------------------------------------------
def dice_loss(logits: Tensor, target: Tensor, n_classes: int) -> Tensor:
    """core.py:67-80 with softmax=True, weight=None: mean_c 1 - (2 sum(p t)+eps)/(sum(p^2)+sum(t^2)+eps), sums over whole batch."""
    p = torch.softmax(logits, dim=1)
    loss = 0.0
    for i in range(n_classes):
        t = (target == i).float()
        s = p[:, i]
        inter = torch.sum(s * t)
        loss = loss + (1 - (2 * inter + 1e-5) / (torch.sum(s * s) + torch.sum(t * t) + 1e-5))
    return loss / n_classes


def boundary_dou_loss(logits: Tensor, target: Tensor, n_classes: int) -> Tensor:
    """core.py:83-131 (BoundaryDoULoss), device-agnostic.  Per class c, with t = (target == c) over the whole batch:
    C = number of foreground pixels whose 4-neighbourhood (zero padded) is not all foreground (core.py:105-109: cross-kernel
    conv * t, values 5 zeroed, count_nonzero), S = count_nonzero(t), alpha = min(2 (1 - (C+eps)/(S+eps)) - 1, 0.8),
    loss_c = (z + y - 2 I + eps) / (z + y - (1 + alpha) I + eps) with I = sum(p t), y = sum(t t), z = sum(p p);
    result = mean over classes (core.py:127-131).  alpha depends on the labels only and carries no gradient."""
    p = torch.softmax(logits, dim=1)
    kernel = torch.tensor([[0., 1., 0.], [1., 1., 1.], [0., 1., 0.]], dtype=logits.dtype).view(1, 1, 3, 3)
    smooth = 1e-5
    loss = 0.0
    for i in range(n_classes):
        t = (target == i).to(logits.dtype)
        y = F.conv2d(t.unsqueeze(1), kernel, padding=1)[:, 0] * t
        y = torch.where(y == 5, torch.zeros_like(y), y)
        cnt, s = torch.count_nonzero(y), torch.count_nonzero(t)
        alpha = 1 - (cnt + smooth) / (s + smooth)
        alpha = min(float(2 * alpha - 1), 0.8)
        score = p[:, i]
        inter, y_sum, z_sum = torch.sum(score * t), torch.sum(t * t), torch.sum(score * score)
        loss = loss + (z_sum + y_sum - 2 * inter + smooth) / (z_sum + y_sum - (1 + alpha) * inter + smooth)
    return loss / n_classes


def criterion(logits: Tensor, labels: Tensor, n_classes: int, loss_type=("dice", "ce"), weights=(0.5, 0.5)) -> Tensor:
    """core.py:179-188."""
    loss = 0.0
    for w, name in zip(weights, loss_type):
        if name == "ce":
            loss = loss + w * F.cross_entropy(logits, labels.long())
        elif name == "dice":
            loss = loss + w * dice_loss(logits, labels, n_classes)
        elif name == "boundary":
            loss = loss + w * boundary_dou_loss(logits, labels, n_classes)
        else:
            raise NotImplementedError(name)
    return loss


def dice_metric(pred: Tensor, gt: Tensor) -> float:
    """medpy.metric.binary.dc (published formula): 2|A∩B|/(|A|+|B|) on boolean masks, 0.0 if both empty.
    PARITY UNPINNED (medpy absent)."""
    a, b = pred.bool(), gt.bool()
    denom = int(a.sum()) + int(b.sum())
    return 2.0 * int((a & b).sum()) / denom if denom else 0.0


def jaccard_metric(pred, gt) -> float:
    """medpy.metric.binary.jc (published formula): |A∩B| / |A∪B| on boolean masks.  PARITY UNPINNED (medpy absent)."""
    a, b = np.asarray(pred).astype(bool), np.asarray(gt).astype(bool)
    return float(np.count_nonzero(a & b)) / float(np.count_nonzero(a | b))


def surface_distances(result, reference, voxelspacing=None, connectivity: int = 1):
    """medpy.metric.binary.__surface_distances (medpy 0.5.2, published algorithm): border = mask XOR its erosion by
    generate_binary_structure(ndim, connectivity); the Euclidean distance transform of the complement of the reference
    border, read at the result border.  Raises RuntimeError on an empty mask.  PARITY UNPINNED (medpy absent)."""
    from scipy.ndimage import binary_erosion, distance_transform_edt, generate_binary_structure
    result = np.atleast_1d(np.asarray(result).astype(bool))
    reference = np.atleast_1d(np.asarray(reference).astype(bool))
    footprint = generate_binary_structure(result.ndim, connectivity)
    if 0 == np.count_nonzero(result):
        raise RuntimeError("The first supplied array does not contain any binary object.")
    if 0 == np.count_nonzero(reference):
        raise RuntimeError("The second supplied array does not contain any binary object.")
    result_border = result ^ binary_erosion(result, structure=footprint, iterations=1)
    reference_border = reference ^ binary_erosion(reference, structure=footprint, iterations=1)
    dt = distance_transform_edt(~reference_border, sampling=voxelspacing)
    return dt[result_border]


def hd95_metric(result, reference) -> float:
    """medpy.metric.binary.hd95: 95th percentile (numpy, linear interpolation) of both directed distance sets together."""
    hd1 = surface_distances(result, reference)
    hd2 = surface_distances(reference, result)
    return float(np.percentile(np.hstack((hd1, hd2)), 95))


def assd_metric(result, reference) -> float:
    """medpy.metric.binary.assd: mean of the two directed average surface distances."""
    return float(np.mean((surface_distances(result, reference).mean(), surface_distances(reference, result).mean())))


def metric_percase(pred, gt):
    """calculate_metric_percase (src/utils/metrics_eval.py:9-21): (dice, hd95, jaccard, assd) with its empty-mask rules."""
    pred, gt = (np.asarray(pred) > 0), (np.asarray(gt) > 0)
    if pred.sum() > 0 and gt.sum() > 0:
        dice = 2.0 * np.count_nonzero(pred & gt) / float(np.count_nonzero(pred) + np.count_nonzero(gt))
        return dice, hd95_metric(pred, gt), jaccard_metric(pred, gt), assd_metric(pred, gt)
    if pred.sum() > 0 and gt.sum() == 0:
        return 1, 0, 1, 0
    return 0, 0, 0, 0


def predict(logits: Tensor) -> Tensor:
    """main_acdc.py:227."""
    return torch.argmax(torch.softmax(logits, dim=1), dim=1)


def mean_class_dice(logits: Tensor, labels: Tensor, n_classes: int) -> float:
    """Mean over foreground classes of dice_metric(pred==c, gt==c) — the per-class rule of metrics_eval.py:10-21
    (pred>0 & gt==0 -> ... handled by the caller; here both-empty -> counted as 1 like the wrapper's `else` branches)."""
    pred = predict(logits)
    vals = []
    for c in range(1, n_classes):
        p, g = pred == c, labels == c
        if p.sum() > 0 and g.sum() > 0:
            vals.append(dice_metric(p, g))
        elif p.sum() > 0 and g.sum() == 0:
            vals.append(1.0)
        else:
            vals.append(0.0)
    return sum(vals) / max(len(vals), 1)


# --------------------------------------------------------------------------------------
# deterministic weight filler shared by the golden generator, the tests and bench.py
# --------------------------------------------------------------------------------------
def fill_state_dict_(sd: SD, seed: int = 0, layer_scale: float = 0.5) -> SD:
    """Overwrite every tensor of a reference-keyed state dict with reproducible, non-degenerate values.

    CFAM is numerically invisible at init (layer_scale 1e-6), so layer scales are set O(1),
    BN running stats are made non-trivial, and all weights get fan-in scaled gaussians."""
    g = torch.Generator().manual_seed(seed)
    for k in sorted(sd.keys()):
        v = sd[k]
        if k.endswith("num_batches_tracked"):
            v.zero_()
        elif k.endswith("running_mean"):
            v.copy_(0.1 * torch.randn(v.shape, generator=g))
        elif k.endswith("running_var"):
            v.copy_(0.5 + torch.rand(v.shape, generator=g))
        elif "layer_scale" in k:
            v.copy_(layer_scale * (0.5 + torch.rand(v.shape, generator=g)))
        elif k.endswith(".denoising_module.w"):
            v.fill_(0.5)
        elif k.endswith("boundary.w") or k == "out.w":
            v.copy_(0.5 + 0.25 * torch.randn(v.shape, generator=g))
        elif "lambda_" in k:
            v.copy_(0.1 * torch.randn(v.shape, generator=g))
        elif v.ndim <= 1:  # norm weight / bias, linear/conv bias
            if k.endswith("weight"):
                v.copy_(1.0 + 0.1 * torch.randn(v.shape, generator=g))
            else:
                v.copy_(0.05 * torch.randn(v.shape, generator=g))
        else:
            fan_in = v[0].numel()
            v.copy_(torch.randn(v.shape, generator=g) * (1.0 / math.sqrt(fan_in)))
    return sd


# --------------------------------------------------------------------------------------
# state-dict schema — builds an empty reference-keyed state dict
# --------------------------------------------------------------------------------------
def state_dict_schema(cfg: CENetConfig) -> Dict[str, tuple]:
    """key -> shape for `networks.CENet(**kw).state_dict()`; pinned by tests/golden/schema_*.json."""
    s: Dict[str, tuple] = {}

    def lin(p, o, i, bias=True):
        s[p + ".weight"] = (o, i)
        if bias:
            s[p + ".bias"] = (o,)

    def conv(p, o, i, k, bias=True, kw=None):
        s[p + ".weight"] = (o, i, k, k if kw is None else kw)
        if bias:
            s[p + ".bias"] = (o,)

    def norm(p, c):
        s[p + ".weight"] = (c,)
        s[p + ".bias"] = (c,)

    def bn(p, c):
        norm(p, c)
        s[p + ".running_mean"] = (c,)
        s[p + ".running_var"] = (c,)
        s[p + ".num_batches_tracked"] = ()

    E = cfg.embed_dims
    for st in range(4):
        cin = 3 if st == 0 else E[st - 1]
        C = E[st]
        conv(f"backbone.patch_embed{st + 1}.proj", C, cin, 7 if st == 0 else 3)
        norm(f"backbone.patch_embed{st + 1}.norm", C)
        for i in range(cfg.depths[st]):
            p = f"backbone.block{st + 1}.{i}"
            norm(p + ".norm1", C)
            lin(p + ".attn.q", C, C)
            lin(p + ".attn.kv", 2 * C, C)
            lin(p + ".attn.proj", C, C)
            if cfg.sr_ratios[st] > 1:
                conv(p + ".attn.sr", C, C, cfg.sr_ratios[st])
                norm(p + ".attn.norm", C)
            norm(p + ".norm2", C)
            hid = C * cfg.mlp_ratios[st]
            lin(p + ".mlp.fc1", hid, C)
            conv(p + ".mlp.dwconv.dwconv", hid, 1, 3)
            lin(p + ".mlp.fc2", C, hid)
        norm(f"backbone.norm{st + 1}", C)

    chans = {4: E[3], 3: E[2], 2: E[1], 1: E[0]}
    for lvl in (4, 3, 2, 1):
        C = chans[lvl]
        p = f"decoder.dec{lvl}"
        s[p + ".layer_scale_1"] = (1, C, 1, 1)
        s[p + ".layer_scale_2"] = (1, C, 1, 1)
        bn(p + ".norm1", C)
        bn(p + ".norm2", C)
        conv(p + ".mca.gate", C, C, 1)
        conv(p + ".mca.proj_2", C, C, 1)
        g, q = int(5 / 16 * C), int(1 / 16 * C)
        for j in range(3):
            d = f"{p}.mca.value.dlps.{j}"
            conv(d + ".depthwise", g, 1, 3, bias=False)
            bn(d + ".depthwise_bn", g)
            conv(d + ".pointwise", g, g, 1, bias=False)
            bn(d + ".pointwise_bn", g)
        conv(f"{p}.mca.value.dlps.3.1", q, q, 1, bias=False)
        bn(f"{p}.mca.value.dlps.3.2", q)
        conv(p + ".mca.value.PW_conv", C, C, 1)
        n = p + ".mca.denoising_module"
        s[n + ".w"] = ()
        for nm in ("conv_theta", "conv_phi", "conv_g", "conv_out"):
            conv(f"{n}.{nm}", C, C, 1)
        bn(n + ".bn", C)
        s[p + ".mca.ccu.fc1.weight"] = (3 * C, 1, 3)
        s[p + ".mca.ccu.fc2.weight"] = (C, 3, 1)
        bn(p + ".mca.ccu.bn", C)
        conv(p + ".mlp.fc1", 4 * C, C, 1)
        conv(p + ".mlp.dwconv", 4 * C, 1, 3)
        conv(p + ".mlp.fc2", C, 4 * C, 1)
        s[p + ".mlp.srm.pwc.weight"] = (1, 3, 1, 1)
        s[p + ".mlp.srm.dwc.weight"] = (1, 3, 3, 3)
        bn(p + ".mlp.srm.bn", 1)
    for lvl, heads in zip((3, 2, 1), cfg.diffatt_num_heads):
        cin, cout = chans[lvl + 1], chans[lvl]
        p = f"decoder.up{lvl}"
        conv(p + ".up_dwc.1", cin, 1, 3, bias=False)
        bn(p + ".up_dwc.2", cin)
        conv(p + ".pwc.0", cout, cin, 1)
        p = f"decoder.skip_enhancer{lvl}"
        E2 = 2 * cout
        s[p + ".boundary.w"] = (1, E2, 1, 1)
        hd = E2 // heads // 2
        for nm in ("lambda_q1", "lambda_k1", "lambda_q2", "lambda_k2"):
            s[f"{p}.diffattn.{nm}"] = (hd,)
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            lin(f"{p}.diffattn.{nm}", E2, E2, bias=False)
        conv(p + ".mixer", cout, E2, 1, bias=False)

    om = E[0] // 2
    s["out.w"] = (1, om, 1, 1)
    conv("out.rb.0.conv1.conv", om, cfg.input_channels, 5, bias=False)
    conv("out.rb.0.conv2.conv", om, om, 5, bias=False)
    conv("out.rb.0.conv3.conv", om, cfg.input_channels, 1, bias=False)
    for nm in ("norm1", "norm2", "norm3"):
        bn("out.rb.0." + nm, om)
    conv("out.up.up.1", om, E[0], 3, bias=False)
    bn("out.up.up.2", om)
    conv("out.out.0.conv1.conv", 2 * om, 2 * om, 3, bias=False)
    conv("out.out.0.conv2.conv", 2 * om, 2 * om, 3, bias=False)
    bn("out.out.0.norm1", 2 * om)
    bn("out.out.0.norm2", 2 * om)
    conv("out.out.1.conv.conv", cfg.num_classes, 2 * om, 1)
    return s


def make_state_dict(cfg: CENetConfig, seed: int = 42, layer_scale: float = 0.5) -> SD:
    """Empty reference-keyed state dict filled by `fill_state_dict_` (same values the golden generator loaded
    into the reference)."""
    sd = {}
    for k, shp in state_dict_schema(cfg).items():
        sd[k] = torch.zeros(shp, dtype=torch.long if k.endswith("num_batches_tracked") else torch.float32)
    return fill_state_dict_(sd, seed=seed, layer_scale=layer_scale)


def synthetic_batch(B: int, cin: int, K: int, seed: int = 1234, size: int = 224):
    """Seeded images + blocky integer labels as float [B,H,W]; identical to gen_golden's."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, cin, size, size, generator=g)
    low = torch.rand(B, 1, size // 16, size // 16, generator=g)
    lab = torch.floor(F.interpolate(low, size=(size, size), mode="nearest") * K).clamp_(0, K - 1)
    return x, lab[:, 0]
