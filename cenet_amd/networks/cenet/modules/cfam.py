"""Context Feature Attention Module on HIP kernels — mirrors reference src/networks/cenet/modules/cfam.py."""
from __future__ import annotations

import torch
import torch.nn as nn

from .... import ops
from .blocks import SepConvBN, bn_call
from .nlb import Nonlocal


class SRM(nn.Module):
    """cfam.py:86-101."""

    def __init__(self):
        super().__init__()
        self.pwc = nn.Conv2d(3, 1, kernel_size=1, bias=False)
        self.dwc = nn.Conv2d(3, 1, kernel_size=3, padding=1, bias=False)
        self.bn = nn.BatchNorm2d(1)

    def forward(self, x):
        b = self.bn
        return ops.srm(x, self.pwc.weight, self.dwc.weight, b.weight, b.bias, b.running_mean, b.running_var,
                       b.num_batches_tracked, b.training)


class Mlp(nn.Module):
    """cfam.py:104-159: 1x1 -> DW3x3(+bias) -> GELU -> SRM -> 1x1."""

    def __init__(self, embed_dims, feedforward_channels, kernel_size=3, act_type='GELU', ffn_drop=0.):
        super().__init__()
        if kernel_size != 3 or act_type != "GELU" or ffn_drop != 0.:
            raise NotImplementedError
        self.fc1 = nn.Conv2d(embed_dims, feedforward_channels, 1)
        self.dwconv = nn.Conv2d(feedforward_channels, feedforward_channels, 3, 1, 1, bias=True, groups=feedforward_channels)
        self.fc2 = nn.Conv2d(feedforward_channels, embed_dims, 1)
        self.srm = SRM()

    def forward(self, x):
        x = ops.conv1x1(x, self.fc1.weight, self.fc1.bias)
        x = ops.dwconv_nchw(x, self.dwconv.weight, self.dwconv.bias, dil=1, act="gelu")
        x = self.srm(x)
        return ops.conv1x1(x, self.fc2.weight, self.fc2.bias)


class MultiOrderDWConv(nn.Module):
    """cfam.py:162-241: 5/16,5/16,5/16,1/16 channel split; three dilated SepConvBN branches + a pooled branch."""

    def __init__(self, embed_dims, channel_split=[1, 3, 4, 2], rates=[6, 12, 18], flag_useAllChannels=False):
        super().__init__()
        if flag_useAllChannels:
            raise NotImplementedError
        g, q = int(5 / 16 * embed_dims), int(1 / 16 * embed_dims)
        assert q > 0, "Ops. Channel split ratio is not correct"
        self.channel_indices = [(0, g), (g, 2 * g), (2 * g, 3 * g), (3 * g, 3 * g + q)]
        self.rates = list(rates)
        self.dlps = nn.ModuleList([SepConvBN(g, g, kernel_size=3, stride=1, rate=r, depth_activation=True, epsilon=1e-5)
                                   for r in rates])
        self.dlps.append(nn.Sequential(
            nn.AdaptiveAvgPool2d((7, 7)),
            nn.Conv2d(q, q, kernel_size=1, bias=False),
            nn.BatchNorm2d(q, eps=1e-5),
            nn.LeakyReLU(),
            nn.UpsamplingBilinear2d(scale_factor=7)))
        self.embed_dims = embed_dims
        self.PW_conv = nn.Conv2d(embed_dims, embed_dims, kernel_size=1)

    # ---- the three dilated branches as ONE chain of launches (K11 of SURVEY §3: "CFAM multi-scale conv fusion") ------------
    # Their BatchNorms and pointwise convs have identical shapes, so with the parameters of equal role back to back in the
    # ParamArena (arena_groups) one BatchNorm launch covers 3g channels and one batched GEMM the three 1x1 convs; the depthwise
    # kernels write their slices of one [B, 3g, H, W] tensor, which is also where the concat wants them.
    _ROLES = (("dbn_w", lambda m: m.depthwise_bn.weight), ("dbn_b", lambda m: m.depthwise_bn.bias),
              ("pw", lambda m: m.pointwise.weight), ("pbn_w", lambda m: m.pointwise_bn.weight),
              ("pbn_b", lambda m: m.pointwise_bn.bias))

    def arena_groups(self):
        return [[get(m) for m in list(self.dlps)[:3]] for _, get in self._ROLES]

    def _merged(self):
        """joint views of the three branches' parameters / buffers, or None when the parameters are not laid out for it (no
        ParamArena: eval scripts, the parity tests on bare modules) — the branch-by-branch path runs then"""
        b = list(self.dlps)[:3]
        mg = getattr(self, "_mg", None)
        if mg is None or any(mg[k].data_ptr() != get(b[0]).data_ptr() for k, get in self._ROLES):
            mg = {k: ops.merged_param([get(m) for m in b]) for k, get in self._ROLES}
            if any(v is None for v in mg.values()):
                self._mg = None
                return None
            self._mg = mg
        for key, bn in (("dbn", "depthwise_bn"), ("pbn", "pointwise_bn")):
            for stat in ("running_mean", "running_var", "num_batches_tracked"):
                mg[key + "_" + stat] = ops.merged_buffer([getattr(getattr(m, bn), stat) for m in b])
        return mg

    def _branches_merged(self, x, sizes, mg):
        b = list(self.dlps)[:3]
        dbn = b[0].depthwise_bn
        fused = ops.split_dwconv_bn_supported(x, sizes[:3], dbn.training)
        if fused:
            # small maps: the three depthwise convs, their BatchNorm + ReLU and the pooled slice's copy in one launch per pass
            u, rest = ops.split_dwconv_bn(x, sizes[:3], [m.rate for m in b], [m.depthwise.weight for m in b], mg["dbn_w"],
                                          mg["dbn_b"], mg["dbn_running_mean"], mg["dbn_running_var"],
                                          mg["dbn_num_batches_tracked"], dbn.eps, ops.bn_momentum(dbn))
        else:
            u, rest = ops.split_dwconv(x, sizes[:3], [m.rate for m in b], [m.depthwise.weight for m in b], joined=True)
        for key, bn in ((("pbn", b[0].pointwise_bn),) if fused else (("dbn", b[0].depthwise_bn), ("pbn", b[0].pointwise_bn))):
            if key == "pbn":
                u = ops.grouped_conv1x1(u, mg["pw"])
                pool = self.dlps[3]
                if bn.training and rest is not None and ops.pool_branch_supported(rest, pool[2]):
                    # the pointwise BatchNorm + ReLU and the pooled branch write their channel slices of ONE tensor: no concat
                    return ops.join_bn_pool(u, rest, mg["pbn_w"], mg["pbn_b"], mg["pbn_running_mean"], mg["pbn_running_var"],
                                            mg["pbn_num_batches_tracked"], bn.eps, ops.bn_momentum(bn),
                                            pool[1].weight, pool[2]), None
            u = ops.batchnorm(u, mg[key + "_w"], mg[key + "_b"], mg[key + "_running_mean"], mg[key + "_running_var"],
                              mg[key + "_num_batches_tracked"], bn.training, bn.eps, "relu", 0.0,
                              ops.bn_momentum(bn))
        return u, rest

    def forward(self, x):
        H, W = x.shape[2:]
        # the three dilated branches read their channel slice of x in place (ops.split_dwconv); only the small pooled
        # branch gets a copy
        sizes = [hi - lo for lo, hi in self.channel_indices]
        b = list(self.dlps)[:3]
        mg = self._merged()
        if mg is not None:
            ops.refresh_member_shadows(mg["pw"], x)
            u, rest = self._branches_merged(x, sizes, mg)
            if rest is None:  # (the branches and the pooled branch already share one tensor)
                return ops.conv1x1(u, self.PW_conv.weight, self.PW_conv.bias)
            ys = [u]
        else:
            us = ops.split_dwconv(x, sizes[:3], [m.rate for m in b], [m.depthwise.weight for m in b])
            ys = [b[j].after_depthwise(us[j]) for j in range(3)]
            rest = us[3]
        pool = self.dlps[3]
        if ops.pool_branch_supported(rest, pool[2]):
            y = ops.pool_branch(rest, pool[1].weight, pool[2])  # the whole pooled branch: two launches per pass
        else:
            y = ops.adaptive_avgpool(rest, 7, 7)
            y = ops.conv1x1(y, pool[1].weight)
            y = bn_call(pool[2], y, "lrelu", 0.01)
            y = ops.interpolate_bilinear(y, scale_factor=7, align_corners=True)
            if y.shape[2] != H or y.shape[3] != W:
                y = ops.interpolate_bilinear(y, size=(H, W), align_corners=False)
        ys.append(y)
        x = ops.concat(ys)
        return ops.conv1x1(x, self.PW_conv.weight, self.PW_conv.bias)


class CCU(nn.Module):
    """cfam.py:244-264."""

    def __init__(self, channel, hidden_scale=3):
        super().__init__()
        if hidden_scale != 3:
            raise NotImplementedError
        self.fc1 = nn.Conv1d(channel, 3 * channel, kernel_size=3, groups=channel, bias=False)
        self.fc2 = nn.Conv1d(3 * channel, channel, kernel_size=1, groups=channel, bias=False)
        self.bn = nn.BatchNorm1d(channel)

    def forward(self, x, tap=False):
        b = self.bn
        return ops.ccu(x, self.fc1.weight, self.fc2.weight, b.weight, b.bias, b.running_mean, b.running_var,
                       b.num_batches_tracked, b.training, tap)


class MCA(nn.Module):
    """cfam.py:267-306."""

    def __init__(self, embed_dims, attn_channel_split=[1, 3, 4], attn_act_type='SiLU', rates=[2, 3, 4]):
        super().__init__()
        if attn_act_type != "SiLU":
            raise NotImplementedError
        self.embed_dims = embed_dims
        self.gate = nn.Conv2d(embed_dims, embed_dims, 1)
        self.value = MultiOrderDWConv(embed_dims=embed_dims, rates=rates, channel_split=attn_channel_split)
        self.proj_2 = nn.Conv2d(embed_dims, embed_dims, 1)
        self.denoising_module = Nonlocal(embed_dims)
        self.ccu = CCU(embed_dims)

    def forward(self, x, raw=False, gated=None):
        """raw=True: (output conv of the Non-local block before its BatchNorm, the block's input) — see Nonlocal.forward
        gated: the CCU output when the caller computed it together with x (CFAModule, ops.cfam_front); x is then the shortcut"""
        if gated is not None:
            x, shortcut = gated, x
        else:
            x, shortcut = self.ccu(x, True)  # (shortcut = x itself, through the CCU's autograd node: its gradient joins there)
        g, x = ops.conv1x1(x, self.gate.weight, self.gate.bias, tap=True)  # value's gradient joins inside gate's dgrad GEMM
        v = self.value(x)
        x = ops.conv1x1(ops.silu_mul(g, v), self.proj_2.weight, self.proj_2.bias, resid=shortcut)
        return self.denoising_module(x, raw)


class CFAModule(nn.Module):
    """cfam.py:309-374: x + ls1*MCA(BN(x)); x + ls2*Mlp(BN(x))."""

    def __init__(self, embed_dims, ffn_ratio=4., drop_rate=0., drop_path_rate=0., act_type='GELU', norm_type='BN',
                 init_value=1e-5, attn_channel_split=[1, 3, 4], attn_act_type='SiLU', mca_rates=[6, 12, 18], writer=None):
        super().__init__()
        if norm_type != "BN" or drop_path_rate != 0.:
            raise NotImplementedError("CENet uses CFAModule(norm_type='BN', drop_path_rate=0) only")
        self.out_channels = embed_dims
        self.norm1 = nn.BatchNorm2d(embed_dims, eps=1e-5)
        self.mca = MCA(embed_dims, attn_channel_split=attn_channel_split, attn_act_type=attn_act_type, rates=mca_rates)
        self.norm2 = nn.BatchNorm2d(embed_dims, eps=1e-5)
        self.mlp = Mlp(embed_dims, int(embed_dims * ffn_ratio), 3, act_type, drop_rate)
        self.layer_scale_1 = nn.Parameter(init_value * torch.ones((1, embed_dims, 1, 1)), requires_grad=True)
        self.layer_scale_2 = nn.Parameter(init_value * torch.ones((1, embed_dims, 1, 1)), requires_grad=True)

    def forward(self, x):
        # (taps: the residual connection reads x through the BatchNorm's autograd node, whose backward kernel then writes the sum
        # of both gradients of x — no aten::add behind it)
        gated = None
        if ops.cfam_front_supported(x, self.norm1, self.mca.ccu):
            y, gated, x = ops.cfam_front(x, self.norm1, self.mca.ccu)  # norm1 + the CCU gate in one launch per pass
        else:
            y, x = bn_call(self.norm1, x, tap=True)
        nl = self.mca.denoising_module
        if ops.cfam_mid_supported(x, nl.bn, self.norm2):
            # small maps: BatchNorm of the Non-local output conv + residual mix + layer-scale residual + norm2 in one launch per
            # pass (csrc/chanloc.hip: workgroup = channel over the batch)
            p_raw, m = self.mca(y, raw=True, gated=gated)
            x, y = ops.cfam_mid(p_raw, m, x, nl.w, self.layer_scale_1, nl.bn, self.norm2)
        else:
            x = ops.scale_residual(x, self.mca(y, gated=gated), self.layer_scale_1)
            y, x = bn_call(self.norm2, x, tap=True)
        x = ops.scale_residual(x, self.mlp(y), self.layer_scale_2)
        return x
