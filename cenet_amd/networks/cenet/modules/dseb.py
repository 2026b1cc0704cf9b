"""Dual Selective Enhancement Block on HIP kernels — mirrors reference src/networks/cenet/modules/dseb.py:26-165
(use_command='dat-fea'; the 'dog' / 'seq' variants are not used by CENet)."""
from __future__ import annotations

import torch
import torch.nn as nn

from .... import ops
from .multihead_diffattn import MultiheadDiffAttn


class FEA(nn.Module):
    """dseb.py:26-76. Holds the per-channel edge weight; `recons` returns up(down_s(x)) per scale (None for s == 1)."""

    def __init__(self, dim: int, scale_factors: list, label="", writer=None) -> None:
        super().__init__()
        self.scale_factors = list(scale_factors)
        self.n = len(scale_factors)
        assert 2 <= self.n <= 3, "FEA supports 2 or 3 scales"
        self.w = nn.Parameter(torch.randn(1, dim, 1, 1) + 0.5)

    def recons(self, x, tap=False):
        """tap=True: returns (list, x_tap) — x routed through the down-samplings' autograd nodes, for x's remaining consumer"""
        H, W = x.shape[2:]
        out = []
        for s in self.scale_factors:
            if float(s) == 1.0:
                out.append(None)  # interpolate(scale 1.0) is the identity -> e_s == 0 exactly
            else:
                if tap:
                    d, x = ops.interpolate_bilinear(x, scale_factor=s, align_corners=False, tap=True)
                else:
                    d = ops.interpolate_bilinear(x, scale_factor=s, align_corners=False)
                out.append(ops.interpolate_bilinear(d, size=(H, W), align_corners=False))
        return (out, x) if tap else out

    def forward(self, x):
        return ops.dseb_combine(x, self.w, None, self.recons(x), ycoef=1.0)  # x + w*edge


class DSEBlock(nn.Module):
    def __init__(self, dim, scale_factors, num_heads, input_size, mode='add', use_command='dat-fea', depth=1, label="",
                 writer=None):
        super().__init__()
        if mode.lower() != "cat" or use_command != "dat-fea":
            raise NotImplementedError("CENet uses DSEBlock(mode='cat', use_command='dat-fea') only")
        self.input_size = input_size
        _dim = dim * 2
        self.boundary = FEA(dim=_dim, scale_factors=scale_factors, label=label, writer=writer)
        self.diffattn = MultiheadDiffAttn(embed_dim=_dim, depth=depth, num_heads=num_heads)
        self.mixer = nn.Conv2d(_dim, dim, kernel_size=1, stride=1, bias=False)
        self.dec_tap = None  # (set by forward, taken by Decoder.forward)

    def forward(self, skip, dec):
        # dec has a second consumer in the decoder (decoders.py:96, the residual add).  It reads `self.dec_tap` (dec itself, routed
        # through the concat's autograd node) so that both of dec's gradients — and both of skip's — meet in the concat's backward
        # kernel (ops.Concat2Fn) instead of two aten::add launches per level
        y, dec, skip = ops.concat2(dec, skip, tap=True)
        self.dec_tap = dec
        B, C2, H, W = y.shape
        # dseb.py:115: the flat NCHW buffer re-read as [B, HW, 2C] tokens (a view, not a permute)
        # y has four consumers (the attention's projections, two down-samplings, the combine): each hands y on as a tap, so that the
        # four gradients are added by the backward kernels along that chain instead of three aten::add launches on 2C-channel maps
        diff, yt = self.diffattn(y.view(B, H * W, C2), None, None, True)
        rec, yt = self.boundary.recons(yt.view(B, C2, H, W), tap=True)
        z = ops.dseb_combine(yt, self.boundary.w, diff.view(B, C2, H, W), rec)  # (FEA(y)+y) + diff*y
        return ops.conv1x1(z, self.mixer.weight, None, resid=skip)
