"""Output head — mirrors reference src/networks/cenet/out.py:10-75 (merge 'cat', up block 'upcn')."""
from __future__ import annotations

import torch
import torch.nn as nn

from ... import ops
from .modules.blocks import UpConv
from .modules.unet import UnetOutBlock, UnetResBlock


class OutHead(nn.Module):
    def __init__(self, dec_in_channels, x_in_channels, out_channels, dec_in_spatial=56, x_in_spatial=224,
                 merge_mode='cat', up_block="upcn", up_ks=3):
        super().__init__()
        assert up_block in ["uprb", "eucb", "upcn", "uptc"], f"Invalid up_block: {up_block}"
        assert merge_mode in ["cat", "add"], f"Invalid merge_mode: {merge_mode}"
        if up_block != "upcn" or merge_mode != "cat":
            raise NotImplementedError("only out_up_block='upcn', out_merge_mode='cat' are in scope (SURVEY.md §8b)")
        self.merge_mode = merge_mode
        om = dec_in_channels // 2
        act = ("leakyrelu", {"inplace": True, "negative_slope": 0.01})
        self.w = nn.Parameter(torch.randn((1, om, 1, 1)) + 0.75)
        self.out = nn.Sequential(
            UnetResBlock(2, 2 * om, 2 * om, kernel_size=3, stride=1, norm_name='batch', act_name=act, dropout=0),
            UnetOutBlock(spatial_dims=2, in_channels=2 * om, out_channels=out_channels))
        self.up = UpConv(in_channels=dec_in_channels, out_channels=om, kernel_size=up_ks, stride=1, activation='leakyrelu')
        self.rb = nn.Sequential(
            UnetResBlock(2, x_in_channels, om, kernel_size=5, stride=1, norm_name='batch', act_name=act, dropout=0),
            nn.MaxPool2d(kernel_size=2, stride=2))

    def branch(self, x):
        """w * MaxPool2(ResBlock5x5(x)) (out.py:69): block tail and pool fused (UnetResBlock.forward).  Depends on the input image
        only — CENet._forward launches it on a branch stream ahead of the encoder and hands the result to forward(rb=...)"""
        return self.rb[0](x, self.w)

    def forward(self, dec, x, rb=None):
        if rb is None:
            rb = self.branch(x)
        elif rb.is_cuda:
            cur = torch.cuda.current_stream(rb.device)
            bs = ops.branch_stream(rb)
            if bs is not None:
                cur.wait_stream(bs)
                rb.record_stream(cur)
        d = self.up(dec)
        z = ops.concat2(d, rb)
        y = self.out[1](self.out[0](z))
        return ops.interpolate_bilinear(y, scale_factor=2, align_corners=False)
