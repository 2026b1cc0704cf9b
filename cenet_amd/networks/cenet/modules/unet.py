"""UnetResBlock / UnetOutBlock on HIP kernels — mirrors reference src/networks/cenet/modules/unet.py:123-214,357-381.
The `.conv` child level reproduces monai's Convolution(nn.Sequential) wrapper so state-dict keys match."""
from __future__ import annotations

import torch.nn as nn

from .... import ops
from .blocks import bn_call


def _conv_layer(cin, cout, k, bias=False):
    seq = nn.Sequential()
    seq.add_module("conv", nn.Conv2d(cin, cout, k, stride=1, padding=k // 2, bias=bias))
    return seq


def _init(m):
    if isinstance(m, (nn.Conv2d, nn.Linear)):
        nn.init.trunc_normal_(m.weight, std=0.02)
        if m.bias is not None:
            nn.init.zeros_(m.bias)


class UnetResBlock(nn.Module):
    def __init__(self, spatial_dims, in_channels, out_channels, kernel_size, stride, norm_name,
                 act_name=("leakyrelu", {"inplace": True, "negative_slope": 0.01}), dropout=None):
        super().__init__()
        if spatial_dims != 2 or stride != 1 or norm_name != "batch":
            raise NotImplementedError("CENet uses UnetResBlock(2-D, stride 1, batch norm) only")
        self.k = kernel_size
        self.slope = act_name[1].get("negative_slope", 0.01)
        self.conv1 = _conv_layer(in_channels, out_channels, kernel_size)
        self.conv2 = _conv_layer(out_channels, out_channels, kernel_size)
        self.lrelu = nn.LeakyReLU(self.slope)
        self.norm1 = nn.BatchNorm2d(out_channels)
        self.norm2 = nn.BatchNorm2d(out_channels)
        self.downsample = in_channels != out_channels
        if self.downsample:
            self.conv3 = _conv_layer(in_channels, out_channels, 1)
            self.norm3 = nn.BatchNorm2d(out_channels)
        self.apply(_init)

    def forward(self, inp, pool_scale=None):
        """pool_scale (OutHead.w, out.py:43,70): return pool_scale * MaxPool2d(2,2)(block(inp)) instead of block(inp) — on bf16 maps
        in training mode the two BatchNorms, the residual add, the LeakyReLU and the pool are ONE kernel (ops.res_tail_pool)"""
        p = self.k // 2
        out = ops.conv2d_nchw(inp, self.conv1.conv.weight, None, stride=1, pad=p)
        out = bn_call(self.norm1, out, "lrelu", self.slope)
        out = ops.conv2d_nchw(out, self.conv2.conv.weight, None, stride=1, pad=p)
        if pool_scale is not None and self.downsample:
            if ops.res_tail_img_pool_supported(out, inp, self.conv3.conv.weight, self.norm2, self.norm3, pool_scale):
                # one-channel input: the shortcut w3[c] * inp is never materialised (ops.ResTailImgPoolFn)
                return ops.res_tail_img_pool(out, self.norm2, inp, self.conv3.conv.weight, self.norm3, pool_scale, self.slope)
            res = ops.conv1x1(inp, self.conv3.conv.weight)
            if ops.res_tail_pool_supported(out, res, self.norm2, self.norm3, pool_scale):
                return ops.res_tail_pool(out, self.norm2, res, self.norm3, pool_scale, self.slope)
            y = ops.add_act(bn_call(self.norm2, out), bn_call(self.norm3, res), "lrelu", self.slope)
            return ops.maxpool2_scale(y, pool_scale)
        out = bn_call(self.norm2, out)
        res = inp
        if self.downsample:
            res = bn_call(self.norm3, ops.conv1x1(inp, self.conv3.conv.weight))
        y = ops.add_act(out, res, "lrelu", self.slope)
        return ops.maxpool2_scale(y, pool_scale) if pool_scale is not None else y


class UnetOutBlock(nn.Module):
    def __init__(self, spatial_dims, in_channels, out_channels, dropout=None, **kwargs):
        super().__init__()
        self.conv = _conv_layer(in_channels, out_channels, 1, bias=True)
        self.apply(_init)

    def forward(self, inp):
        return ops.conv1x1(inp, self.conv.conv.weight, self.conv.conv.bias)
