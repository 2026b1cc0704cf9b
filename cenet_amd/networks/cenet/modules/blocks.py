"""SepConvBN / EUCB / UpConv on HIP kernels — mirrors reference src/networks/cenet/modules/blocks.py:131-185,206-221,297-321."""
from __future__ import annotations

import math

import torch.nn as nn

from .... import ops


def bn_call(bn: nn.BatchNorm2d, x, act="none", slope=0.0, tap=False):
    """Train/eval BatchNorm (+fused activation) through the HIP kernels using the container's tensors.
    tap=True: returns (y, x_tap) — x itself routed through the BatchNorm's autograd node, for the residual connection around it"""
    return ops.batchnorm(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, bn.training,
                         bn.eps, act, slope, ops.bn_momentum(bn), tap)


def _init_conv(m, scheme="normal"):
    """blocks.py:93-127 ('normal' scheme for convs; BN weight 1 / bias 0)."""
    if isinstance(m, nn.Conv2d):
        if scheme == "normal":
            nn.init.normal_(m.weight, std=.02)
        else:
            fan_out = m.kernel_size[0] * m.kernel_size[1] * m.out_channels // m.groups
            nn.init.normal_(m.weight, 0, math.sqrt(2.0 / fan_out))
        if m.bias is not None:
            nn.init.zeros_(m.bias)
    elif isinstance(m, nn.BatchNorm2d):
        nn.init.ones_(m.weight)
        nn.init.zeros_(m.bias)


class SepConvBN(nn.Module):
    """blocks.py:131-185 with depth_activation=True: DW(dilated) -> BN -> ReLU -> PW -> BN -> ReLU."""

    def __init__(self, in_channels, filters, kernel_size=3, stride=1, rate=1, depth_activation=False, epsilon=1e-3):
        super().__init__()
        if kernel_size != 3 or stride != 1 or not depth_activation:
            raise NotImplementedError("CENet uses SepConvBN(k=3, stride=1, depth_activation=True) only")
        self.rate = rate
        self.depthwise = nn.Conv2d(in_channels, in_channels, 3, stride=1, padding=rate, dilation=rate, groups=in_channels,
                                   bias=False)
        self.depthwise_bn = nn.BatchNorm2d(in_channels, eps=epsilon)
        self.pointwise = nn.Conv2d(in_channels, filters, 1, bias=False)
        self.pointwise_bn = nn.BatchNorm2d(filters, eps=epsilon)
        self.apply(_init_conv)

    def forward(self, x):
        return self.after_depthwise(ops.dwconv_nchw(x, self.depthwise.weight, None, dil=self.rate))

    def after_depthwise(self, x):
        """the block from its depthwise output on (MultiOrderDWConv runs the depthwise convs of its branches itself,
        reading their channel slices in place)"""
        x = bn_call(self.depthwise_bn, x, "relu")
        x = ops.conv1x1(x, self.pointwise.weight)
        return bn_call(self.pointwise_bn, x, "relu")


class EUCB(nn.Module):
    """blocks.py:297-321: nearest x2 -> DW3x3 -> BN -> LeakyReLU(0.2) -> (channel_shuffle(groups=C) == identity) -> 1x1+bias."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, activation='relu'):
        super().__init__()
        if kernel_size != 3 or stride != 1 or activation != "leakyrelu":
            raise NotImplementedError("CENet uses EUCB(k=3, stride=1, leakyrelu) only")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.up_dwc = nn.Sequential(
            nn.Upsample(scale_factor=2),
            nn.Conv2d(in_channels, in_channels, 3, stride=1, padding=1, groups=in_channels, bias=False),
            nn.BatchNorm2d(in_channels),
            nn.LeakyReLU(0.2))
        self.pwc = nn.Sequential(nn.Conv2d(in_channels, out_channels, 1, bias=True))
        self.apply(_init_conv)

    def forward(self, x):
        bn = self.up_dwc[2]
        if ops.eucb_front_supported(x, bn.training):
            # one launch per pass for everything in front of the 1x1 conv (csrc/chanloc.hip: workgroup = channel over the batch)
            x = ops.eucb_front(x, self.up_dwc[1].weight, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                               bn.num_batches_tracked, bn.eps, 0.2, ops.bn_momentum(bn))
        else:
            x = ops.nearest2x(x)
            x = ops.dwconv_nchw(x, self.up_dwc[1].weight, None, dil=1)
            x = bn_call(bn, x, "lrelu", 0.2)
        return ops.conv1x1(x, self.pwc[0].weight, self.pwc[0].bias)


class UpConv(nn.Module):
    """blocks.py:206-221: bilinear(align_corners=True) x2 -> 3x3 conv -> BN -> LeakyReLU(0.2)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, activation='relu'):
        super().__init__()
        if stride != 1 or activation != "leakyrelu":
            raise NotImplementedError("CENet uses UpConv(stride=1, leakyrelu) only")
        self.kernel_size = kernel_size
        self.up = nn.Sequential(
            nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True),
            nn.Conv2d(in_channels, out_channels, kernel_size, stride=1, padding=kernel_size // 2, bias=False),
            nn.BatchNorm2d(out_channels),
            nn.LeakyReLU(0.2))
        self.apply(_init_conv)

    def forward(self, x):
        x = ops.interpolate_bilinear(x, scale_factor=2, align_corners=True)
        x = ops.conv2d_nchw(x, self.up[1].weight, None, stride=1, pad=self.kernel_size // 2)
        return bn_call(self.up[2], x, "lrelu", 0.2)
