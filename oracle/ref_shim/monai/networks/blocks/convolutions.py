"""monai Convolution stand-in, conv_only=True / 2-D only: a Sequential with one child named `conv`."""
from torch import nn


class Convolution(nn.Sequential):
    def __init__(self, spatial_dims, in_channels, out_channels, strides=1, kernel_size=3, act=None, norm=None,
                 dropout=None, bias=True, conv_only=False, is_transposed=False, padding=None,
                 output_padding=None, **_):
        super().__init__()
        if spatial_dims != 2 or not conv_only:
            raise NotImplementedError("shim covers the reference's only usage: 2-D, conv_only=True")
        if padding is None:
            padding = (kernel_size - strides + 1) // 2
        if is_transposed:
            conv = nn.ConvTranspose2d(in_channels, out_channels, kernel_size, stride=strides, padding=padding,
                                      output_padding=output_padding or 0, bias=bias)
        else:
            conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=strides, padding=padding, bias=bias)
        self.add_module("conv", conv)
