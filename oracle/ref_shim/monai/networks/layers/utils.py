from torch import nn


def get_act_layer(name):
    kind, kw = (name, {}) if isinstance(name, str) else name
    if kind.lower() != "leakyrelu":
        raise NotImplementedError(kind)
    return nn.LeakyReLU(**kw)


def get_norm_layer(name, spatial_dims=2, channels=1):
    kind = name if isinstance(name, str) else name[0]
    if kind.lower() != "batch" or spatial_dims != 2:
        raise NotImplementedError(kind)
    return nn.BatchNorm2d(channels)
