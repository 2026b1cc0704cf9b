"""Encoder factory — mirrors reference src/networks/cenet/encoder.py:6-88 (PVTv2 branches; ResNets out of scope)."""
from __future__ import annotations

import torch

from . import pvtv2

_PVT_CHANNELS = {"pvt_v2_b0": [256, 160, 64, 32], "pvt_v2_b1": [512, 320, 128, 64], "pvt_v2_b2": [512, 320, 128, 64],
                 "pvt_v2_b3": [512, 320, 128, 64], "pvt_v2_b4": [512, 320, 128, 64], "pvt_v2_b5": [512, 320, 128, 64]}


def get_encoder2d(input_channels=1, encoder='pvt_v2_b2', pretrain=False, freeze_bb=False, base_ptdir='.'):
    if 'resnet' in encoder:
        raise NotImplementedError("ResNet encoders are out of scope (SURVEY.md §2 #18); use a pvt_v2_* encoder")
    if encoder not in _PVT_CHANNELS:
        print('Encoder not implemented! Continuing with default encoder pvt_v2_b2.')
        encoder = 'pvt_v2_b2'
    backbone = getattr(pvtv2, encoder)()
    channels = _PVT_CHANNELS[encoder]
    path = f'{base_ptdir}/pvt/{encoder}.pth'
    if pretrain and base_ptdir:
        print(f'Loading pretrained weights from {path}')
        saved = torch.load(path, map_location="cpu")
        own = backbone.state_dict()
        own.update({k: v for k, v in saved.items() if k in own})
        backbone.load_state_dict(own)
        if freeze_bb:
            for p in backbone.parameters():
                p.requires_grad = False
    else:
        print('No pretrained weights loaded! ...')
    return backbone, channels
