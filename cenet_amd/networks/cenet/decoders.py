"""CENet decoder — mirrors reference src/networks/cenet/decoders.py:35-105."""
from __future__ import annotations

from functools import partial

import torch.nn as nn

from ... import ops
from .modules.blocks import EUCB
from .modules.cfam import CFAModule
from .modules.dseb import DSEBlock


class Decoder(nn.Module):
    def __init__(self, channels=[512, 320, 128, 64], input_size=[14, 28, 56, 112], scale_factors=[0.8, 0.4],
                 skip_mode='add', num_heads=[2, 2, 2], up_block='eucb', writer=None):
        super().__init__()
        assert up_block in ["uprb", "eucb", "upcn", "uptc"], f"Invalid up_block: {up_block}"
        if up_block != "eucb":
            raise NotImplementedError("only dec_up_block='eucb' is in scope (SURVEY.md §8b)")
        self.input_size = input_size
        up = partial(EUCB, kernel_size=3, stride=1, activation='leakyrelu')
        rates = [[2, 3, 5], [1, 2, 4], [1, 2, 3], [1, 2, 2]]  # 56x56, 28x28, 14x14, 7x7 (decoders.py:64)
        dec = partial(CFAModule, ffn_ratio=4, drop_rate=0, drop_path_rate=0, act_type='GELU', norm_type="BN",
                      init_value=1e-6, attn_channel_split=[1, 3, 4], attn_act_type="SiLU")
        skip = partial(DSEBlock, scale_factors=scale_factors, mode=skip_mode, writer=writer)
        self.dec4 = dec(embed_dims=channels[0], mca_rates=rates[3])
        self.up3 = up(in_channels=channels[0], out_channels=channels[1])
        self.skip_enhancer3 = skip(dim=channels[1], num_heads=num_heads[0], input_size=input_size[0], depth=4, label="S14")
        self.dec3 = dec(embed_dims=channels[1], mca_rates=rates[2])
        self.up2 = up(in_channels=channels[1], out_channels=channels[2])
        self.skip_enhancer2 = skip(dim=channels[2], num_heads=num_heads[1], input_size=input_size[1], depth=3, label="S28")
        self.dec2 = dec(embed_dims=channels[2], mca_rates=rates[1])
        self.up1 = up(in_channels=channels[2], out_channels=channels[3])
        self.skip_enhancer1 = skip(dim=channels[3], num_heads=num_heads[2], input_size=input_size[2], depth=2, label="S56")
        self.dec1 = dec(embed_dims=channels[3], mca_rates=rates[0])

    def forward(self, x, skips):
        d = self.dec4(x)
        for lvl, s in zip((3, 2, 1), skips):
            d = getattr(self, f"up{lvl}")(d)
            se = getattr(self, f"skip_enhancer{lvl}")
            e = se(s, d)
            d, se.dec_tap = se.dec_tap, None  # (d itself, routed through the block's concat node: see DSEBlock.forward)
            d = getattr(self, f"dec{lvl}")(ops.add_act(d, e))
        return d
