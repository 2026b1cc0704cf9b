"""Golden-vector case table (TEST INFRASTRUCTURE).

Shared by `oracle/gen_golden.py` (which instantiates the *reference* class named in `ref` through the
timm/monai shim, in the build container only) and by the tests (which call the `oracle` restatement and
the HIP product path on the stored inputs/weights). Nothing here imports the reference.

Each case: name, ref=(module, class, ctor kwargs), inputs (shapes), call (how the reference forward is
invoked), oracle(sd, ins, training) -> output tensor.  All module parameters live under prefix "m".
"""
from __future__ import annotations

from functools import partial

import torch.nn as nn

from . import cenet_oracle as O

P = "m"


def _case(name, ref, inputs, oracle, call=None, train=True, scale=1.0):
    return dict(name=name, ref=ref, inputs=inputs, oracle=oracle, call=call, train=train, scale=scale)


def _ln6():
    return partial(nn.LayerNorm, eps=1e-6)


CASES = [
    # ---- PVTv2 pieces (pvtv2.py) ----
    _case("patch_embed_k7s4", ("networks.cenet.pvtv2", "OverlapPatchEmbed",
                               dict(img_size=32, patch_size=7, stride=4, in_chans=3, embed_dim=16)),
          [(2, 3, 32, 32)], lambda sd, ins, tr: O.overlap_patch_embed(sd, P, ins[0], 7, 4)[0],
          call=lambda m, ins: m(ins[0])[0]),
    _case("patch_embed_k3s2", ("networks.cenet.pvtv2", "OverlapPatchEmbed",
                               dict(img_size=16, patch_size=3, stride=2, in_chans=16, embed_dim=24)),
          [(2, 16, 14, 14)], lambda sd, ins, tr: O.overlap_patch_embed(sd, P, ins[0], 3, 2)[0],
          call=lambda m, ins: m(ins[0])[0]),
    _case("pvt_block_sr2", ("networks.cenet.pvtv2", "Block",
                            dict(dim=32, num_heads=2, mlp_ratio=4, qkv_bias=True, sr_ratio=2, norm_layer="LN6")),
          [(2, 64, 32)], lambda sd, ins, tr: O.pvt_block(sd, P, ins[0], 8, 8, 2, 2, 0.0, None),
          call=lambda m, ins: m(ins[0], 8, 8)),
    _case("pvt_block_sr1", ("networks.cenet.pvtv2", "Block",
                            dict(dim=48, num_heads=3, mlp_ratio=2, qkv_bias=True, sr_ratio=1, norm_layer="LN6")),
          [(2, 20, 48)], lambda sd, ins, tr: O.pvt_block(sd, P, ins[0], 4, 5, 3, 1, 0.0, None),
          call=lambda m, ins: m(ins[0], 4, 5)),
    # ---- CFAM pieces (cfam.py / nlb.py / blocks.py) ----
    _case("ccu_b2", ("networks.cenet.modules.cfam", "CCU", dict(channel=12)),
          [(2, 12, 6, 5)], lambda sd, ins, tr: O.ccu(sd, P, ins[0], tr)),
    _case("ccu_b1", ("networks.cenet.modules.cfam", "CCU", dict(channel=12)),
          [(1, 12, 6, 5)], lambda sd, ins, tr: O.ccu(sd, P, ins[0], tr)),
    _case("srm", ("networks.cenet.modules.cfam", "SRM", dict()),
          [(2, 10, 6, 7)], lambda sd, ins, tr: O.srm(sd, P, ins[0], tr)),
    _case("multi_order_dw_14", ("networks.cenet.modules.cfam", "MultiOrderDWConv",
                                dict(embed_dims=32, rates=[1, 2, 3])),
          [(2, 32, 14, 14)], lambda sd, ins, tr: O.multi_order_dwconv(sd, P, ins[0], (1, 2, 3), tr)),
    _case("multi_order_dw_7", ("networks.cenet.modules.cfam", "MultiOrderDWConv",
                               dict(embed_dims=32, rates=[1, 2, 2])),
          [(2, 32, 7, 7)], lambda sd, ins, tr: O.multi_order_dwconv(sd, P, ins[0], (1, 2, 2), tr)),
    _case("nonlocal", ("networks.cenet.modules.nlb", "Nonlocal", dict(dim_inner=16)),
          [(2, 16, 6, 5)], lambda sd, ins, tr: O.nonlocal_block(sd, P, ins[0], tr)),
    _case("cfam_mlp", ("networks.cenet.modules.cfam", "Mlp", dict(embed_dims=8, feedforward_channels=32)),
          [(2, 8, 6, 6)], lambda sd, ins, tr: O.cfam_mlp(sd, P, ins[0], tr)),
    _case("cfa_module_14", ("networks.cenet.modules.cfam", "CFAModule",
                            dict(embed_dims=32, mca_rates=[1, 2, 3], init_value=1e-6)),
          [(2, 32, 14, 14)], lambda sd, ins, tr: O.cfa_module(sd, P, ins[0], (1, 2, 3), tr)),
    _case("cfa_module_7", ("networks.cenet.modules.cfam", "CFAModule",
                           dict(embed_dims=16, mca_rates=[1, 2, 2], init_value=1e-6)),
          [(3, 16, 7, 7)], lambda sd, ins, tr: O.cfa_module(sd, P, ins[0], (1, 2, 2), tr)),
    # ---- up blocks ----
    _case("eucb", ("networks.cenet.modules.blocks", "EUCB",
                   dict(in_channels=16, out_channels=8, kernel_size=3, stride=1, activation="leakyrelu")),
          [(2, 16, 5, 6)], lambda sd, ins, tr: O.eucb(sd, P, ins[0], tr)),
    _case("upconv", ("networks.cenet.modules.blocks", "UpConv",
                     dict(in_channels=16, out_channels=8, kernel_size=3, stride=1, activation="leakyrelu")),
          [(2, 16, 5, 6)], lambda sd, ins, tr: O.up_conv(sd, P, ins[0], tr)),
    # ---- DSEB pieces ----
    _case("fea_1_0.5_14", ("networks.cenet.modules.dseb", "FEA", dict(dim=6, scale_factors=[1.0, 0.5])),
          [(2, 6, 14, 14)], lambda sd, ins, tr: O.fea(sd, P, ins[0], (1.0, 0.5))),
    _case("fea_0.8_0.4_14", ("networks.cenet.modules.dseb", "FEA", dict(dim=6, scale_factors=[0.8, 0.4])),
          [(2, 6, 14, 14)], lambda sd, ins, tr: O.fea(sd, P, ins[0], (0.8, 0.4))),
    _case("fea_0.8_0.4_28", ("networks.cenet.modules.dseb", "FEA", dict(dim=4, scale_factors=[0.8, 0.4])),
          [(1, 4, 28, 28)], lambda sd, ins, tr: O.fea(sd, P, ins[0], (0.8, 0.4))),
    _case("fea_3scales_28", ("networks.cenet.modules.dseb", "FEA", dict(dim=4, scale_factors=[1.0, 0.75, 0.5])),
          [(1, 4, 28, 28)], lambda sd, ins, tr: O.fea(sd, P, ins[0], (1.0, 0.75, 0.5))),
    _case("diffattn_hd8", ("networks.cenet.modules.multihead_diffattn", "MultiheadDiffAttn",
                           dict(embed_dim=32, depth=2, num_heads=2)),
          [(2, 24, 32)], lambda sd, ins, tr: O.multihead_diff_attn(sd, P, ins[0], 2, 2)),
    _case("diffattn_hd20", ("networks.cenet.modules.multihead_diffattn", "MultiheadDiffAttn",
                            dict(embed_dim=80, depth=4, num_heads=2)),
          [(1, 40, 80)], lambda sd, ins, tr: O.multihead_diff_attn(sd, P, ins[0], 2, 4)),
    _case("dseb", ("networks.cenet.modules.dseb", "DSEBlock",
                   dict(dim=8, scale_factors=[1.0, 0.5], num_heads=2, input_size=8, mode="cat", depth=3)),
          [(2, 8, 8, 8), (2, 8, 8, 8)],
          lambda sd, ins, tr: O.dse_block(sd, P, ins[0], ins[1], (1.0, 0.5), 2, 3),
          call=lambda m, ins: m(ins[0], ins[1])),
    _case("dseb_synapse", ("networks.cenet.modules.dseb", "DSEBlock",
                           dict(dim=16, scale_factors=[0.8, 0.4], num_heads=4, input_size=10, mode="cat", depth=2)),
          [(2, 16, 10, 10), (2, 16, 10, 10)],
          lambda sd, ins, tr: O.dse_block(sd, P, ins[0], ins[1], (0.8, 0.4), 4, 2),
          call=lambda m, ins: m(ins[0], ins[1])),
    # ---- OutHead pieces ----
    _case("resblock_k5", ("networks.cenet.modules.unet", "UnetResBlock",
                          dict(spatial_dims=2, in_channels=1, out_channels=8, kernel_size=5, stride=1,
                               norm_name="batch", dropout=0)),
          [(2, 1, 12, 12)], lambda sd, ins, tr: O.unet_res_block(sd, P, ins[0], 5, tr)),
    _case("resblock_k3", ("networks.cenet.modules.unet", "UnetResBlock",
                          dict(spatial_dims=2, in_channels=8, out_channels=8, kernel_size=3, stride=1,
                               norm_name="batch", dropout=0)),
          [(2, 8, 9, 9)], lambda sd, ins, tr: O.unet_res_block(sd, P, ins[0], 3, tr)),
    _case("out_head", ("networks.cenet.out", "OutHead",
                       dict(dec_in_channels=16, x_in_channels=1, out_channels=4, merge_mode="cat", up_block="upcn",
                            up_ks=3)),
          [(2, 16, 8, 8), (2, 1, 32, 32)], lambda sd, ins, tr: O.out_head(sd, P, ins[0], ins[1], tr),
          call=lambda m, ins: m(ins[0], ins[1])),
    # ---- appended in round 3 (case index = RNG seed: new cases go to the END so the older fixtures stay reproducible) ----
    _case("mca_14", ("networks.cenet.modules.cfam", "MCA", dict(embed_dims=32, rates=[1, 2, 3])),
          [(2, 32, 14, 14)], lambda sd, ins, tr: O.mca(sd, P, ins[0], (1, 2, 3), tr)),
    _case("mca_7", ("networks.cenet.modules.cfam", "MCA", dict(embed_dims=16, rates=[1, 2, 2])),
          [(3, 16, 7, 7)], lambda sd, ins, tr: O.mca(sd, P, ins[0], (1, 2, 2), tr)),
    # ---- appended in round 5: SURVEY G1's two remaining MultiheadDiffAttn head dimensions (16: the 56x56 level of the ACDC
    # preset and 14x14 of Synapse; 80: ACDC 14x14 — until now pinned only through the whole-model goldens) ----
    _case("diffattn_hd16", ("networks.cenet.modules.multihead_diffattn", "MultiheadDiffAttn",
                            dict(embed_dim=64, depth=2, num_heads=2)),
          [(2, 49, 64)], lambda sd, ins, tr: O.multihead_diff_attn(sd, P, ins[0], 2, 2)),
    _case("diffattn_hd80", ("networks.cenet.modules.multihead_diffattn", "MultiheadDiffAttn",
                            dict(embed_dim=160, depth=4, num_heads=1)),
          [(1, 36, 160)], lambda sd, ins, tr: O.multihead_diff_attn(sd, P, ins[0], 1, 4)),
]

# multihead_diffattn.py:106 applies torch.nan_to_num to the scores.  A case whose q.k products overflow fp32 (q_proj / k_proj
# weights scaled by `wscale`, inputs and values ordinary): the reference's forward stays finite and non-trivial; eval output only
# (its gradients are meaningless).  Generated by gen_golden.gen_nonfinite_case.
NONFINITE_CASE = dict(name="diffattn_nonfinite", ref=("networks.cenet.modules.multihead_diffattn", "MultiheadDiffAttn",
                                                      dict(embed_dim=32, depth=2, num_heads=2)),
                      inputs=[(1, 12, 32)], wscale=1e19, seed=4242,
                      oracle=lambda sd, ins, tr: O.multihead_diff_attn(sd, P, ins[0], 2, 2))

CASE_BY_NAME = {c["name"]: c for c in CASES}

# whole-model configurations: name -> CENet kwargs, batch
MODEL_CONFIGS = {
    "acdc": dict(kw=dict(input_channels=1, num_classes=4, scale_factors=[1.0, 0.5], diffatt_num_heads=[4, 4, 4],
                         out_up_block="upcn"), batch=2),
    "synapse": dict(kw=dict(input_channels=1, num_classes=9, scale_factors=[0.8, 0.4],
                            diffatt_num_heads=[16, 8, 8], out_up_block="upcn"), batch=2),
    "skin": dict(kw=dict(input_channels=3, num_classes=2, scale_factors=[1.0, 0.75, 0.5],
                         diffatt_num_heads=[2, 2, 2], out_up_block="upcn"), batch=2),
}


def config_from_kwargs(kw) -> O.CENetConfig:
    return O.CENetConfig(input_channels=kw["input_channels"], num_classes=kw["num_classes"],
                         scale_factors=tuple(kw["scale_factors"]), diffatt_num_heads=tuple(kw["diffatt_num_heads"]))
