"""Non-local block on HIP kernels — mirrors reference src/networks/cenet/modules/nlb.py:6-148."""
from __future__ import annotations

import torch
import torch.nn as nn

from .... import ops
from .blocks import bn_call


class Nonlocal(nn.Module):
    def __init__(self, dim_inner, pool_size=None, instantiation="softmax", zero_init_final_conv=False,
                 zero_init_final_norm=True, norm_eps=1e-5, norm_momentum=0.1, norm_module=nn.BatchNorm2d):
        super().__init__()
        if pool_size is not None or instantiation != "softmax":
            raise NotImplementedError("CENet uses Nonlocal(softmax, no pooling) only")
        self.dim_inner = dim_inner
        self.conv_theta = nn.Conv2d(dim_inner, dim_inner, 1)
        self.conv_phi = nn.Conv2d(dim_inner, dim_inner, 1)
        self.conv_g = nn.Conv2d(dim_inner, dim_inner, 1)
        self.conv_out = nn.Conv2d(dim_inner, dim_inner, 1)
        self.bn = norm_module(num_features=dim_inner, eps=norm_eps, momentum=norm_momentum)
        self.w = nn.Parameter(torch.tensor(0.5))

    def arena_groups(self):
        return [[self.conv_theta.weight, self.conv_phi.weight, self.conv_g.weight],
                [self.conv_theta.bias, self.conv_phi.bias, self.conv_g.bias]]

    def _merged_tpg(self):
        """([3C, C, 1, 1] weight, [3C] bias) aliasing conv_theta / conv_phi / conv_g when a ParamArena laid them out back to back
        (arena_groups), else None: the three projections then run as ONE 1x1 conv whose output the attention reads in place"""
        m = getattr(self, "_wtpg", None)
        if m is None or m[0] is None or m[0].data_ptr() != self.conv_theta.weight.data_ptr() \
                or m[1].data_ptr() != self.conv_theta.bias.data_ptr():
            Wm = ops.merged_param([self.conv_theta.weight, self.conv_phi.weight, self.conv_g.weight], flat=True)
            bm = ops.merged_param([self.conv_theta.bias, self.conv_phi.bias, self.conv_g.bias], flat=True)
            m = (Wm, bm) if Wm is not None and bm is not None else (None, None)
            self._wtpg = m
        return m if m[0] is not None else None

    def forward(self, x, raw=False):
        """raw=True: stop in front of the BatchNorm -> (conv_out output, x as the residual mix wants it): the caller fuses the
        normalisation and the mix with what follows (CFAModule, ops.cfam_mid)"""
        m = self._merged_tpg()
        if m is not None:
            ops.refresh_member_shadows(m[0], x)
            tpg, x = ops.conv1x1(x, m[0], m[1], tap=True)  # (tap: the residual mix below reads x)
            y = ops.nonlocal_attention_joint(tpg)
            p = ops.conv1x1(y, self.conv_out.weight, self.conv_out.bias)
            if raw:
                return p, x
            p = bn_call(self.bn, p)
            return ops.mix(x, p, self.w)
        # x has four consumers (theta, phi, g, the residual mix): each 1x1 conv hands x on as a tap, so the four gradients meet
        # inside the data-gradient GEMMs instead of three aten::add launches
        theta, x = ops.conv1x1(x, self.conv_theta.weight, self.conv_theta.bias, tap=True)
        phi, x = ops.conv1x1(x, self.conv_phi.weight, self.conv_phi.bias, tap=True)
        g, x = ops.conv1x1(x, self.conv_g.weight, self.conv_g.bias, tap=True)
        y = ops.nonlocal_attention(theta, phi, g)  # flash-style: the N x N map is never materialised
        p = ops.conv1x1(y, self.conv_out.weight, self.conv_out.bias)
        if raw:
            return p, x
        p = bn_call(self.bn, p)
        return ops.mix(x, p, self.w)
