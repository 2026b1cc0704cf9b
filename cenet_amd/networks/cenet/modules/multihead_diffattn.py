"""Differential attention on HIP kernels — mirrors reference src/networks/cenet/modules/multihead_diffattn.py:28-129."""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from .... import ops


def lambda_init_fn(depth):
    return 0.8 - 0.6 * math.exp(-0.3 * depth)


class MultiheadDiffAttn(nn.Module):
    def __init__(self, embed_dim, depth, num_heads, model_parallel_size=1, decoder_kv_attention_heads=None, vis=False,
                 return_2=False):
        super().__init__()
        if model_parallel_size != 1 or decoder_kv_attention_heads is not None or vis or return_2:
            raise NotImplementedError("CENet uses MultiheadDiffAttn(embed_dim, depth, num_heads) only")
        self.embed_dim, self.num_heads = embed_dim, num_heads
        self.head_dim = embed_dim // num_heads // 2
        self.q_proj = nn.Linear(embed_dim, embed_dim, bias=False)
        self.k_proj = nn.Linear(embed_dim, embed_dim, bias=False)
        self.v_proj = nn.Linear(embed_dim, embed_dim, bias=False)
        self.out_proj = nn.Linear(embed_dim, embed_dim, bias=False)
        self.lambda_init = lambda_init_fn(depth)
        for n in ("lambda_q1", "lambda_k1", "lambda_q2", "lambda_k2"):
            setattr(self, n, nn.Parameter(torch.zeros(self.head_dim, dtype=torch.float32).normal_(mean=0, std=0.1)))

    def arena_groups(self):
        return [[self.q_proj.weight, self.k_proj.weight, self.v_proj.weight]]

    def _merged_qkv(self):
        """[3, E, E] alias of the q / k / v weights when a ParamArena laid them out back to back (arena_groups), else None"""
        Wm = getattr(self, "_wqkv", None)
        if Wm is None or Wm.data_ptr() != self.q_proj.weight.data_ptr():
            Wm = ops.merged_param([self.q_proj.weight, self.k_proj.weight, self.v_proj.weight])
            self._wqkv = Wm
        return Wm

    def forward(self, x, rel_pos=None, attn_mask=None, tap=False):
        """tap=True: returns (out, x_tap) — x routed through the projections' autograd node, for x's other consumers (DSEBlock)"""
        if rel_pos is not None or attn_mask is not None:
            raise NotImplementedError
        Wm = self._merged_qkv()
        if Wm is not None:  # the three projections as one batched launch per pass (weights back to back in the ParamArena)
            ops.refresh_member_shadows(Wm, x)
            q, k, v, x = ops.multi_linear(x, Wm, tap=True)
        else:
            q, x = ops.linear(x, self.q_proj.weight, tap=True)  # (taps: the three data gradients add up inside the GEMMs)
            k, x = ops.linear(x, self.k_proj.weight, tap=True)
            v, x = ops.linear(x, self.v_proj.weight, tap=True)
        U = ops.diff_attention_heads(q, k, v, self.num_heads)  # [B, 2H, N, 2hd], two softmaxes per head, tiled
        a = ops.diff_attention_combine(U, self.lambda_q1, self.lambda_k1, self.lambda_q2, self.lambda_k2, self.lambda_init)
        out = ops.linear(a, self.out_proj.weight)
        return (out, x) if tap else out
