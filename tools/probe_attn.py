"""Launches only the DSEB-56^2 differential-attention backward (dq + dkv kernels) a few times for rocprofv3 --pmc passes."""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch

from cenet_amd import kern, ops

kern.set_compute_bf16(True)
dev = torch.device("cuda:0")
B, N, H, hd = 32, 3136, 4, 16
E = 2 * H * hd
q, k, v = (torch.randn(B, N, E, device=dev, requires_grad=True) for _ in range(3))
U = ops.diff_attention_heads(q, k, v, H)
g = torch.randn_like(U)
for _ in range(3):
    ops.diff_attention_heads(q, k, v, H).backward(g)
torch.cuda.synchronize()
