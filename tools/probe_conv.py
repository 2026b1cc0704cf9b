"""Launches only the roofline kernel of bench.py (out.rb.0.conv2 forward, 5x5 32->32 @224^2, B=32) a few times so that
rocprofv3 --pmc passes can attribute FETCH_SIZE / WRITE_SIZE to it.  Usage: python tools/probe_conv.py [f32|bf16]"""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch

from cenet_amd import kern, ops

kern.set_compute_bf16(len(sys.argv) > 1 and sys.argv[1] == "bf16")
dev = torch.device("cuda:0")
x = torch.randn(32, 32, 224, 224, device=dev)
w = torch.randn(32, 32, 5, 5, device=dev) * 0.03
with torch.no_grad():
    for _ in range(4):
        y = ops.conv2d_nchw(x, w, None, stride=1, pad=2)
torch.cuda.synchronize()
print("algorithmic bytes per launch:", (x.numel() + w.numel() + y.numel()) * 4)
