"""Launches the direct-conv forward / data-gradient kernels of the output head alone (timing and rocprofv3 counter passes)."""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch

from cenet_amd import kern, ops

kern.set_compute_bf16(True)
dev = torch.device("cuda:0")
for (Cin, Cout, H, k) in ((32, 32, 224, 5), (64, 64, 112, 3), (64, 32, 112, 3)):
    x = torch.randn(32, Cin, H, H, device=dev).bfloat16()
    w = (torch.randn(Cout, Cin, k, k, device=dev) * 0.03)
    with torch.no_grad():
        for _ in range(3):
            y = ops.conv2d_nchw(x, w, None, stride=1, pad=k // 2)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            y = ops.conv2d_nchw(x, w, None, stride=1, pad=k // 2)
        e1.record()
        torch.cuda.synchronize()
    fl = 2.0 * 32 * H * H * Cin * Cout * k * k
    t = e0.elapsed_time(e1) / 10
    print(f"fwd {Cin}->{Cout} {k}x{k} @{H}: {t * 1e3:.0f} us = {fl / t / 1e9:.0f} TFLOP/s", flush=True)
