"""Launches the direct-conv weight-gradient kernels of the output head alone (for rocprofv3 counter passes / timing):
5x5 32->32 @224^2, 3x3 64->64 @112^2, 3x3 64->32 @112^2, B = 32.   python tools/probe_wgrad.py"""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch

from cenet_amd import kern

dev = torch.device("cuda:0")
for (Cin, Cout, H, k) in ((32, 32, 224, 5), (64, 64, 112, 3), (64, 32, 112, 3)):
    x = torch.randn(32, Cin, H, H, device=dev).bfloat16()
    dy = torch.randn(32, Cout, H, H, device=dev).bfloat16()
    dw = torch.zeros(Cout, Cin, k, k, device=dev)
    for _ in range(3):
        kern.conv_wgrad_direct(x, dy, dw, 32, Cin, Cout, H, H, k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        kern.conv_wgrad_direct(x, dy, dw, 32, Cin, Cout, H, H, k)
    e1.record()
    torch.cuda.synchronize()
    fl = 2.0 * 32 * H * H * Cin * Cout * k * k
    t = e0.elapsed_time(e1) / 10
    print(f"{Cin}->{Cout} {k}x{k} @{H}: {t * 1e3:.0f} us = {fl / t / 1e9:.0f} TFLOP/s", flush=True)
