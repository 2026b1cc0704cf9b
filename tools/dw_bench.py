"""Micro-benchmark of the token-layout depthwise 3x3 kernels at the PVT Mlp shapes (B=32)."""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch

from cenet_amd import kern

dev = torch.device("cuda:0")


def t(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


B = 32
for H, C in [(56, 512), (28, 1024), (14, 1280), (7, 2048)]:
    x = torch.randn(B, H * H, C, device=dev)
    w, b = torch.randn(C, 9, device=dev), torch.randn(C, device=dev)
    y, a = torch.empty_like(x), torch.empty_like(x)
    dw, db = torch.zeros(C, 9, device=dev), torch.zeros(C, device=dev)
    mb = x.numel() * 4 / 1e6
    tf = t(lambda: kern.dw_tok(x, w, b, y, a, B, C, H, H, False, act="gelu"))
    td = t(lambda: kern.dw_tok(x, w, None, y, None, B, C, H, H, True))
    tw = t(lambda: kern.dw_wgrad_tok(x, y, dw, db, B, C, H, H))
    print(f"H={H} C={C} ({mb:.0f} MB/tensor): fwd {tf*1e3:.0f} us ({3*mb/tf/1e3:.2f} TB/s)  dgrad {td*1e3:.0f} us ({2*mb/td/1e3:.2f} TB/s)"
          f"  wgrad {tw*1e3:.0f} us ({2*mb/tw/1e3:.2f} TB/s)")
