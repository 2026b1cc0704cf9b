"""Times the token-layout depthwise 3x3 kernels of the PVTv2 Mlp (forward + GELU, activation backward, data gradient,
weight gradient) at the four stage shapes of the ACDC preset, B = 32, bf16 tensors; prints time and effective HBM rate
against the algorithmic bytes.  python tools/dw_bench.py"""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch

from cenet_amd import kern

dev = torch.device("cuda:0")


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


B = 32
for H, C in ((56, 512), (28, 1024), (14, 1280), (7, 2048)):
    n = B * H * H * C
    x = torch.randn(B, H * H, C, device=dev).bfloat16()
    g = torch.randn_like(x)
    w = torch.randn(C, 1, 3, 3, device=dev) * 0.2
    b = torch.randn(C, device=dev) * 0.1
    u, a, gu, dx = (torch.empty_like(x) for _ in range(4))
    dw, db = torch.zeros_like(w), torch.zeros_like(b)
    mb = n * 2 / 1e6
    rows = [("fwd+gelu (x -> a)", 2 * mb, lambda: kern.dw_tok(x, w, b, None, a, B, C, H, H, 0, "gelu")),
            ("bwd_pre (x, g -> gu, dw)", 3 * mb, lambda: kern.dw_tok_bwd_pre(x, g, w, b, gu, dw, db, B, C, H, H, "gelu")),
            ("fwd+gelu (x -> u, a)", 3 * mb, lambda: kern.dw_tok(x, w, b, u, a, B, C, H, H, 0, "gelu")),
            ("act_bwd (u, g -> gu)", 3 * mb, lambda: kern.act_bwd(u, g, gu, n, "gelu")),
            ("dgrad (gu -> dx)", 2 * mb, lambda: kern.dw_tok(gu, w, None, dx, None, B, C, H, H, 1)),
            ("wgrad (x, gu -> dw, db)", 2 * mb, lambda: kern.dw_wgrad_tok(x, gu, dw, db, B, C, H, H))]
    for name, mbytes, fn in rows:
        t = timeit(fn)
        print(f"{H:3d}x{H:<3d} C={C:5d} {name:26s} {t:8.1f} us  {mbytes / t:7.2f} TB/s  (ideal {mbytes / 8.0:6.1f} us)", flush=True)
