"""Times the differential-attention pair kernels (attn_diff.hip) on the three DSEB problems of the ACDC preset at B = 32,
forward and backward, with HIP events; prints TFLOP/s against the dense bf16 MFMA peak.  python tools/dattn_bench.py"""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch

from cenet_amd import ops

dev = torch.device("cuda:0")


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for B, N, H, hd in ((32, 3136, 4, 16), (32, 784, 4, 32), (24, 3136, 8, 8), (24, 784, 8, 16)):
    E = 2 * H * hd
    q, k, v = (torch.randn(B, N, E, device=dev).bfloat16().requires_grad_(True) for _ in range(3))
    U = ops.diff_attention_heads(q, k, v, H)
    g = torch.randn_like(U)
    tf = timeit(lambda: ops.diff_attention_heads(q.detach(), k.detach(), v.detach(), H))
    tfb = timeit(lambda: ops.diff_attention_heads(q, k, v, H).backward(g))
    fl_f = 2.0 * B * 2 * H * N * N * (hd + 2 * hd)
    fl_b = 2.0 * B * 2 * H * N * N * (hd + 2 * hd) * 2.5
    print(f"B={B} N={N} H={H} hd={hd}: fwd {tf:.3f} ms = {fl_f / tf / 1e9:.0f} TF ({fl_f / tf / 1e9 / 2500:.3f} of peak); "
          f"bwd {tfb - tf:.3f} ms = {fl_b / (tfb - tf) / 1e9:.0f} TF ({fl_b / (tfb - tf) / 1e9 / 2500:.3f})", flush=True)
