"""How a ring-GEMM launch's time splits into a fixed part and a per-K-step part: y[R, N] = x[R, K] W[N, K]^T (bf16) timed
back to back for a scan over K at several (R, N).   python tools/probes/gemm_kscan.py"""
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


for R, N in ((6272, 320), (6272, 1280), (25088, 128), (100352, 64), (1568, 512)):
    row = []
    for K in (64, 128, 256, 512, 1024, 2048, 4096):
        x = torch.randn(R, K, device=dev).bfloat16()
        W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        y = torch.empty(R, N, device=dev, dtype=torch.bfloat16)
        t = timeit(lambda: kern.gemm(kern.mat_plain(x, K, 1, kfast=1), kern.mat_plain(W, 1, K, kfast=1), y, R, N, K, scr=N, scc=1))
        row.append(f"K{K}: {t:6.1f}us {2.0 * R * N * K / t / 1e6:5.0f}TF [{kern.last_gemm_kernel().split('<')[1][:22]}]")
    print(f"R{R} N{N}: " + " | ".join(row))
