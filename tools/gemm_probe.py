"""Per-iteration cost of the GEMM main loop: weight-gradient shape (M x N outputs, K = tokens, both operands row-contiguous)
run un-split, so time / K-tiles = one workgroup's serial iteration time."""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch

from cenet_amd import kern

dev = torch.device("cuda:0")
kern.set_compute_bf16(True)


def t(fn, reps=5):
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


for (M, N, R) in [(320, 1280, 6272), (128, 128, 6272), (64, 64, 6272), (1280, 1280, 6272)]:
    g = torch.randn(R, M, device=dev)
    x = torch.randn(R, N, device=dev)
    dW = torch.zeros(M, N, device=dev)
    for splits in (1, 2, 18):
        ms = t(lambda: kern.gemm(kern.mat_plain(g, 1, M, kfast=0), kern.mat_plain(x, N, 1, kfast=0), dW, M, N, R, scr=N, scc=1,
                                 splits=splits, atomic=True))
        print(f"wgrad M={M} N={N} K={R} splits_req={splits}: {ms*1e3:.1f} us  ({ms*1e3/(R/32):.3f} us per 32-k tile if unsplit)")
    # forward-like: x[R,N] @ W[M,N]^T, both k-contiguous
    W = torch.randn(M, N, device=dev)
    y = torch.empty(R, M, device=dev)
    ms = t(lambda: kern.gemm(kern.mat_plain(x, N, 1, kfast=1), kern.mat_plain(W, 1, N, kfast=1), y, R, M, N, scr=M, scc=1))
    print(f"fwd   R={R} out={M} K={N}: {ms*1e3:.1f} us")
