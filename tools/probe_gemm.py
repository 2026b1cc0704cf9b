"""Launches two representative bf16 GEMMs a few times for rocprofv3 --pmc passes: stage-3 fc2-like (6272x320, K=1280; 64x64
tiles, K step 64) and stage-1 fc1 (100352x512, K=64; 128x128 tiles)."""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch

from cenet_amd import kern

kern.set_compute_bf16(True)
dev = torch.device("cuda:0")
for (R, K, N) in [(6272, 1280, 320), (100352, 64, 512)]:
    x = torch.randn(R, K, device=dev)
    W = torch.randn(N, K, device=dev) * 0.05
    y = torch.empty(R, N, device=dev)
    for _ in range(4):
        kern.gemm(kern.mat_plain(x, K, 1, kfast=1), kern.mat_plain(W, 1, K, kfast=1), y, R, N, K, scr=N, scc=1)
torch.cuda.synchronize()
