"""flat weight gradients dW[N, K] += g[R, N]^T x[R, K] (bf16 operands, fp32 atomic accumulate) at the step's shapes, for the
split multiplier given in CENET_RING_SPLIT_MUL / tile in CENET_RING_TILE (read once per process).
python tools/probes/wgrad_split_probe.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from cenet_amd import kern
dev = torch.device('cuda:0')
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
out = []
for N, K, R in ((320, 320, 6272), (320, 1280, 6272), (1280, 320, 6272), (128, 128, 25088), (1024, 128, 25088), (64, 64, 100352),
                (512, 64, 100352), (512, 512, 1568), (320, 1280, 1568)):
    g = torch.randn(R, N, device=dev).bfloat16(); x = torch.randn(R, K, device=dev).bfloat16(); dW = torch.zeros(N, K, device=dev)
    us = t(lambda: kern.gemm(kern.mat_plain(g, 1, N, kfast=0), kern.mat_plain(x, K, 1, kfast=0), dW, N, K, R, scr=K, scc=1,
                             splits=kern.pick_splits(N, K, 1, (R + 31) // 32), atomic=True))
    out.append(f"{N}x{K}x{R}: {us:5.1f}")
print(os.environ.get("CENET_RING_SPLIT_MUL", "1"), os.environ.get("CENET_RING_TILE", "-"), " | ".join(out))
