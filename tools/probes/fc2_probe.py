"""stage-3 fc2 forward (x [6272, 1280] W [320, 1280]^T) with and without the fused epilogue terms.  python tools/probes/fc2_probe.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from cenet_amd import kern
dev = torch.device('cuda:0')
R, K, N, B = 6272, 1280, 320, 32
x = torch.randn(R, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
y = torch.empty(R, N, device=dev, dtype=torch.bfloat16); res = torch.randn(R, N, device=dev).bfloat16()
b = torch.randn(N, device=dev); bs = torch.rand(B, device=dev)
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
A, Bm = kern.mat_plain(x, K, 1, kfast=1), kern.mat_plain(W, 1, K, kfast=1)
print('plain        ', t(lambda: kern.gemm(A, Bm, y, R, N, K, scr=N, scc=1)), kern.last_gemm_kernel())
print('bias         ', t(lambda: kern.gemm(A, Bm, y, R, N, K, scr=N, scc=1, bias=b)))
print('bias+resid   ', t(lambda: kern.gemm(A, Bm, y, R, N, K, scr=N, scc=1, bias=b, R=res, srr=N, src=1)))
print('bias+res+bsc ', t(lambda: kern.gemm(A, Bm, y, R, N, K, scr=N, scc=1, bias=b, R=res, srr=N, src=1, bscale=bs, bscale_rows=R // B)))
print('bscale only  ', t(lambda: kern.gemm(A, Bm, y, R, N, K, scr=N, scc=1, bscale=bs, bscale_rows=R // B)))
