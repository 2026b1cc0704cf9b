"""Micro-benchmark of the representative contractions of one CENet step (B=32) on the GEMM / implicit-GEMM core.
Usage (GPU box): python tools/gemm_bench.py [f32|bf16|both]"""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch

from cenet_amd import kern, ops

dev = torch.device("cuda:0")


def timeit(fn, reps=10):
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


def lin_cases():
    out = []
    for name, R, K, N in [("s1.fc1", 100352, 64, 512), ("s1.fc2", 100352, 512, 64), ("s2.fc1", 25088, 128, 1024),
                          ("s3.fc1", 6272, 320, 1280), ("s3.q", 6272, 320, 320), ("s4.fc1", 1568, 512, 2048),
                          ("dseb1.qproj", 100352, 128, 128)]:
        x = torch.randn(R, K, device=dev)
        W = torch.randn(N, K, device=dev) * 0.05
        g = torch.randn(R, N, device=dev)
        dx = torch.empty_like(x)
        dW = torch.zeros_like(W)
        y = torch.empty(R, N, device=dev)
        fl = 2.0 * R * K * N
        out.append((name + ".fwd", fl, lambda x=x, W=W, y=y, R=R, K=K, N=N: kern.gemm(
            kern.mat_plain(x, K, 1, kfast=1), kern.mat_plain(W, 1, K, kfast=1), y, R, N, K, scr=N, scc=1)))
        out.append((name + ".dx", fl, lambda g=g, W=W, dx=dx, R=R, K=K, N=N: kern.gemm(
            kern.mat_plain(g, N, 1, kfast=1), kern.mat_plain(W, K, 1, kfast=0), dx, R, K, N, scr=K, scc=1)))
        out.append((name + ".dW", fl, lambda g=g, x=x, dW=dW, R=R, K=K, N=N: kern.gemm(
            kern.mat_plain(g, 1, N, kfast=0), kern.mat_plain(x, K, 1, kfast=0), dW, N, K, R, scr=K, scc=1,
            splits=kern.pick_splits(N, K, 1, (R + 31) // 32), atomic=True)))
    return out


def conv1x1_cases():
    out = []
    B = 32
    for name, Cin, Cout, HW in [("dec1.gate", 64, 64, 3136), ("dec1.mlp.fc1", 64, 256, 3136), ("dec1.mlp.fc2", 256, 64, 3136),
                                ("dec2.mlp.fc1", 128, 512, 784), ("dec4.mlp.fc1", 512, 2048, 49)]:
        x = torch.randn(B, Cin, HW, 1, device=dev, requires_grad=True)
        W = torch.nn.Parameter(torch.randn(Cout, Cin, 1, 1, device=dev) * 0.05)
        fl = 2.0 * B * Cin * Cout * HW
        y = ops.conv1x1(x, W)
        gy = torch.randn_like(y)
        out.append((name + ".fwd", fl, lambda x=x, W=W: ops.conv1x1(x.detach(), W.detach())))
        out.append((name + ".bwd(dx+dW)", 2 * fl, lambda x=x, W=W, gy=gy: ops.conv1x1(x, W).backward(gy)))
    return out


def conv_cases():
    out = []
    B = 32
    for name, Cin, Cout, H, k in [("out.rb.conv2 5x5 32->32@224", 32, 32, 224, 5), ("out.out.conv 3x3 64->64@112", 64, 64, 112, 3),
                                  ("out.up.conv 3x3 64->32@112", 64, 32, 112, 3), ("out.rb.conv1 5x5 1->32@224", 1, 32, 224, 5)]:
        x = torch.randn(B, Cin, H, H, device=dev, requires_grad=True)
        W = torch.nn.Parameter(torch.randn(Cout, Cin, k, k, device=dev) * 0.05)
        fl = 2.0 * B * Cin * Cout * H * H * k * k
        y = ops.conv2d_nchw(x, W, None, 1, k // 2)
        gy = torch.randn_like(y)
        out.append((name + ".fwd", fl, lambda x=x, W=W, k=k: ops.conv2d_nchw(x.detach(), W.detach(), None, 1, k // 2)))
        out.append((name + ".fwd+bwd", 3 * fl, lambda x=x, W=W, k=k, gy=gy: ops.conv2d_nchw(x, W, None, 1, k // 2).backward(gy)))
    return out


def main():
    modes = {"f32": [False], "bf16": [True], "both": [False, True]}[sys.argv[1] if len(sys.argv) > 1 else "both"]
    cases = lin_cases() + conv1x1_cases() + conv_cases()
    print(f"{'case':44s} " + " ".join(f"{'bf16' if m else 'f32':>18s}" for m in modes))
    for name, fl, fn in cases:
        cols = []
        for m in modes:
            kern.set_compute_bf16(m)
            ms = timeit(fn)
            cols.append(f"{ms:8.3f}ms {fl / ms / 1e9:7.1f}TF")
        print(f"{name:44s} " + " ".join(cols))
    kern.set_compute_bf16(False)


if __name__ == "__main__":
    main()
