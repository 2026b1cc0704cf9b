"""Grouped weight-gradient kernel (gemm_group.hip) on the weight-gradient problems of one ACDC training step (B = 32), stage by
stage: time per launch set (HIP events over repeated launches), algorithmic bytes / flops, achieved GB/s and TFLOP/s.
Usage: python tools/wgrad_bench.py [reps]      (env CENET_GROUP_DEPTH / CENET_GROUP_ITEMS tune the K-slice plan)"""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch

from cenet_amd import kern

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
B = 32
BF = torch.bfloat16


def lin(R, N, K, bias=True):
    """token-major Linear: dW [N, K] += dY[R, N]^T X[R, K]"""
    return dict(kind="lin", dy=torch.randn(R, N, device=dev, dtype=BF), x=torch.randn(R, K, device=dev, dtype=BF),
                dW=torch.zeros(N, K, device=dev), db=torch.zeros(N, device=dev) if bias else None, M=N, N=K, K=R, nkb=1)


def conv(Cout, Cin, HW, bias=True):
    return dict(kind="conv", dy=torch.randn(B, Cout, HW, device=dev, dtype=BF), x=torch.randn(B, Cin, HW, device=dev, dtype=BF),
                dW=torch.zeros(Cout, Cin, device=dev), db=torch.zeros(Cout, device=dev) if bias else None, M=Cout, N=Cin, K=HW, nkb=B)


def block(C, N, r, sr, depth):
    R = B * N
    ps = []
    for _ in range(depth):
        ps += [lin(R, C, C), lin(B * 49, 2 * C, C), lin(R, C, C), lin(R, r * C, C), lin(R, C, r * C)]
        if sr > 1:
            ps.append(lin(B * 49, C, C * sr * sr))
    return ps


def cfam(C, HW):
    return [conv(C, C, HW), conv(C, C, HW), conv(C, C, HW), conv(3 * C, C, HW), conv(C, C, HW), conv(4 * C, C, HW), conv(C, 4 * C, HW)]


SETS = {
    "stage1": lambda: block(64, 3136, 8, 8, 3),
    "stage2": lambda: block(128, 784, 8, 4, 4),
    "stage3": lambda: block(320, 196, 4, 2, 6),
    "stage4": lambda: block(512, 49, 4, 1, 3),
    "dseb": lambda: [lin(B * 196, 640, 640, False) for _ in range(4)] + [lin(B * 784, 256, 256, False) for _ in range(4)]
    + [lin(B * 3136, 128, 128, False) for _ in range(4)],
    "decoder_1x1": lambda: cfam(512, 49) + cfam(320, 196) + cfam(128, 784) + cfam(64, 3136),
}


def items(ps):
    out = []
    for p in ps:
        db = p["db"].data_ptr() if p["db"] is not None else None
        if p["kind"] == "lin":
            out.append((p["dy"].data_ptr(), p["x"].data_ptr(), p["dW"].data_ptr(), db, p["M"], p["N"], 0, 0, p["M"], p["N"], p["K"], 1, 0))
        else:
            out.append((p["dy"].data_ptr(), p["x"].data_ptr(), p["dW"].data_ptr(), db, p["K"], p["K"], p["M"] * p["K"], p["N"] * p["K"],
                        p["M"], p["N"], p["K"], p["nkb"], 1))
    return out


only = os.environ.get("WGRAD_SETS")
for name, mk in SETS.items():
    if only and name not in only.split(","):
        continue
    ps = mk()
    it = items(ps)
    fl = sum(2.0 * p["M"] * p["N"] * p["K"] * p["nkb"] for p in ps)
    by = sum(2.0 * p["K"] * p["nkb"] * (p["M"] + p["N"]) + 8.0 * p["M"] * p["N"] for p in ps)
    for _ in range(2):
        kern.wgrad_group(it, dev)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        kern.wgrad_group(it, dev)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name:12s} {len(ps):3d} problems  {ms * 1e3:8.1f} us  {by / 1e6:8.1f} MB  {by / ms / 1e6:7.0f} GB/s  {fl / ms / 1e9:7.1f} TFLOP/s"
          f"   (ideal at 4 TB/s {by / 4e6:6.1f} us)", flush=True)
    del ps, it
    torch.cuda.empty_cache()
