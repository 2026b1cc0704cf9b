"""Records every GEMM launch of one bf16 training step (B=32) with its operand tensors kept alive, then REPLAYS the most
expensive distinct shapes in isolation (HIP events around 20 back-to-back launches of the same call): per-shape time without
launch gaps, against max(bytes / 6 TB/s, flops / 2.5 PF).  Usage: python tools/gemm_replay.py [top_n]"""
import argparse
import collections
import os
import sys

sys.path.insert(0, os.getcwd())
import torch

import bench
from cenet_amd import kern, losses, optim

top_n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
kern.set_compute_bf16(True)
net = bench.make_model(dev)
x, lab = bench.synthetic(32, dev, 0)
crit = losses.Criterion(4, argparse.Namespace(loss_type="dice,ce", loss_weights="0.5,0.5"))
arena = optim.ParamArena(net, optim.cenet_segments())
opt = optim.FusedSGD(arena, lr=0.01, momentum=0.9, weight_decay=1e-4)


def step():
    opt.zero_grad()
    crit(net(x), lab).backward()
    opt.step()


for _ in range(2):
    step()
torch.cuda.synchronize()

log = []
orig_gemm, orig_plain, orig_im = kern.gemm, kern.mat_plain, kern.mat_im2col


def plain(t, *a, **k):
    m = orig_plain(t, *a, **k)
    m._t = t
    return m


def im2col(t, *a, **k):
    m = orig_im(t, *a, **k)
    m._t = t
    return m


def wrapped(A, B, Cout, M, N, K, **kw):
    orig_gemm(A, B, Cout, M, N, K, **kw)
    nb, nkb = kw.get("nbatch", 1), kw.get("nkb", 1)
    key = (M, N, K, nb, nkb, "im" if B.mode else "pl", f"a{A.kfast}b{B.kfast}", "at" if kw.get("atomic") else
           ("c2i" if kw.get("col2im") else ("T" if kw.get("scc", 1) != 1 else "")), "R" if kw.get("R") is not None else "",
           "bias" if kw.get("bias") is not None else "")
    log.append((key, A, B, Cout, M, N, K, kw))


kern.gemm, kern.mat_plain, kern.mat_im2col = wrapped, plain, im2col
step()
torch.cuda.synchronize()
kern.gemm, kern.mat_plain, kern.mat_im2col = orig_gemm, orig_plain, orig_im

first = {}
count = collections.Counter()
for rec in log:
    count[rec[0]] += 1
    first.setdefault(rec[0], rec)


def timeit(rec, reps=20):
    _, A, B, Cout, M, N, K, kw = rec
    for _ in range(3):
        orig_gemm(A, B, Cout, M, N, K, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        orig_gemm(A, B, Cout, M, N, K, **kw)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us


def yardstick(key, reps=20):
    """the same logical product through torch.matmul (hipBLASLt / rocBLAS) on fresh operands of the same orientation: a YARDSTICK
    for how far the hand-written kernels are from a tuned library on this shape — never a product path"""
    M, N, K, nb, nkb = key[:5]
    if nkb != 1 or key[5] != "pl" or key[7] in ("at", "c2i"):
        return float("nan")
    akf, bkf = key[6][1] == "1", key[6][3] == "1"
    bs = (nb,) if nb > 1 else ()
    a = torch.randn(*bs, M, K, device=dev, dtype=torch.bfloat16) if akf else torch.randn(*bs, K, M, device=dev, dtype=torch.bfloat16).transpose(-1, -2)
    b = torch.randn(*bs, N, K, device=dev, dtype=torch.bfloat16).transpose(-1, -2) if bkf else torch.randn(*bs, K, N, device=dev, dtype=torch.bfloat16)
    out = torch.empty(*bs, M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        torch.matmul(a, b, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        torch.matmul(a, b, out=out)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


YARD = os.environ.get("GEMM_YARDSTICK") == "1"
rows = []
for key, rec in first.items():
    us = timeit(rec)
    rec[-1]["_yard"] = yardstick(key) if YARD else float("nan")
    M, N, K, nb, nkb = key[:5]
    flops = 2.0 * M * N * K * nb * nkb
    byt = nb * (2.0 * (M * K * nkb + K * N * nkb) + M * N * (8 if key[7] == "at" else 2))
    ideal = max(byt / 6e12, flops / 2.5e15) * 1e6
    rows.append((us * count[key], us, count[key], ideal, flops, byt, key, rec[-1]["_yard"]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"{len(log)} gemm launches, {len(rows)} distinct; isolated total {tot / 1e3:.2f} ms per step; ideal {sum(r[2] * r[3] for r in rows) / 1e3:.2f} ms")
if YARD:
    both = [r for r in rows if r[7] == r[7]]
    print(f"yardstick (torch.matmul) over the {len(both)} comparable shapes: ours {sum(r[1] * r[2] for r in both) / 1e3:.2f} ms, library {sum(r[7] * r[2] for r in both) / 1e3:.2f} ms")
print(f"{'tot us':>8s} {'n':>3s} {'us':>7s} {'lib us':>7s} {'ideal':>6s} {'TF':>6s} {'GB/s':>6s}  key")
for t, us, n, ideal, fl, byt, key, yd in rows[:top_n]:
    print(f"{t:8.0f} {n:3d} {us:7.1f} {yd:7.1f} {ideal:6.1f} {fl / us / 1e6:6.1f} {byt / us / 1e3:6.0f}  {key}")
