"""Census of every GEMM / implicit-GEMM launch of one training step (B=32, ACDC preset): shape, flags, measured time
(HIP events around each call) and a lower bound max(bytes / 6 TB/s, flops / peak).  Usage: python tools/gemm_census.py [bf16|f32]"""
import argparse
import collections
import os
import sys

sys.path.insert(0, os.getcwd())
import torch

import bench
from cenet_amd import kern, losses, optim

bf = (sys.argv[1] if len(sys.argv) > 1 else "bf16") == "bf16"
dev = torch.device("cuda:0")
kern.set_compute_bf16(bf)
net = bench.make_model(dev)
x, lab = bench.synthetic(32, dev, 0)
crit = losses.Criterion(4, argparse.Namespace(loss_type="dice,ce", loss_weights="0.5,0.5"))
arena = optim.ParamArena(net, optim.cenet_segments())
opt = optim.FusedSGD(arena, lr=0.01, momentum=0.9, weight_decay=1e-4)


def step():
    opt.zero_grad()
    loss = crit(net(x), lab)
    loss.backward()
    opt.step()


for _ in range(2):
    step()
torch.cuda.synchronize()

log = []
orig = kern.gemm


def wrapped(A, B, Cout, M, N, K, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    orig(A, B, Cout, M, N, K, **kw)
    e1.record()
    nb, nkb = kw.get("nbatch", 1), kw.get("nkb", 1)
    key = (M, N, K, nb, nkb, "im" if B.mode else "pl", f"a{A.kfast}b{B.kfast}", "at" if kw.get("atomic") else
           ("c2i" if kw.get("col2im") else ("T" if kw.get("scc", 1) != 1 else "")), kw.get("act", "none"),
           "R" if kw.get("R") is not None else "")
    log.append((key, e0, e1))


kern.gemm = wrapped
step()
torch.cuda.synchronize()
kern.gemm = orig
agg = collections.defaultdict(lambda: [0, 0.0])
for key, e0, e1 in log:
    agg[key][0] += 1
    agg[key][1] += e0.elapsed_time(e1)
esz = 2.0 if bf else 4.0
peak = 2.5e15 if bf else 1.57e14
rows = []
for key, (n, ms) in agg.items():
    M, N, K, nb, nkb = key[:5]
    flops = 2.0 * M * N * K * nb * nkb
    if key[5] == "im":
        byt = esz * (M * K * nkb + M * N * nb + N * K * nb * nkb / 9)  # rough: patches re-use the image ~k*k times
    else:
        byt = nb * (esz * (M * K * nkb + K * N * nkb) + M * N * (8 if key[7] == "at" else esz))
    ideal = max(byt / 6e12, flops / peak) * 1e3 + 0.004
    rows.append((ms - n * ideal, ms, n, ideal, flops, byt, key))
rows.sort(reverse=True)
tot = sum(r[1] for r in rows)
print(f"{len(log)} gemm launches, {tot:.2f} ms total (event-timed, includes launch gaps), ideal {sum(r[2] * r[3] for r in rows):.2f} ms")
print(f"{'excess':>7s} {'total':>7s} {'n':>3s} {'avg us':>8s} {'ideal us':>8s} {'TF':>6s} {'GB/s':>6s}  key")
for ex, ms, n, ideal, fl, byt, key in rows[:70]:
    print(f"{ex:7.2f} {ms:7.2f} {n:3d} {ms / n * 1e3:8.1f} {ideal * 1e3:8.1f} {fl * n / ms / 1e9:6.1f} {byt * n / ms / 1e6:6.0f}  {key}")
