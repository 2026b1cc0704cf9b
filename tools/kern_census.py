"""Census of the non-GEMM kernel launches of one training step (B=32): per C-ABI entry point, calls, total time (HIP
events) and the bytes of the tensor arguments (a crude traffic estimate) -> effective GB/s."""
import argparse
import collections
import os
import sys

sys.path.insert(0, os.getcwd())
import torch

import bench
from cenet_amd import kern, losses, optim

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
orig = kern._call


def wrapped(name, *args):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    byt = sum(a.numel() * a.element_size() for a in args if isinstance(a, torch.Tensor))
    e0.record()
    orig(name, *args)
    e1.record()
    log.append((name, byt, e0, e1))


kern._call = wrapped
step()
torch.cuda.synchronize()
kern._call = orig
agg = collections.defaultdict(lambda: [0, 0.0, 0])
for name, byt, e0, e1 in log:
    a = agg[name]
    a[0] += 1
    a[1] += e0.elapsed_time(e1)
    a[2] += byt
print(f"{'entry point':44s} {'calls':>5s} {'ms':>7s} {'MB':>8s} {'GB/s':>7s}")
for name, (n, ms, byt) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{name:44s} {n:5d} {ms:7.2f} {byt / 1e6:8.0f} {byt / ms / 1e6:7.0f}")
