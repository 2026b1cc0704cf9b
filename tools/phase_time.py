"""Forward / backward / optimizer split of one training step (B=32, bf16 mode, weight-gradient overlap on)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import torch

import bench
from cenet_amd import kern, losses, ops, optim

dev = torch.device("cuda:0")
kern.set_compute_bf16(True)
ops.set_wgrad_overlap(True)
net = bench.make_model(dev)
x, lab = bench.synthetic(32, dev, 0)
crit = losses.Criterion(4, argparse.Namespace(loss_type="dice,ce", loss_weights="0.5,0.5"))
arena = optim.ParamArena(net, optim.cenet_segments())
opt = optim.FusedSGD(arena, lr=0.01, momentum=0.9, weight_decay=1e-4)


def sync():
    torch.cuda.synchronize()
    return time.perf_counter()


for _ in range(3):
    opt.zero_grad(); crit(net(x), lab).backward(); opt.step()
tf = tb = to = 0.0
N = 5
for _ in range(N):
    opt.zero_grad()
    t0 = sync()
    loss = crit(net(x), lab)
    t1 = sync()
    loss.backward()
    ops.wgrad_join()
    t2 = sync()
    opt.step()
    t3 = sync()
    tf += t1 - t0; tb += t2 - t1; to += t3 - t2
print(f"forward+loss {tf/N*1e3:.1f} ms   backward {tb/N*1e3:.1f} ms   sgd {to/N*1e3:.2f} ms")
with torch.no_grad():
    net.eval()
    for _ in range(2):
        net(x)
    t0 = sync()
    for _ in range(5):
        net(x)
    t1 = sync()
print(f"eval forward (no autograd) {(t1-t0)/5*1e3:.1f} ms = {32*5/(t1-t0):.0f} images/s")
