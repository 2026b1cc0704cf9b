"""How long does the host take to ISSUE one training step (no GPU sync inside the timed loop)?  If this is close to the
measured step time the run is launch-bound."""
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
ops.set_wgrad_overlap(len(sys.argv) < 2 or sys.argv[1] != "nooverlap")
net = bench.make_model(dev)
x, lab = bench.synthetic(32, dev, 0)
crit = losses.Criterion(4, argparse.Namespace(loss_type="dice,ce", loss_weights="0.5,0.5"))
arena = optim.ParamArena(net, optim.cenet_segments())
opt = optim.FusedSGD(arena, lr=0.01, momentum=0.9, weight_decay=1e-4)


def step():
    opt.zero_grad()
    crit(net(x), lab).backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"issue {1e3 * (t1 - t0) / 3:.1f} ms/step; issue+drain {1e3 * (t2 - t0) / 3:.1f} ms/step")
