"""cProfile of the host side of two training steps (where does the launch overhead go?)."""
import argparse
import cProfile
import os
import pstats
import sys

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


def step():
    opt.zero_grad()
    crit(net(x), lab).backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(2):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
