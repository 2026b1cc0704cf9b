"""Finite-ness / consistency of one training step at unusual batch sizes (both precision modes)."""
import argparse
import os
import sys

sys.path.insert(0, os.getcwd())
import torch

import bench
from cenet_amd import kern, losses, ops, optim

dev = torch.device("cuda:0")
crit = losses.Criterion(4, argparse.Namespace(loss_type="dice,ce,boundary", loss_weights="0.4,0.3,0.3"))
ops.set_wgrad_overlap(True)
for bf in (False, True):
    kern.set_compute_bf16(bf)
    for B in (1, 3, 7, 64):
        net = bench.make_model(dev)
        arena = optim.ParamArena(net, optim.cenet_segments())
        opt = optim.FusedSGD(arena, lr=0.01, momentum=0.9, weight_decay=1e-4)
        x, lab = bench.synthetic(B, dev, B)
        for _ in range(2):
            opt.zero_grad()
            loss = crit(net(x), lab)
            loss.backward()
            opt.step()
        torch.cuda.synchronize()
        ok = bool(torch.isfinite(arena.grads).all() and torch.isfinite(arena.params).all())
        print(f"{'bf16' if bf else 'f32 '} B={B:3d} loss {loss.item():.5f} grad finite/params finite: {ok}  |g| {arena.grads.norm().item():.4f}")
kern.set_compute_bf16(False)
