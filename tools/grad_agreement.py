"""bf16-operand mode vs fp32 parity mode on the same batch (B=4, ACDC preset): loss and cosine similarity of the full
gradient vector, per arena segment."""
import argparse
import os
import sys

sys.path.insert(0, os.getcwd())
import torch

import bench
from cenet_amd import kern, losses, optim

dev = torch.device("cuda:0")
crit = losses.Criterion(4, argparse.Namespace(loss_type="dice,ce", loss_weights="0.5,0.5"))
x, lab = bench.synthetic(4, dev, 7)
res = {}
for mode in (False, True):
    kern.set_compute_bf16(mode)
    net = bench.make_model(dev)
    net.backbone.reset_drop_path(0.0)
    arena = optim.ParamArena(net, optim.cenet_segments())
    arena.zero_grad()
    loss = crit(net(x), lab)
    loss.backward()
    torch.cuda.synchronize()
    res[mode] = (loss.item(), arena.grads.clone(), arena)
kern.set_compute_bf16(False)
(l32, g32, arena), (l16, g16, _) = res[False], res[True]
print(f"loss fp32 {l32:.6f}  bf16 {l16:.6f}")
cos = torch.nn.functional.cosine_similarity(g32, g16, dim=0).item()
print(f"whole gradient: cosine {cos:.5f}  |g32| {g32.norm():.4f} |g16| {g16.norm():.4f}  rel L2 diff {(g32 - g16).norm() / g32.norm():.4f}")
for i, (name, s, e) in enumerate(arena.segments):
    a, b = g32[s:e], g16[s:e]
    print(f"  {name:14s} cosine {torch.nn.functional.cosine_similarity(a, b, dim=0).item():.5f}  rel L2 diff {(a - b).norm() / a.norm():.4f}")
