"""Which torch-native operators (aten::copy_, add, clone, ...) does one training step still launch, and from where?
torch.profiler with stacks, grouped by the innermost cenet_amd / bench frame.   python tools/torch_ops_census.py"""
import argparse
import collections
import os
import sys

sys.path.insert(0, os.getcwd())
import torch
from torch.profiler import ProfilerActivity, profile

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


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
agg = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith("aten::"):
        continue
    if ev.name in ("aten::empty", "aten::empty_like", "aten::view", "aten::as_strided", "aten::empty_strided", "aten::reshape",
                   "aten::detach", "aten::alias", "aten::_unsafe_view", "aten::transpose", "aten::permute", "aten::select",
                   "aten::slice", "aten::unsqueeze", "aten::expand", "aten::t", "aten::squeeze", "aten::narrow", "aten::flatten",
                   "aten::contiguous", "aten::to", "aten::_to_copy", "aten::clone", "aten::zeros", "aten::zeros_like", "aten::ones_like",
                   "aten::result_type", "aten::item", "aten::_local_scalar_dense", "aten::lift_fresh", "aten::resolve_conj",
                   "aten::resolve_neg", "aten::view_as", "aten::unbind", "aten::split", "aten::chunk", "aten::is_nonzero"):
        continue
    where = "?"
    for fr in ev.stack:
        if "cenet_amd" in fr or "bench.py" in fr or "tools/" in fr:
            where = fr.split("/root/repo/")[-1] if "/root/repo/" in fr else fr
            where = where[-90:]
            break
    agg[(ev.name, where)] += 1
for (name, where), n in agg.most_common(60):
    print(f"{n:5d}  {name:28s} {where}")
