"""Where do the glue launches of one training step come from?  Every C-ABI call whose entry-point name is in GLUE is attributed to the
innermost cenet_amd frame above kern.py (file:line function), and torch's own elementwise adds (autograd's gradient accumulation at
fan-out points) to the autograd node that received them.   python tools/glue_census.py"""
import argparse
import collections
import os
import sys
import traceback

sys.path.insert(0, os.getcwd())
import torch

import bench
from cenet_amd import kern, losses, optim

GLUE = ("zero", "cast", "scale_batch", "transpose", "copy", "fill", "add")
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
orig = kern._call
agg = collections.Counter()


def wrapped(name, *args):
    if any(g in name for g in GLUE):
        fr = [f for f in traceback.extract_stack()[:-1] if "cenet_amd" in f.filename and not f.filename.endswith("kern.py")]
        chain = " < ".join(f"{os.path.basename(f.filename)}:{f.lineno} {f.name}" for f in fr[-3:][::-1])
        shp = "x".join(str(s) for a in args if isinstance(a, torch.Tensor) for s in [tuple(a.shape)][:1])
        agg[(name, chain, shp[:60])] += 1
    return orig(name, *args)


def pywrap(fn_name):
    f0 = getattr(kern, fn_name)

    def w(*a, **k):
        fr = [f for f in traceback.extract_stack()[:-1] if "cenet_amd" in f.filename and not f.filename.endswith("kern.py")]
        chain = " < ".join(f"{os.path.basename(f.filename)}:{f.lineno} {f.name}" for f in fr[-3:][::-1])
        t = a[0]
        agg[("py:" + fn_name, chain, str(tuple(t.shape)) + str(t.dtype)[6:])] += 1
        return f0(*a, **k)

    setattr(kern, fn_name, w)
    return f0


saved = {n: pywrap(n) for n in ("cast", "cast_into", "zero_")}
kern._call = wrapped
step()
for n, f0 in saved.items():
    setattr(kern, n, f0)
torch.cuda.synchronize()
kern._call = orig
for (name, chain, shp), n in sorted(agg.items(), key=lambda kv: (kv[0][0], -kv[1])):
    print(f"{n:3d}  {name:28s} {shp:40s} {chain}")

# torch's own adds: which autograd node's input buffer accumulated (two gradients met there)
from torch.profiler import ProfilerActivity, profile

with profile(activities=[ProfilerActivity.CPU], with_stack=False, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
seq = [e for e in prof.events() if e.name in ("aten::add", "aten::add_") or "Backward" in e.name]
seq.sort(key=lambda e: e.time_range.start)
adds = collections.Counter()
last = "?"
for i, e in enumerate(seq):
    if e.name in ("aten::add", "aten::add_"):
        nxt = next((f.name for f in seq[i + 1:] if "Backward" in f.name and not f.name.startswith("autograd::engine")), "?")
        adds[(e.name, str(e.input_shapes[:1]), last[:50], nxt[:50])] += 1
    elif not e.name.startswith("autograd::engine"):
        last = e.name
for k, n in sorted(adds.items(), key=lambda kv: -kv[1]):
    print(n, k)
