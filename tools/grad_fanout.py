"""Which forward tensors have more than one consumer in the autograd graph of one training step?  Each extra consumer costs
one torch add kernel in the backward pass (the engine sums the incoming gradients).  Walks loss.grad_fn and counts the edges
into every (node, output index); prints them grouped by producing node and the consumers' names.
python tools/grad_fanout.py"""
import argparse
import collections
import os
import sys

sys.path.insert(0, os.getcwd())
import torch

import bench
from cenet_amd import kern, losses

dev = torch.device("cuda:0")
kern.set_compute_bf16(True)
net = bench.make_model(dev)
x, lab = bench.synthetic(32, dev, 0)
crit = losses.Criterion(4, argparse.Namespace(loss_type="dice,ce", loss_weights="0.5,0.5"))
loss = crit(net(x), lab)
edges = collections.defaultdict(list)  # (producer node, output nr) -> consumer names
seen, stack = set(), [loss.grad_fn]
while stack:
    n = stack.pop()
    if n is None or n in seen:
        continue
    seen.add(n)
    for nxt, nr in n.next_functions:
        if nxt is not None:
            edges[(nxt, nr)].append(type(n).__name__)
            stack.append(nxt)
agg = collections.Counter()
for (prod, nr), cons in edges.items():
    if len(cons) > 1 and type(prod).__name__ != "AccumulateGrad":
        agg[(type(prod).__name__, nr, tuple(sorted(cons)))] += 1
print(f"{sum(len(c) - 1 for (p, _), c in edges.items() if len(c) > 1 and type(p).__name__ != 'AccumulateGrad')} gradient adds per step")
for (p, nr, cons), n in agg.most_common():
    print(f"{n:4d}x  {p}[{nr}] -> {', '.join(cons)}")
