"""Prints the loss of the first N eager training steps of the bench workload (debug aid)."""
import argparse, os, sys
sys.path.insert(0, os.getcwd())
import torch
import bench
from cenet_amd import kern, losses, optim
ap = argparse.ArgumentParser(); ap.add_argument("--dtype", default="f32"); ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--lr", type=float, default=0.01)
a = ap.parse_args()
dev = torch.device("cuda:0")
kern.set_compute_bf16(a.dtype == "bf16")
net = bench.make_model(dev)
arena = optim.ParamArena(net, optim.cenet_segments())
opt = optim.FusedSGD(arena, lr=a.lr)
crit = losses.Criterion(4, argparse.Namespace(loss_type="dice,ce", loss_weights="0.5,0.5"))
x, lab = bench.synthetic(32, dev, 1234)
for i in range(a.steps):
    opt.zero_grad(); loss = crit(net(x), lab); loss.backward(); opt.step()
    gn = arena.grads.norm().item()
    print(i, round(loss.item(), 5), "gradnorm", round(gn, 4), "param absmax", round(arena.params.abs().max().item(), 3), flush=True)
