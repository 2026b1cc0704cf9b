"""Which parameter gradients differ between eager steps and hipGraph replays from identical state (debug aid)."""
import argparse, copy, os, sys
sys.path.insert(0, os.getcwd())
import torch
import bench
from cenet_amd import kern, losses, optim
from cenet_amd.graph import GraphedStep
dev = torch.device("cuda:0")
crit = losses.Criterion(4, argparse.Namespace(loss_type="dice,ce", loss_weights="0.5,0.5"))
x, lab = bench.synthetic(8, dev, 1234)
def mk():
    net = bench.make_model(dev); net.backbone.reset_drop_path(0.0)
    arena = optim.ParamArena(net, optim.cenet_segments()); opt = optim.FusedSGD(arena, lr=0.0)  # lr 0: state never changes
    return net, arena, opt
netA, arA, optA = mk(); netB, arB, optB = mk()
netB.load_state_dict(netA.state_dict())
def bodyA():
    optA.zero_grad(); l = crit(netA(x), lab); l.backward(); optA.step(); return l
def bodyB():
    optB.zero_grad(); l = crit(netB(x), lab); l.backward(); optB.step(sync_hyper=False); return l
g = GraphedStep(bodyB, optimizer=optB, warmup=2)
for it in range(3):
    la = bodyA(); lb = g(); torch.cuda.synchronize()
    d = (arA.grads - arB.grads).abs()
    print("iter", it, "loss", la.item(), lb.item(), "max grad diff", d.max().item(), "gradnorm A/B", arA.grads.norm().item(), arB.grads.norm().item())
    if d.max().item() > 1e-3:
        worst = []
        for n, (o, cnt) in arB.index.items():
            dd = d[o:o+cnt].max().item()
            if dd > 1e-3: worst.append((dd, n))
        worst.sort(reverse=True)
        print(" differing params:", len(worst)); [print("   ", round(w, 4), n) for w, n in worst[:25]]
        break
