"""Loss per replay of the hipGraph-captured step vs eager (debug aid)."""
import argparse, os, sys
sys.path.insert(0, os.getcwd())
import torch
import bench
from cenet_amd import kern, losses, optim
from cenet_amd.graph import GraphedStep
dev = torch.device("cuda:0")
net = bench.make_model(dev)
if len(sys.argv) > 1 and sys.argv[1] == "nodrop":
    net.backbone.reset_drop_path(0.0)
arena = optim.ParamArena(net, optim.cenet_segments())
opt = optim.FusedSGD(arena, lr=0.01)
crit = losses.Criterion(4, argparse.Namespace(loss_type="dice,ce", loss_weights="0.5,0.5"))
x, lab = bench.synthetic(32, dev, 1234)
def body():
    opt.zero_grad(); loss = crit(net(x), lab); loss.backward(); opt.step(sync_hyper=False); return loss
g = GraphedStep(body, optimizer=opt, warmup=2)
for i in range(8):
    l = g()
    torch.cuda.synchronize()
    print(i, l.item(), "gradnorm", arena.grads.norm().item(), "params finite", torch.isfinite(arena.params).all().item(), flush=True)
