"""Which part of the network puts the noise into the bf16-mode gradients?  On a golden configuration, runs the all-fp32 step,
the all-bf16 step, and hybrids where the named submodules run on fp32 tensors (casts at their boundary), and prints the
cosine of every arena segment of the gradient against the fp32 run.  python tools/bf16_bisect.py [config]"""
import argparse
import os
import sys

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
from backend import use_hip
from cenet_amd import kern, losses, optim
from test_model_parity import build

name = sys.argv[1] if len(sys.argv) > 1 else "synapse"
d = use_hip()


def cast_tree(t, dt):
    if isinstance(t, torch.Tensor):
        return t.to(dt) if t.is_floating_point() else t
    if isinstance(t, (list, tuple)):
        return type(t)(cast_tree(u, dt) for u in t)
    return t


def run(bf16, fp32_modules=(), round_input=False):
    net, cfg, z, x, lab = build(name, d)
    if round_input:  # the ONLY perturbation: the network input rounded to bf16 values (relative 2^-9), everything else fp32
        x = x.bfloat16().float()
    net.train()
    net.backbone.reset_drop_path(0.0)
    arena = optim.ParamArena(net, optim.cenet_segments())
    crit = losses.Criterion(cfg.num_classes, argparse.Namespace(loss_type="dice,ce", loss_weights="0.5,0.5"))
    mods = dict(net.named_modules())
    for mn in fp32_modules:
        m = mods[mn]
        orig = m.forward
        m.forward = (lambda o: lambda *a, **k: cast_tree(o(*cast_tree(a, torch.float32), **k), torch.bfloat16))(orig)
    kern.set_compute_bf16(bf16)
    loss = crit(net(x), lab)
    loss.backward()
    kern.set_compute_bf16(False)
    torch.cuda.synchronize()
    return loss.item(), arena.grads.clone(), arena


l32, g32, arena = run(False)
cs = torch.nn.functional.cosine_similarity


def report(label, res):
    l, g, _ = res
    segs = " ".join(f"{n}:{cs(g32[s:e], g[s:e], dim=0).item():.4f}" for n, s, e in arena.segments)
    print(f"{label:44s} loss {l:.5f}  whole {cs(g32, g, dim=0).item():.4f}  {segs}", flush=True)


print(f"{name}: fp32 loss {l32:.5f}")
report("all bf16", run(True))
report("all fp32, input rounded to bf16 values", run(False, round_input=True))
if "--quick" in sys.argv:
    sys.exit(0)
report("fp32: decoder + out", run(True, ["decoder", "out"]))
report("fp32: backbone", run(True, ["backbone"]))
for mn in ("out", "decoder.dec1", "decoder.skip_enhancer1", "decoder.up1", "decoder.dec2", "decoder.skip_enhancer2",
           "decoder.dec3", "decoder.skip_enhancer3", "decoder.dec4"):
    report("fp32: " + mn, run(True, [mn]))
