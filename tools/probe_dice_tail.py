"""Where do the bf16 mode's borderline argmax flips come from?  For the three whole-model goldens (eval): Dice of the bf16 product
against the reference's, (a) as shipped (bf16 logits), (b) with the head's last 1x1 conv + bilinear x2 redone in fp32 from the bf16
input of that conv (what an fp32 tail would give), (c) the fp32 product.  python tools/probe_dice_tail.py"""
import os
import sys

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import torch
import torch.nn.functional as F

from backend import use_hip
from cenet_amd import kern
from oracle import cenet_oracle as O
from test_model_parity import build

d = use_hip()
for preset in ("acdc", "synapse", "skin"):
    net, cfg, z, x, lab = build(preset, d)
    net.eval()
    K = cfg.num_classes
    ref = float(z["dice_eval"])
    with torch.no_grad():
        l32 = net(x).float().cpu()
    cap = {}
    h = net.out.out[1].register_forward_hook(lambda m, i, o: cap.__setitem__("in", i[0].detach()))
    kern.set_compute_bf16(True)
    try:
        with torch.no_grad():
            lb = net(x).float().cpu()
    finally:
        kern.set_compute_bf16(False)
        h.remove()
    conv = net.out.out[1].conv.conv
    with torch.no_grad():
        y = F.conv2d(cap["in"].float(), conv.weight.float(), conv.bias.float())
        lt = F.interpolate(y, scale_factor=2, mode="bilinear", align_corners=False).cpu()
    dd = lambda l: abs(O.mean_class_dice(l, lab.cpu(), K) - ref)
    fl = lambda l: float((O.predict(l) != O.predict(l32)).float().mean())
    print(f"{preset:8s} dDice: fp32 product {dd(l32):.2e} | bf16 as shipped {dd(lb):.2e} (flips vs fp32 {fl(lb):.2e}) | "
          f"bf16 body + fp32 tail {dd(lt):.2e} (flips {fl(lt):.2e})", flush=True)
