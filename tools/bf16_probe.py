import sys, os, time, argparse
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np, torch
from test_model_parity import build
from backend import use_hip
from cenet_amd import kern, losses
from oracle import cenet_oracle as O
dev = use_hip()
net, cfg, z, x, lab = build("acdc", dev)
net.eval()
for mode in (False, True):
    kern.set_compute_bf16(mode)
    with torch.no_grad():
        le = net(x).cpu()
    ref = z["logits_eval_sub"]
    d = np.abs(le[:, :, ::9, ::9].numpy() - ref)
    pred = O.predict(le)[:, ::5, ::5].numpy()
    print("bf16" if mode else "fp32", "max|dlogit|", d.max(), "mean", d.mean(), "max|logit|", np.abs(ref).max(),
          "mask mismatch", (pred != z["pred_eval_sub"]).mean(), "dice", O.mean_class_dice(le, lab.cpu(), 4), float(z["dice_eval"]))
kern.set_compute_bf16(False)
