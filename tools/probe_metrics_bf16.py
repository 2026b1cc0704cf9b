"""bf16 (benched) mode vs the fp32 parity mode and vs the reference's float64 goldens on the three whole-model golden
configurations: loss, train logits, per-arena-segment cosine of the FULL gradient vector against the fp32 product run,
per-probe-tensor norm ratio / head cosine against g64.  python tools/probe_metrics_bf16.py"""
import argparse
import os
import sys

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import torch
from backend import use_hip
from cenet_amd import kern, losses, optim
from oracle.gen_golden_keys import PROBE_KEYS
from test_model_parity import build

for name in ("acdc", "synapse", "skin"):
    d = use_hip()
    res = {}
    for mode in (False, True):
        net, cfg, z, x, lab = build(name, d)
        net.train()
        net.backbone.reset_drop_path(0.0)
        arena = optim.ParamArena(net, optim.cenet_segments())
        crit = losses.Criterion(cfg.num_classes, argparse.Namespace(loss_type="dice,ce", loss_weights="0.5,0.5"))
        kern.set_compute_bf16(mode)
        lt = net(x)
        loss = crit(lt, lab)
        loss.backward()
        kern.set_compute_bf16(False)
        torch.cuda.synchronize()
        res[mode] = (loss.item(), arena.grads.clone(), arena, dict(net.named_parameters()), lt.detach().float().cpu())
    (l32, g32, arena, _, _), (l16, g16, _, params, lt) = res[False], res[True]
    print(f"{name}: loss golden {float(z['loss']):.6f} fp32 {l32:.6f} bf16 {l16:.6f}")
    ref = z["logits_train_sub"]
    print("  logits mean|d|/range", np.abs(lt[:, :, ::9, ::9].numpy() - ref).mean() / np.abs(ref).max())
    cs = torch.nn.functional.cosine_similarity
    print(f"  whole gradient vs fp32 product: cosine {cs(g32, g16, dim=0).item():.5f} rel L2 {((g32 - g16).norm() / g32.norm()).item():.4f}")
    for seg, s, e in arena.segments:
        print(f"    {seg:14s} cosine {cs(g32[s:e], g16[s:e], dim=0).item():.5f}")
    for k in PROBE_KEYS:
        g = params[k].grad.reshape(-1).double().cpu()
        n64, n32 = float(z["g64." + k + ".norm"]), float(z["g." + k + ".norm"])
        h, h64 = g[:16].numpy(), z["g64." + k + ".head"]
        c = float(h @ h64 / (np.linalg.norm(h) * np.linalg.norm(h64) + 1e-300))
        print(f"  {k:56s} n {g.numel():7d} norm/g64 {g.norm().item() / n64:7.4f} (ref fp32 {n32 / n64:6.4f}) head cos {c:8.5f}")
