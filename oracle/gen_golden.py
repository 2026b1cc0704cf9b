"""Golden-vector generator (TEST INFRASTRUCTURE; runs ONLY in the build container).

Imports the unmodified reference from /root/reference/src through the timm/monai shim in
`oracle/ref_shim/`, runs it on seeded inputs/weights and writes small `.npz`
fixtures under `tests/golden/`.  The fixtures are data (inputs, weights, outputs, gradients, BN buffers);
no reference source is stored.  Re-run:  python -m oracle.gen_golden
"""
from __future__ import annotations

import argparse
import importlib
import importlib.util
import os
import sys
import types
from functools import partial

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF_SRC = "/root/reference/src"
sys.path[:0] = [os.path.join(HERE, "ref_shim"), REF_SRC]

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

sys.path.insert(0, ROOT)
from oracle import cenet_oracle as O  # noqa: E402
from oracle.golden_cases import CASES, MODEL_CONFIGS, NONFINITE_CASE  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

from oracle.gen_golden_keys import PROBE_BUFFERS, PROBE_KEYS  # noqa: E402


def load_reference_losses():
    """utils/core.py pulls thop/fvcore through utils/__init__ — load it by path with a stub `utils.utils.flatten`."""
    pkg = types.ModuleType("utils")
    pkg.__path__ = [os.path.join(REF_SRC, "utils")]
    sub = types.ModuleType("utils.utils")
    sub.flatten = lambda t: t.reshape(-1)
    sys.modules["utils"], sys.modules["utils.utils"] = pkg, sub
    spec = importlib.util.spec_from_file_location("utils.core", os.path.join(REF_SRC, "utils", "core.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["utils.core"] = mod
    spec.loader.exec_module(mod)
    return mod


def synthetic_batch(B, cin, K, seed=1234, size=224):
    """Seeded images + blocky integer labels (float dtype, like the reference's loaders)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, cin, size, size, generator=g)
    low = torch.rand(B, 1, size // 16, size // 16, generator=g)
    lab = torch.floor(torch.nn.functional.interpolate(low, size=(size, size), mode="nearest") * K).clamp_(0, K - 1)
    return x, lab[:, 0]


def gen_module_cases(only_case=None):
    for idx, c in enumerate(CASES):
        if only_case and c["name"] not in only_case:
            continue
        modname, clsname, kw = c["ref"]
        kw = dict(kw)
        if kw.get("norm_layer") == "LN6":
            kw["norm_layer"] = partial(nn.LayerNorm, eps=1e-6)
        cls = getattr(importlib.import_module(modname), clsname)
        torch.manual_seed(100 + idx)
        m = cls(**kw)
        O.fill_state_dict_(m.state_dict(), seed=1000 + idx)
        g = torch.Generator().manual_seed(2000 + idx)
        ins = [torch.randn(s, generator=g).requires_grad_(True) for s in c["inputs"]]
        call = c["call"] or (lambda mod, xs: mod(xs[0]))
        rec = {}
        for k, v in m.state_dict().items():
            rec["sd." + k] = v.detach().clone().numpy()
        for i, t in enumerate(ins):
            rec[f"in{i}"] = t.detach().numpy()
        m.eval()
        with torch.no_grad():
            rec["out_eval"] = call(m, [t.detach() for t in ins]).numpy()
        m.train()
        out = call(m, ins)
        cot = torch.randn(out.shape, generator=g)
        (out * cot).sum().backward()
        rec["out"] = out.detach().numpy()
        rec["cot"] = cot.numpy()
        for i, t in enumerate(ins):
            rec[f"gin{i}"] = t.grad.numpy()
        for k, p in m.named_parameters():
            if p.grad is not None:
                rec["gsd." + k] = p.grad.numpy()
        for k, b in m.named_buffers():
            rec["after." + k] = b.detach().numpy()
        np.savez_compressed(os.path.join(OUT, f"mod_{c['name']}.npz"), **rec)
        print(f"[golden] {c['name']}: out {tuple(out.shape)}  {sum(v.nbytes for v in rec.values()) / 1024:.0f} KiB raw")


def gen_nonfinite_case():
    """multihead_diffattn.py:106 (nan_to_num of the scores): eval forward of the reference on inputs whose scores overflow"""
    c = NONFINITE_CASE
    modname, clsname, kw = c["ref"]
    cls = getattr(importlib.import_module(modname), clsname)
    torch.manual_seed(c["seed"])
    m = cls(**kw)
    O.fill_state_dict_(m.state_dict(), seed=c["seed"] + 1)
    with torch.no_grad():
        m.q_proj.weight.mul_(c["wscale"])
        m.k_proj.weight.mul_(c["wscale"])
    g = torch.Generator().manual_seed(c["seed"] + 2)
    x = torch.randn(c["inputs"][0], generator=g)
    m.eval()
    with torch.no_grad():
        q = torch.nn.functional.linear(x, m.q_proj.weight)
        k = torch.nn.functional.linear(x, m.k_proj.weight)
        assert not torch.isfinite(q @ k.transpose(-1, -2)).all(), "the case is meant to overflow the scores"
        out = m(x)
    assert torch.isfinite(out).all(), "the reference's forward is finite on this input"
    rec = {"sd." + k_: v.detach().clone().numpy() for k_, v in m.state_dict().items()}
    rec["in0"] = x.numpy()
    rec["out_eval"] = out.numpy()
    np.savez_compressed(os.path.join(OUT, f"mod_{c['name']}.npz"), **rec)
    print(f"[golden] {c['name']}: finite output, max |out| {out.abs().max().item():.3g}")


def gen_loss_cases(core):
    g = torch.Generator().manual_seed(77)
    rec = {}
    for K in (4, 9, 2):
        logits = torch.randn(2, K, 24, 24, generator=g).requires_grad_(True)
        labels = torch.randint(0, K, (2, 24, 24), generator=g).float()
        args = argparse.Namespace(loss_type="dice,ce", loss_weights="0.5,0.5")
        crit = core.Criterion(K, args)
        loss = crit(logits, labels)
        loss.backward()
        dl = core.DiceLoss(K)(logits.detach(), labels, softmax=True)
        rec[f"K{K}.logits"] = logits.detach().numpy()
        rec[f"K{K}.labels"] = labels.numpy()
        rec[f"K{K}.loss"] = np.float64(loss.item())
        rec[f"K{K}.dice_loss"] = np.float64(dl.item())
        rec[f"K{K}.grad"] = logits.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "loss_dice_ce.npz"), **rec)
    print("[golden] loss_dice_ce")


def gen_model_cases(core):
    from networks import CENet
    for name, mc in MODEL_CONFIGS.items():
        kw, B = mc["kw"], mc["batch"]
        K = kw["num_classes"]
        torch.manual_seed(5)
        net = CENet(**kw)
        sd = net.state_dict()
        O.fill_state_dict_(sd, seed=42)
        x, lab = synthetic_batch(B, kw["input_channels"], K)
        rec = {"x_seed": np.int64(1234), "fill_seed": np.int64(42)}
        net.eval()
        with torch.no_grad():
            le = net(x)
        rec["logits_eval_sub"] = le[:, :, ::9, ::9].numpy()
        rec["logits_eval_sum"] = np.float64(le.double().sum().item())
        rec["logits_eval_abs"] = np.float64(le.double().abs().sum().item())
        rec["pred_eval_sub"] = torch.argmax(torch.softmax(le, 1), 1)[:, ::5, ::5].numpy().astype(np.int8)
        rec["dice_eval"] = np.float64(O.mean_class_dice(le, lab, K))
        # train mode, stochastic depth off (reset_drop_path(0.), pvtv2.py:272)
        net.train()
        net.backbone.reset_drop_path(0.0)
        crit = core.Criterion(K, argparse.Namespace(loss_type="dice,ce", loss_weights="0.5,0.5"))
        opt = torch.optim.SGD(net.parameters(), lr=0.01, weight_decay=1e-4, momentum=0.9)
        opt.zero_grad()
        lt = net(x)
        loss = crit(lt, lab)
        loss.backward()
        rec["logits_train_sub"] = lt.detach()[:, :, ::9, ::9].numpy()
        rec["logits_train_sum"] = np.float64(lt.detach().double().sum().item())
        rec["loss"] = np.float64(loss.item())
        params = dict(net.named_parameters())
        for k in PROBE_KEYS:
            gk = params[k].grad.reshape(-1)
            rec["g." + k + ".norm"] = np.float64(gk.double().norm().item())
            rec["g." + k + ".head"] = gk[:16].numpy().copy()
        bufs = dict(net.named_buffers())
        for k in PROBE_BUFFERS:
            rec["b." + k] = bufs[k].detach().reshape(-1)[:16].numpy().copy()
        opt.step()
        for k in PROBE_KEYS:
            pk = params[k].detach().reshape(-1)
            rec["p1." + k + ".head"] = pk[:16].numpy().copy()
        # second step exercises the momentum buffer
        opt.zero_grad()
        loss2 = crit(net(x), lab)
        loss2.backward()
        opt.step()
        rec["loss2"] = np.float64(loss2.item())
        for k in PROBE_KEYS:
            rec["p2." + k + ".head"] = params[k].detach().reshape(-1)[:16].numpy().copy()
        # the same first training pass evaluated by the reference in float64: lets the tests bound the product's error
        # by the reference's OWN fp32 rounding error on cancellation-dominated gradients (e.g. conv weights feeding a BN)
        torch.manual_seed(5)
        net64 = CENet(**kw)
        sd64 = net64.state_dict()
        O.fill_state_dict_(sd64, seed=42)
        net64 = net64.double()
        net64.train()
        net64.backbone.reset_drop_path(0.0)
        lt64 = net64(x.double())
        loss64 = crit(lt64, lab.double())
        loss64.backward()
        rec["loss64"] = np.float64(loss64.item())
        p64 = dict(net64.named_parameters())
        for k in PROBE_KEYS:
            gk = p64[k].grad.reshape(-1)
            rec["g64." + k + ".norm"] = np.float64(gk.norm().item())
            rec["g64." + k + ".head"] = gk[:16].numpy().copy()
        np.savez_compressed(os.path.join(OUT, f"model_{name}.npz"), **rec)
        print(f"[golden] model_{name}: loss {loss.item():.6f} -> {loss2.item():.6f}  dice_eval {rec['dice_eval']:.4f}")


def main():
    os.makedirs(OUT, exist_ok=True)
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="all", choices=["all", "modules", "loss", "models", "nonfinite"])
    ap.add_argument("--case", action="append", help="module cases to (re)generate (default: all)")
    a = ap.parse_args()
    torch.set_num_threads(8)
    core = load_reference_losses()
    if a.only in ("all", "modules"):
        gen_module_cases(a.case)
    if a.only in ("all", "nonfinite"):
        gen_nonfinite_case()
    if a.only in ("all", "loss"):
        gen_loss_cases(core)
    if a.only in ("all", "models"):
        gen_model_cases(core)


if __name__ == "__main__":
    main()
