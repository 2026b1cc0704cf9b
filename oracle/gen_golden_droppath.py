"""TEST INFRASTRUCTURE — golden vectors for the stochastic-depth path of the PVT blocks (reference pvtv2.py:145-149,
SURVEY.md §8 a5) with INJECTED per-sample keep masks.

Runs the unmodified reference CENet (ACDC preset, default drop_path_rate 0.1) in train mode in THIS container.  The only
randomness of that forward is `Tensor.bernoulli_` inside DropPath (timm semantics, oracle/ref_shim); it is patched for the
duration of the pass to hand out prepared 0/1 masks in call order — (stage, block): attention branch, then MLP branch;
block (0, 0) has rate 0 and is an Identity (pvtv2.py:123).  Recorded: the masks, train-mode logits, Dice+CE loss and the
probe gradients in fp32 and fp64 (same probes as model_*.npz).  Writes tests/golden/model_acdc_droppath.npz."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import cenet_oracle as O  # noqa: E402
from oracle.gen_golden import OUT, load_reference_losses, synthetic_batch  # noqa: E402  (puts the reference on sys.path)
from oracle.gen_golden_keys import PROBE_KEYS  # noqa: E402
from oracle.golden_cases import MODEL_CONFIGS  # noqa: E402

DEPTHS = (3, 4, 6, 3)


def make_masks(B, g):
    """0/1 keep masks with ~30 % drops (any 0/1 pattern is a legal draw; the scale 1/keep_prob comes from the module)"""
    masks = {}
    for s, d in enumerate(DEPTHS):
        for i in range(d):
            if s == 0 and i == 0:
                continue
            masks[(s, i)] = tuple((torch.rand(B, generator=g) > 0.3).float() for _ in range(2))
    return masks


class _Inject:
    def __init__(self, masks):
        self.queue = [m for key in sorted(masks) for m in masks[key]]
        self.pos = 0

    def __enter__(self):
        self.orig = torch.Tensor.bernoulli_
        inj = self

        def fake(t, p=0.5, *, generator=None):
            m = inj.queue[inj.pos]
            inj.pos += 1
            return t.copy_(m.to(t.dtype).view(t.shape))
        torch.Tensor.bernoulli_ = fake
        return self

    def __exit__(self, *exc):
        torch.Tensor.bernoulli_ = self.orig


def run(net, x, lab, crit, masks, double=False):
    net.train()
    if double:
        net, x, lab = net.double(), x.double(), lab.double()
    with _Inject(masks) as inj:
        lt = net(x)
        assert inj.pos == len(inj.queue), (inj.pos, len(inj.queue))
    loss = crit(lt, lab)
    loss.backward()
    return lt.detach(), loss


def main():
    from networks import CENet
    core = load_reference_losses()
    mc = MODEL_CONFIGS["acdc"]
    kw, B = mc["kw"], mc["batch"]
    K = kw["num_classes"]
    x, lab = synthetic_batch(B, kw["input_channels"], K)
    masks = make_masks(B, torch.Generator().manual_seed(2024))
    crit = core.Criterion(K, argparse.Namespace(loss_type="dice,ce", loss_weights="0.5,0.5"))
    rec = {"x_seed": np.int64(1234), "fill_seed": np.int64(42)}
    for (s, i), (ma, mm) in masks.items():
        rec[f"mask.{s}.{i}"] = torch.stack([ma, mm]).numpy()
    out = {}
    for tag, dbl in (("", False), ("64", True)):
        torch.manual_seed(5)
        net = CENet(**kw)
        sd = net.state_dict()
        O.fill_state_dict_(sd, seed=42)
        lt, loss = run(net, x, lab, crit, masks, double=dbl)
        params = dict(net.named_parameters())
        rec["loss" + tag] = np.float64(loss.item())
        if not dbl:
            rec["logits_train_sub"] = lt[:, :, ::9, ::9].float().numpy()
        for k in PROBE_KEYS:
            gk = params[k].grad.reshape(-1)
            rec[f"g{tag}." + k + ".norm"] = np.float64(gk.double().norm().item())
            rec[f"g{tag}." + k + ".head"] = gk[:16].numpy().copy()
        out[tag] = loss.item()
    np.savez_compressed(os.path.join(OUT, "model_acdc_droppath.npz"), **rec)
    print("[golden] model_acdc_droppath: loss", out)


if __name__ == "__main__":
    main()
