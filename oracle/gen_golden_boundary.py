"""TEST INFRASTRUCTURE — golden vectors for BoundaryDoULoss (reference src/utils/core.py:83-131).

Imports the reference's own class in THIS container and runs it on CPU; the reference hard-codes `.cuda()` for two
temporaries (core.py:102,104), so `torch.Tensor.cuda` is patched to the identity for the duration of the call (the
arithmetic is unchanged).  Writes tests/golden/loss_boundary.npz: inputs, loss, input gradient for K = 4, 9, 2 with blocky
labels (regions with real boundaries, one case with an absent class) and for the 'boundary,ce' combination."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.gen_golden import OUT, load_reference_losses  # noqa: E402


def blocky_labels(B, K, size, g, absent=None):
    low = torch.rand(B, 1, size // 4, size // 4, generator=g)
    lab = torch.floor(torch.nn.functional.interpolate(low, size=(size, size), mode="nearest") * K).clamp_(0, K - 1)[:, 0]
    if absent is not None:
        lab[lab == absent] = 0
    return lab


def main():
    core = load_reference_losses()
    g = torch.Generator().manual_seed(4321)
    rec = {}
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        for K, absent, spec in ((4, None, ("boundary", "1")), (9, 7, ("boundary", "1")), (2, None, ("boundary,ce", "0.7,0.3"))):
            logits = torch.randn(3, K, 24, 24, generator=g).requires_grad_(True)
            labels = blocky_labels(3, K, 24, g, absent)
            crit = core.Criterion(K, argparse.Namespace(loss_type=spec[0], loss_weights=spec[1]))
            loss = crit(logits, labels)
            loss.backward()
            rec[f"K{K}.logits"] = logits.detach().numpy()
            rec[f"K{K}.labels"] = labels.numpy()
            rec[f"K{K}.loss"] = np.float64(loss.item())
            rec[f"K{K}.grad"] = logits.grad.numpy()
            rec[f"K{K}.spec"] = np.array(spec)
    finally:
        torch.Tensor.cuda = orig_cuda
    np.savez_compressed(os.path.join(OUT, "loss_boundary.npz"), **rec)
    print("[golden] loss_boundary", {k: float(v) for k, v in rec.items() if k.endswith(".loss")})


if __name__ == "__main__":
    main()
