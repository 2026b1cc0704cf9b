"""HIP loss kernels (Dice + CE + BoundaryDoU as terms of one fused pass, loss_optim.hip) against the golden vectors the
reference's own classes produced (tests/golden/loss_dice_ce.npz, loss_boundary.npz) through the reference-shaped interface
`Criterion(num_classes, args)(outputs, labels)` (core.py:161-188), and against the oracle on larger random cases."""
import argparse
import os

import numpy as np
import pytest
import torch

from backend import dev  # noqa: F401
from cenet_amd import losses
from oracle import cenet_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _run(dev, K, spec, logits, labels):
    crit = losses.Criterion(K, argparse.Namespace(loss_type=spec[0], loss_weights=spec[1]))
    lg = torch.from_numpy(logits).to(dev).requires_grad_(True)
    loss = crit(lg, torch.from_numpy(labels).to(dev))
    loss.backward()
    return float(loss.item()), lg.grad.cpu().numpy()


@pytest.mark.parametrize("K", [4, 9, 2])
def test_dice_ce_golden(dev, K):
    z = np.load(os.path.join(GOLDEN, "loss_dice_ce.npz"))
    loss, grad = _run(dev, K, ("dice,ce", "0.5,0.5"), z[f"K{K}.logits"], z[f"K{K}.labels"])
    assert abs(loss - float(z[f"K{K}.loss"])) < 1e-5
    np.testing.assert_allclose(grad, z[f"K{K}.grad"], rtol=1e-3, atol=1e-8)


@pytest.mark.parametrize("K", [4, 9, 2])  # 9: one class absent from the labels; 2: 'boundary,ce' with weights 0.7 / 0.3
def test_boundary_golden(dev, K):
    z = np.load(os.path.join(GOLDEN, "loss_boundary.npz"))
    spec = (str(z[f"K{K}.spec"][0]), str(z[f"K{K}.spec"][1]))
    loss, grad = _run(dev, K, spec, z[f"K{K}.logits"], z[f"K{K}.labels"])
    assert abs(loss - float(z[f"K{K}.loss"])) < 1e-5
    np.testing.assert_allclose(grad, z[f"K{K}.grad"], rtol=1e-3, atol=1e-8)


def test_all_three_terms_vs_oracle(dev):
    """dice + ce + boundary together on a ragged image size (H != W, neither a multiple of the workgroup chunk)."""
    g = torch.Generator().manual_seed(5)
    K, B, H, W = 5, 3, 37, 50
    logits = torch.randn(B, K, H, W, generator=g) * 2
    low = torch.rand(B, 1, 5, 7, generator=g)
    labels = torch.floor(torch.nn.functional.interpolate(low, size=(H, W), mode="nearest") * K).clamp_(0, K - 1)[:, 0]
    lr = logits.clone().requires_grad_(True)
    ref = O.criterion(lr, labels, K, loss_type=("dice", "ce", "boundary"), weights=(0.3, 0.2, 0.5))
    ref.backward()
    loss, grad = _run(dev, K, ("dice,ce,boundary", "0.3,0.2,0.5"), logits.numpy(), labels.numpy())
    assert abs(loss - ref.item()) < 1e-5
    np.testing.assert_allclose(grad, lr.grad.numpy(), rtol=1e-3, atol=1e-8)


def test_boundary_class_module(dev):
    g = torch.Generator().manual_seed(6)
    logits = torch.randn(2, 3, 16, 16, generator=g)
    labels = torch.randint(0, 3, (2, 16, 16), generator=g).float()
    out = losses.BoundaryDoULoss(3)(logits.to(dev), labels.to(dev))
    assert abs(out.item() - O.boundary_dou_loss(logits, labels, 3).item()) < 1e-5
