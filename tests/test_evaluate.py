"""Device-side evaluation (cenet_amd/evaluate.py) against the oracle's restatement of the reference's
metric rules (oracle.cenet_oracle.predict / dice_metric / mean_class_dice; medpy itself is absent: parity unpinned)."""
import pytest
import torch

from backend import dev  # noqa: F401
from cenet_amd import evaluate
from oracle import cenet_oracle as O


@pytest.mark.parametrize("B,K,H,W", [(3, 4, 17, 23), (2, 9, 16, 16), (1, 2, 8, 40)])
def test_predict_counts_and_dice_rules(dev, B, K, H, W):
    g = torch.Generator().manual_seed(B + K + H)
    logits = torch.randn(B, K, H, W, generator=g)
    labels = torch.randint(0, K, (B, H, W), generator=g).float()
    if K == 9:
        labels[labels == 5] = 0  # class absent from the ground truth (pred>0 & gt==0 -> 1 rule)
    pred, counts = evaluate.predict_counts(logits.to(dev), labels.to(dev))
    ref_pred = O.predict(logits)
    assert torch.equal(pred.cpu().long(), ref_pred)
    c = counts.cpu()
    for k in range(K):
        assert int(c[k, 0]) == int(((ref_pred == k) & (labels == k)).sum())
        assert int(c[k, 1]) == int((ref_pred == k).sum()) and int(c[k, 2]) == int((labels == k).sum())
    assert abs(evaluate.binary_dice(c) - O.dice_metric(ref_pred > 0, labels > 0)) < 1e-12
    cd = evaluate.class_dice(c)
    assert abs(sum(cd) / max(len(cd), 1) - O.mean_class_dice(logits, labels, K)) < 1e-12


def test_argmax_first_maximum_on_ties(dev):
    logits = torch.zeros(1, 3, 2, 2)
    logits[0, 2, 0, 0] = 1.0
    logits[0, 1, 0, 1] = 1.0
    logits[0, 2, 0, 1] = 1.0  # tie between classes 1 and 2 -> 1 (first maximum, torch.argmax)
    pred, _ = evaluate.predict_counts(logits.to(dev))
    assert pred.cpu().long().tolist() == torch.argmax(torch.softmax(logits, 1), 1).tolist()


def test_validate_and_volume(dev):
    """a stand-in 'network' (1x1 projection) through the reference-shaped helpers: batching must not change the result"""
    g = torch.Generator().manual_seed(3)
    w = torch.randn(4, 1, generator=g)

    class Net(torch.nn.Module):
        def forward(self, x):  # [B,1,H,W] -> [B,4,H,W]
            return (x * w.view(1, 4, 1, 1).to(x.device)).contiguous()

    net = Net()
    vol = torch.randn(7, 12, 12, generator=g)
    lab = torch.randint(0, 4, (7, 12, 12), generator=g).float()
    whole = evaluate.volume_class_dice(net, vol.to(dev), lab.to(dev), 4, batch_slices=7)
    parts = evaluate.volume_class_dice(net, vol.to(dev), lab.to(dev), 4, batch_slices=3)
    assert whole == parts
    ref_logits = net(vol.unsqueeze(1))
    cd_ref = []
    pred = O.predict(ref_logits)
    for c in range(1, 4):
        p, t = pred == c, lab == c
        cd_ref.append(O.dice_metric(p, t) if (p.sum() > 0 and t.sum() > 0) else (1.0 if p.sum() > 0 else 0.0))
    assert all(abs(a - b) < 1e-12 for a, b in zip(whole, cd_ref))
    v = evaluate.validate(net, [(vol[:4].unsqueeze(1).to(dev), lab[:4].to(dev)), (vol[4:].unsqueeze(1).to(dev), lab[4:].to(dev))])
    ref = (O.dice_metric(pred[:4] > 0, lab[:4] > 0) + O.dice_metric(pred[4:] > 0, lab[4:] > 0)) / 2
    assert abs(v - ref) < 1e-12
