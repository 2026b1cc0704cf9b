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


def _blobs(shape, seed, thresh=0.55):
    """smooth random blobs: low-resolution noise upsampled, thresholded"""
    import numpy as np
    from scipy.ndimage import zoom
    rng = np.random.default_rng(seed)
    low = rng.random(tuple(max(2, s // 4) for s in shape))
    v = zoom(low, [s / l for s, l in zip(shape, low.shape)], order=1)
    return v > thresh


# surface distances / HD95 / ASSD / Jaccard (metrics_eval.py:9-21 -> medpy 0.5.2) against the scipy-based restatement:
# bit-exact distances (integer squared distances on the device, float64 square root), objects touching the volume faces,
# a single voxel, identical masks, 2-D masks, and a set large enough to span several LDS tiles and b-chunks
@pytest.mark.parametrize("shape,seed", [((6, 20, 24), 0), ((3, 9, 7), 1), ((10, 40, 36), 2), ((1, 12, 12), 3), ((24, 24), 4),
                                        ((12, 64, 64), 5)])
def test_surface_metrics_match_restated_medpy(dev, shape, seed):
    import numpy as np
    a, b = _blobs(shape, seed), _blobs(shape, seed + 100)
    if not a.any():
        a.flat[0] = True
    if not b.any():
        b.flat[-1] = True
    ta, tb = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    d1, d2 = evaluate.surface_distances(ta, tb)
    r1, r2 = O.surface_distances(a, b), O.surface_distances(b, a)
    # same multiset of distances in the same (C) order of border voxels, exactly
    assert d1.shape == r1.shape and d2.shape == r2.shape
    assert np.array_equal(d1, r1) and np.array_equal(d2, r2)
    assert evaluate.hd95(ta, tb) == O.hd95_metric(a, b)
    assert evaluate.assd(ta, tb) == O.assd_metric(a, b)
    got, want = evaluate.metric_percase(ta, tb), O.metric_percase(a, b)
    assert all(abs(g - w) < 1e-12 for g, w in zip(got, want))


def test_surface_metrics_edge_cases(dev):
    import numpy as np
    z = np.zeros((4, 8, 8), bool)
    one = z.copy()
    one[2, 3, 4] = True
    far = z.copy()
    far[0, 0, 0] = True
    full = np.ones((4, 8, 8), bool)
    T = lambda m: torch.from_numpy(m).to(dev)
    assert evaluate.metric_percase(T(one), T(one)) == O.metric_percase(one, one) == (1.0, 0.0, 1.0, 0.0)
    assert evaluate.metric_percase(T(one), T(far)) == O.metric_percase(one, far)
    assert evaluate.metric_percase(T(full), T(one)) == O.metric_percase(full, one)
    assert evaluate.metric_percase(T(one), T(z)) == (1, 0, 1, 0) == O.metric_percase(one, z)  # pred>0, gt empty
    assert evaluate.metric_percase(T(z), T(one)) == (0, 0, 0, 0) == O.metric_percase(z, one)
    assert evaluate.metric_percase(T(z), T(z)) == (0, 0, 0, 0)
    with pytest.raises(RuntimeError):
        evaluate.surface_distances(T(z), T(one))


class _ConstNet(torch.nn.Module):
    """stands in for the network: logits that depend on the input intensity only (so the zoom path decides the result)"""

    def __init__(self, K):
        super().__init__()
        self.K = K
        self.p = torch.nn.Parameter(torch.zeros(1))

    def forward(self, x):
        levels = torch.linspace(0.2, 0.8, self.K, device=x.device).view(1, self.K, 1, 1)
        return -(x - levels).abs()


# whole test_single_volume: cubic zoom to the patch size, batched prediction, order-0 zoom back, per-class metrics; compared
# with the reference's slice-by-slice procedure restated on the host
@pytest.mark.parametrize("size,patch", [((5, 30, 26), (16, 16)), ((4, 16, 16), (16, 16))])
def test_single_volume_matches_slicewise_procedure(dev, size, patch):
    import numpy as np
    from scipy.ndimage import zoom
    K = 4
    rng = np.random.default_rng(7)
    low = rng.random((size[0], 6, 6))
    image = np.stack([zoom(s, (size[1] / 6, size[2] / 6), order=1) for s in low]).astype(np.float32)
    label = np.clip((image * K).astype(np.int64), 0, K - 1).astype(np.float32)
    net = _ConstNet(K).to(dev)
    got = evaluate.test_single_volume(torch.from_numpy(image)[None], torch.from_numpy(label)[None], net, K, patch_size=patch,
                                      batch_slices=2, device=dev)
    # the reference's procedure (metrics_eval.py:37-71) on the host
    pred = np.zeros_like(label)
    for i in range(size[0]):
        s = image[i]
        x, y = s.shape
        if (x, y) != tuple(patch):
            s = zoom(s, (patch[0] / x, patch[1] / y), order=3)
        out = O.predict(net.cpu()(torch.from_numpy(s)[None, None].float())).squeeze(0).numpy()
        net.to(dev)
        pred[i] = zoom(out, (x / patch[0], y / patch[1]), order=0) if (x, y) != tuple(patch) else out
    want = [O.metric_percase(pred == c, label == c) for c in range(1, K)]
    for g, w in zip(got, want):
        assert all(abs(a - b) < 1e-12 for a, b in zip(g, w))


@pytest.mark.gpu
def test_batched_eval_equals_slice_by_slice_on_real_cenet():
    """CENet is NOT batch-invariant in eval — CCU applies its BatchNorm1d only `if B > 1` (cfam.py:260) and the
    reference evaluates one slice per forward (metrics_eval.py:46-49).  Under `ops.batch1_semantics()` (what evaluate.py
    uses) a batch of slices must give the logits of the slice-by-slice procedure; without it, it must not."""
    from backend import use_hip
    from cenet_amd import ops
    from cenet_amd.networks import CENet
    from oracle import cenet_oracle as O
    from oracle.golden_cases import MODEL_CONFIGS, config_from_kwargs
    dev = use_hip()
    kw = MODEL_CONFIGS["acdc"]["kw"]
    net = CENet(**kw)
    net.load_state_dict(O.make_state_dict(config_from_kwargs(kw), seed=42), strict=True)
    net = net.to(dev).eval()
    x = torch.randn(3, 1, 224, 224, generator=torch.Generator().manual_seed(8)).to(dev)
    with torch.no_grad():
        single = torch.cat([net(x[i:i + 1]) for i in range(3)])
        with ops.batch1_semantics():
            batched = net(x)
        plain = net(x)
    torch.testing.assert_close(batched, single, rtol=1e-4, atol=1e-4)
    assert (plain - single).abs().max().item() > 1e-3  # the B > 1 branch really is a different function


class _Proj(torch.nn.Module):
    """a stand-in network: fixed 1x1 projection of the (normalised) slice to K logit maps"""

    def __init__(self, K, cin=1):
        super().__init__()
        g = torch.Generator().manual_seed(2)
        self.w = torch.nn.Parameter(torch.randn(K, cin, generator=g) * 3)
        self.b = torch.nn.Parameter(torch.randn(K, generator=g))

    def forward(self, x):
        return torch.einsum("kc,bchw->bkhw", self.w.to(x.dtype), x) + self.b.view(1, -1, 1, 1)


def test_synapse_volume_wrapper_follows_utils_synapse(dev):
    """utils_synapse.py:49-97 restated on the host (zoom order 3 in, Normalize([0.5],[0.5]), argmax, zoom order 0 out, then
    (dice, hd95) per class with the 1,0 / 0,0 rules) against evaluate.synapse_test_single_volume; medpy absent: the oracle's
    scipy restatement of its surface distances is the checker (parity unpinned, as for the ACDC wrapper)."""
    import numpy as np
    from scipy.ndimage import zoom
    K, D, H, W, P = 4, 5, 20, 24, 16
    g = torch.Generator().manual_seed(1)
    vol = torch.rand(1, D, H, W, generator=g)
    lab = torch.randint(0, K, (1, D, H, W), generator=g).float()
    lab[lab == 3] = 0  # class 3 absent from the ground truth
    net = _Proj(K).to(dev)
    got = evaluate.synapse_test_single_volume(vol, lab, net, K, patch_size=(P, P), batch_slices=2, device=dev)
    pred = np.zeros((D, H, W), dtype=np.float32)
    for i in range(D):
        sl = zoom(vol[0, i].numpy(), (P / H, P / W), order=3)
        x = ((torch.from_numpy(sl).float() - 0.5) / 0.5)[None, None]
        out = torch.argmax(torch.softmax(net.cpu()(x), dim=1), dim=1)[0].numpy()
        pred[i] = zoom(out, (H / P, W / P), order=0)
    net.to(dev)
    for c in range(1, K):
        p, t = pred == c, lab[0].numpy() == c
        if p.sum() > 0 and t.sum() > 0:
            want = (O.dice_metric(torch.from_numpy(p), torch.from_numpy(t)), O.hd95_metric(p, t))
        elif p.sum() > 0:
            want = (1, 0)
        else:
            want = (0, 0)
        assert abs(got[c - 1][0] - want[0]) < 1e-12 and abs(got[c - 1][1] - want[1]) < 1e-9, (c, got[c - 1], want)


def test_skin_wrappers_follow_utils_skin(dev):
    """utils_skin.py:97-113 (val) and :131-170 (test) restated with numpy on the host: dc on the broadcast of the argmax output
    against the label batch, pixel accuracy against label[0, 0], calc_iou with its 0.0 for an empty union."""
    import numpy as np
    g = torch.Generator().manual_seed(4)
    net = _Proj(2, cin=3).to(dev)
    loader = [{"image": torch.rand(1, 3, 12, 10, generator=g), "label": torch.randint(0, 2, (1, 1, 12, 10), generator=g).float(),
               "id": torch.tensor([i])} for i in range(3)]
    loader.append({"image": torch.zeros(1, 3, 12, 10), "label": torch.zeros(1, 1, 12, 10), "id": torch.tensor([3])})

    def np_dc(a, b):
        a, b = a.astype(bool), b.astype(bool)
        s = np.count_nonzero(a) + np.count_nonzero(b)
        return 2.0 * np.count_nonzero(a & b) / s if s else 0.0

    dcs, corr, tot, ious = [], 0, 0, []
    cpu = _Proj(2, cin=3)
    for b in loader:
        pd = torch.argmax(torch.softmax(cpu(b["image"]), dim=1), dim=1).squeeze(0).numpy()
        gt = b["label"][0, 0].numpy()
        dcs.append(np_dc(pd, b["label"].numpy()))
        corr += (pd == gt).sum()
        tot += gt.size
        u = np.logical_or(pd > 0.5, gt > 0.5).sum()
        ious.append(np.logical_and(pd > 0.5, gt > 0.5).sum() / u if u > 0 else 0.0)
    d, acc, iou = evaluate.skin_test(net, loader, device=dev)
    assert abs(d - np.mean(dcs)) < 1e-12 and abs(acc - corr / tot) < 1e-12 and abs(iou - np.mean(ious)) < 1e-12
    assert abs(evaluate.skin_val(net, loader, device=dev) - np.mean(dcs)) < 1e-12
