"""Device-side ACDC augmentation (csrc/augment.hip through cenet_augment_acdc; cenet_amd/data.py DeviceSlices / DeviceAugmenter /
DeviceTrainLoader) against
  * the vectors the reference's own RandomGenerator produced (tests/golden/data_acdc.npz, oracle/gen_golden_data.py), and
  * cenet_amd.data.RandomGenerator — the host mirror of dataset_acdc.py:32-48, itself pinned bit-exactly to those vectors by
    tests/test_data.py — on seeded random slices of ragged sizes, up to the preset's 224 x 224 output.
Labels: bit-exact.  Images: within 1e-6 (fp64 restatement of scipy's operation order; in practice every float32 is identical,
which the tests also count).  `sim` runs the kernel sources on the host SIMT checker, `hip` (marker gpu) the gfx950 kernels."""
import os
import random

import numpy as np
import pytest
import torch

from backend import dev  # noqa: F401
from cenet_amd import data as D

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "data_acdc.npz")


def _slices(shapes, seed):
    rng = np.random.default_rng(seed)
    out = []
    for h, w in shapes:
        img = rng.random((h, w)).astype(np.float32)
        lab = rng.integers(0, 4, (h, w)).astype(np.uint8)
        out.append((img, lab))
    return out


def _host(samples, order, size, seed):
    random.seed(seed)
    np.random.seed(seed)
    g = D.RandomGenerator(list(size))
    outs = [g({"image": samples[i][0].copy(), "label": samples[i][1].copy()}) for i in order]
    return torch.stack([o["image"] for o in outs]), torch.stack([o["label"] for o in outs])


def _device(samples, order, size, seed, dev):
    sl = D.DeviceSlices(samples, dev)
    aug = D.DeviceAugmenter(sl, size)
    random.seed(seed)
    np.random.seed(seed)
    tab, dp = aug.draw(order)
    out = aug.apply(tab, dp)
    return out["image"].cpu(), out["label"].cpu(), tab


def test_reference_vectors(dev):
    z = np.load(GOLD)
    seen = set()
    for i in range(int(z["n"])):
        seed = int(z[f"c{i}.seed"])
        img, lab, tab = _device([(z[f"c{i}.image"], z[f"c{i}.label"])], [0], (32, 32), seed, dev)
        assert img.shape == (1, 1, 32, 32) and img.dtype == torch.float32 and lab.shape == (1, 32, 32)
        assert np.array_equal(lab[0].numpy().astype(np.int64), z[f"c{i}.out_label"]), (i, str(z[f"c{i}.branch"]))
        np.testing.assert_allclose(img[0].numpy(), z[f"c{i}.out_image"], rtol=0, atol=1e-6, err_msg=str(z[f"c{i}.branch"]))
        seen.add(("none", "rot_flip", "rotate")[int(tab[0, 3])])
        assert seen and ("none", "rot_flip", "rotate")[int(tab[0, 3])] == str(z[f"c{i}.branch"])
    assert seen == {"rot_flip", "rotate", "none"}


@pytest.mark.parametrize("size,shapes", [((32, 32), [(40, 36), (32, 32), (50, 60), (28, 44), (33, 32), (32, 47)]),
                                         ((64, 48), [(64, 48), (48, 64), (70, 30), (31, 90)])])
def test_ragged_batch_equals_the_host_generator(dev, size, shapes):
    samples = _slices(shapes, 3)
    exact = total = 0
    branches = set()
    for seed in range(1, 9):
        order = list(np.random.default_rng(seed).permutation(len(samples))) * 2  # every slice twice, mixed order
        hi, hl = _host(samples, order, size, seed)
        di, dl, tab = _device(samples, order, size, seed, dev)
        assert torch.equal(dl.long(), hl)
        np.testing.assert_allclose(di.numpy(), hi.numpy(), rtol=0, atol=1e-6)
        exact += int((di == hi).sum())
        total += di.numel()
        branches |= set(tab[:, 3].tolist())
    assert branches == {0, 1, 2}
    assert exact >= 0.999 * total, (exact, total)


def test_preset_size_224(dev):
    """the ACDC preset's own resize (dataset_acdc.py:42-45 with img_size 224) from slice sizes like the data set's"""
    samples = _slices([(216, 256), (154, 224), (224, 224), (256, 208)], 5)
    if dev.type == "cpu":
        samples = samples[1:3]  # (the host checker runs the fp64 prefilter lane by lane)
    order = list(range(len(samples)))
    for seed in (2, 5, 11):
        hi, hl = _host(samples, order, (224, 224), seed)
        di, dl, _ = _device(samples, order, (224, 224), seed, dev)
        assert torch.equal(dl.long(), hl)
        np.testing.assert_allclose(di.numpy(), hi.numpy(), rtol=0, atol=1e-6)


@pytest.mark.parametrize("shape", [(58, 63), (232, 256), (248, 256)])
def test_sizes_whose_last_zoom_coordinate_rounds_outside(dev, shape):
    """scipy's zoom (mode='constant') maps a coordinate above len - 1 to cval: (OH - 1) * ((Ha - 1) / (OH - 1)) rounds above Ha - 1 for
    Ha in {58, 63, 232, 248, ...} at OH = 224, so the reference's LAST output row / column is zero there, image and label alike
    (dataset_acdc.py:42-45).  The device path must do the same — not mirror the taps."""
    if dev.type == "cpu" and shape[0] > 100:
        pytest.skip("the host checker runs the fp64 prefilter lane by lane: the small case covers the rule")
    samples = _slices([shape], 9)
    zeroed = 0
    for seed in (2, 5, 11):
        hi, hl = _host(samples, [0], (224, 224), seed)
        di, dl, _ = _device(samples, [0], (224, 224), seed, dev)
        assert torch.equal(dl.long(), hl)
        np.testing.assert_allclose(di.numpy(), hi.numpy(), rtol=0, atol=1e-6)
        zeroed += int((hi[0, 0, -1] == 0).all()) + int((hi[0, 0, :, -1] == 0).all())
    assert zeroed > 0  # the case really is exercised: a whole last row or column of the reference's output is cval


def test_loader_order_is_the_dataloaders(dev):
    """same torch seed -> the index batches of DataLoader(shuffle=True) (main_acdc.py:140), epoch after epoch; and the same
    `random` / `np.random` seeds -> the host pipeline's samples"""
    from torch.utils.data import DataLoader, Dataset

    samples = _slices([(36, 40)] * 5 + [(32, 32)] * 4 + [(44, 28)] * 2, 7)

    class DS(Dataset):
        def __len__(self):
            return len(samples)

        def __getitem__(self, i):
            s = D.RandomGenerator([32, 32])({"image": samples[i][0].copy(), "label": samples[i][1].copy()})
            s["idx"] = i
            return s

    torch.manual_seed(123)
    random.seed(9)
    np.random.seed(9)
    ref_loader = DataLoader(DS(), batch_size=4, shuffle=True)
    ref = [b for _ in range(2) for b in ref_loader]
    torch.manual_seed(123)
    random.seed(9)
    np.random.seed(9)
    sl = D.DeviceSlices(samples, dev, names=[f"s{i}" for i in range(len(samples))])
    loader = D.DeviceTrainLoader(sl, (32, 32), batch_size=4, shuffle=True)
    assert len(loader) == len(ref_loader) == 3
    got = [b for _ in range(2) for b in loader]
    assert len(got) == len(ref) == 6
    for g, r in zip(got, ref):
        assert g["case_name"] == [f"s{int(i)}" for i in r["idx"]]
        assert torch.equal(g["label"].cpu().long(), r["label"])
        np.testing.assert_allclose(g["image"].cpu().numpy(), r["image"].numpy(), rtol=0, atol=1e-6)


def test_loader_with_a_distributed_sampler_shards_the_set(dev):
    """one process per GPU: DistributedSampler over the resident set, as make_train_loader(sampler=...) does for the host pipeline —
    two ranks see disjoint halves that cover the set, in the sampler's order"""
    from torch.utils.data.distributed import DistributedSampler
    samples = _slices([(32, 32)] * 10, 2)
    sl = D.DeviceSlices(samples, dev, names=[f"s{i}" for i in range(10)])
    seen = []
    for rank in range(2):
        smp = DistributedSampler(range(len(sl)), num_replicas=2, rank=rank, shuffle=True, seed=5)
        smp.set_epoch(3)
        loader = D.DeviceTrainLoader(sl, (32, 32), batch_size=2, sampler=smp)
        random.seed(rank)
        np.random.seed(rank)
        names = [n for b in loader for n in b["case_name"]]
        assert names == [f"s{i}" for i in smp] and len(names) == 5
        seen += names
    assert sorted(seen) == sorted(f"s{i}" for i in range(10))


@pytest.mark.gpu
def test_throughput_floor():
    """one host process feeds >= 2 000 augmented 224 x 224 samples/s (the reference's host pipeline: ~30/s)"""
    import time
    d = torch.device("cuda:0")
    samples = _slices([(216, 256), (232, 256), (154, 224), (256, 208)] * 64, 1)
    sl = D.DeviceSlices(samples, d)
    loader = D.DeviceTrainLoader(sl, (224, 224), batch_size=32, shuffle=True)
    for _ in loader:  # warm-up epoch
        pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    for _ in range(3):
        for b in loader:
            n += b["image"].shape[0]
    torch.cuda.synchronize()
    rate = n / (time.perf_counter() - t0)
    print(f"device augmentation: {rate:.0f} samples/s")
    assert rate >= 2000, rate
