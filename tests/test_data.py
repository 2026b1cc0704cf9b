"""Input pipeline (cenet_amd/data.py) against vectors the reference's own RandomGenerator produced
(oracle/gen_golden_data.py -> tests/golden/data_acdc.npz), the dataset contract, the seeded multi-worker loader and the
device prefetcher."""
import os
import random

import numpy as np
import pytest
import torch

from cenet_amd import data as D

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "data_acdc.npz")


def test_random_generator_matches_reference_vectors():
    z = np.load(GOLD)
    n = int(z["n"])
    seen = set()
    for i in range(n):
        seed = int(z[f"c{i}.seed"])
        random.seed(seed)
        np.random.seed(seed)
        out = D.RandomGenerator([32, 32])({"image": z[f"c{i}.image"].copy(), "label": z[f"c{i}.label"].copy()})
        assert out["image"].dtype == torch.float32 and out["image"].shape == (1, 32, 32)
        assert out["label"].dtype == torch.int64 and out["label"].shape == (32, 32)
        assert np.array_equal(out["image"].numpy(), z[f"c{i}.out_image"]), (i, str(z[f"c{i}.branch"]))
        assert np.array_equal(out["label"].numpy(), z[f"c{i}.out_label"]), (i, str(z[f"c{i}.branch"]))
        seen.add(str(z[f"c{i}.branch"]))
    assert seen == {"rot_flip", "rotate", "none"}


def _make_tree(d, n_slices=6, n_vols=2, hw=(36, 40)):
    rng = np.random.default_rng(0)
    os.makedirs(os.path.join(d, "train"))
    os.makedirs(os.path.join(d, "valid"))
    os.makedirs(os.path.join(d, "lists"))
    names = {"train": [], "valid": [], "test": []}
    for i in range(n_slices):
        img = rng.random(hw).astype(np.float32)
        lab = rng.integers(0, 4, hw).astype(np.uint8)
        for split in ("train", "valid"):
            np.savez(os.path.join(d, split, f"s{i}.npz"), img=img, label=lab)
            names[split].append(f"s{i}.npz")
    for i in range(n_vols):
        np.savez(os.path.join(d, f"v{i}.npz"), img=rng.random((3,) + hw).astype(np.float32),
                 label=rng.integers(0, 4, (3,) + hw).astype(np.uint8))
        names["test"].append(f"v{i}.npz")
    for split, ns in names.items():
        with open(os.path.join(d, "lists", split + ".txt"), "w") as f:
            f.write("\n".join(ns) + "\n")
    return os.path.join(d, "lists")


@pytest.mark.parametrize("cls", [D.ACDCdataset, D.ACDCdatasetFast])
def test_dataset_contract(tmp_path, cls):
    lists = _make_tree(str(tmp_path))
    tr = cls(str(tmp_path), lists, "train", transform=D.RandomGenerator([32, 32]))
    va = cls(str(tmp_path), lists, "valid", transform=D.RandomGenerator([32, 32]))
    te = cls(str(tmp_path), lists, "test")
    assert (len(tr), len(va), len(te)) == (6, 6, 2)
    s = tr[1]
    assert set(s) == {"image", "label", "case_name"} and s["case_name"] == "s1.npz"
    assert s["image"].shape == (1, 32, 32) and s["label"].dtype == torch.int64
    v = va[1]  # the transform only runs on the training split (dataset_acdc.py:75)
    assert isinstance(v["image"], np.ndarray) and v["image"].shape == (36, 40)
    t = te[0]
    assert t["image"].shape == (3, 36, 40) and t["case_name"] == "v0.npz"


def test_multi_worker_loader_is_seeded_and_workers_differ(tmp_path):
    lists = _make_tree(str(tmp_path), n_slices=8)
    ds = D.ACDCdatasetFast(str(tmp_path), lists, "train", transform=D.RandomGenerator([32, 32]))

    def epoch(seed):
        dl = D.make_train_loader(ds, batch_size=2, num_workers=2, seed=seed)
        out = [(b["case_name"], b["image"].clone(), b["label"].clone()) for b in dl]
        del dl
        return out

    a, b, c = epoch(5), epoch(5), epoch(6)
    assert len(a) == 4 and a[0][1].shape == (2, 1, 32, 32) and a[0][2].dtype == torch.int64
    for (na, ia, la), (nb, ib, lb) in zip(a, b):  # same seed: same order, same augmentations
        assert na == nb and torch.equal(ia, ib) and torch.equal(la, lb)
    assert [x[0] for x in a] != [x[0] for x in c] or any(not torch.equal(x[1], y[1]) for x, y in zip(a, c))


def test_single_process_loader_is_the_reference_loader(tmp_path):
    lists = _make_tree(str(tmp_path))
    ds = D.ACDCdataset(str(tmp_path), lists, "train", transform=D.RandomGenerator([32, 32]))
    dl = D.make_train_loader(ds, batch_size=4, num_workers=0, seed=3)
    assert dl.num_workers == 0 and len(dl) == 2
    names = [n for b in dl for n in b["case_name"]]
    assert sorted(names) == sorted(f"s{i}.npz" for i in range(6))


def _check_prefetcher(device):
    batches = [{"image": torch.full((2, 1, 8, 8), float(i)), "label": torch.full((2, 8, 8), i, dtype=torch.int64),
                "case_name": [f"a{i}", f"b{i}"]} for i in range(5)]
    pf = D.DevicePrefetcher(batches, device)
    assert len(pf) == 5
    got = list(pf)
    assert len(got) == 5
    for i, b in enumerate(got):
        assert b["image"].device.type == torch.device(device).type and b["label"].dtype == torch.float32
        assert float(b["image"].mean()) == float(i) and float(b["label"].mean()) == float(i)
        assert b["case_name"] == [f"a{i}", f"b{i}"]


def test_prefetcher_cpu():
    _check_prefetcher("cpu")


@pytest.mark.gpu
def test_prefetcher_overlapped_upload_gpu():
    _check_prefetcher("cuda:0")
    # pinned source + work on the consumer stream between batches: values must still arrive intact
    src = [{"image": torch.randn(4, 1, 224, 224).pin_memory(), "label": torch.randint(0, 4, (4, 224, 224))} for _ in range(6)]
    acc = []
    for b in D.DevicePrefetcher(src, "cuda:0"):
        acc.append((b["image"].double().sum() + b["label"].double().sum()).item())
        torch.mm(torch.randn(512, 512, device="cuda:0"), torch.randn(512, 512, device="cuda:0"))
    want = [(s["image"].double().sum() + s["label"].double().sum()).item() for s in src]
    assert np.allclose(acc, want, rtol=0, atol=1e-6)
