"""TEST INFRASTRUCTURE — golden vectors for the ACDC input pipeline (reference src/datasets/dataset_acdc.py:15-114,
SURVEY.md §8f row 4).

Loads the reference's own module BY FILE PATH in THIS container (its package __init__ pulls h5py, which is absent; the
module itself needs only numpy / scipy / torch / tqdm) and runs `RandomGenerator` on seeded inputs.  Writes
tests/golden/data_acdc.npz: per case the input slice and label, the seed of `random` / `np.random`, and the reference's
output image / label — chosen so that all three branches occur (quarter turn + flip, small-angle rotation, neither), with
and without the resize.  Also checks here, once, that the reference's ACDCdataset / ACDCdatasetFast and cenet_amd.data's
return identical samples from the same files (train with transform, valid, test volumes)."""
import importlib.util
import os
import random
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
OUT = os.path.join(ROOT, "tests", "golden")


def load_reference_dataset():
    spec = importlib.util.spec_from_file_location("ref_dataset_acdc", "/root/reference/src/datasets/dataset_acdc.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def slice_pair(h, w, seed):
    """smooth image in [0,1] and a blocky 4-class label, like a cardiac slice"""
    from scipy.ndimage import zoom
    rng = np.random.default_rng(seed)
    img = zoom(rng.random((h // 4 + 1, w // 4 + 1)), 4, order=1)[:h, :w].astype(np.float32)
    lab = np.clip((zoom(rng.random((h // 8 + 1, w // 8 + 1)), 8, order=0)[:h, :w] * 4).astype(np.uint8), 0, 3)
    return img, lab


def branch_of(seed):
    random.seed(seed)
    if random.random() > 0.5:
        return "rot_flip"
    return "rotate" if random.random() > 0.5 else "none"


def main():
    R = load_reference_dataset()
    rec, want = {}, {"rot_flip": 3, "rotate": 3, "none": 2}
    shapes = [(40, 36), (32, 32), (50, 60), (28, 44)]
    seed, n = 0, 0
    while any(v > 0 for v in want.values()):
        seed += 1
        b = branch_of(seed)
        if want[b] == 0:
            continue
        want[b] -= 1
        h, w = shapes[n % len(shapes)]
        img, lab = slice_pair(h, w, 1000 + seed)
        random.seed(seed)
        np.random.seed(seed)
        out = R.RandomGenerator([32, 32])({"image": img.copy(), "label": lab.copy()})
        rec[f"c{n}.seed"] = np.int64(seed)
        rec[f"c{n}.image"], rec[f"c{n}.label"] = img, lab
        rec[f"c{n}.out_image"] = out["image"].numpy()
        rec[f"c{n}.out_label"] = out["label"].numpy()
        rec[f"c{n}.branch"] = np.array(b)
        n += 1
    rec["n"] = np.int64(n)
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "data_acdc.npz"), **rec)
    print("wrote", n, "cases ->", os.path.join(OUT, "data_acdc.npz"))

    # dataset classes: same files through both implementations
    from cenet_amd import data as D
    with tempfile.TemporaryDirectory() as d:
        os.makedirs(os.path.join(d, "train"))
        os.makedirs(os.path.join(d, "valid"))
        os.makedirs(os.path.join(d, "lists"))
        names = {"train": [], "valid": [], "test": []}
        for i in range(4):
            img, lab = slice_pair(36, 40, i)
            for split in ("train", "valid"):
                np.savez(os.path.join(d, split, f"s{i}.npz"), img=img, label=lab)
                names[split].append(f"s{i}.npz")
        for i in range(2):
            vol = np.stack([slice_pair(36, 40, 10 * i + k)[0] for k in range(3)])
            lab = np.stack([slice_pair(36, 40, 10 * i + k)[1] for k in range(3)])
            np.savez(os.path.join(d, f"v{i}.npz"), img=vol, label=lab)
            names["test"].append(f"v{i}.npz")
        for split, ns in names.items():
            with open(os.path.join(d, "lists", split + ".txt"), "w") as f:
                f.write("\n".join(ns) + "\n")
        for RC, MC in ((R.ACDCdataset, D.ACDCdataset), (R.ACDCdatasetFast, D.ACDCdatasetFast)):
            for split in ("train", "valid", "test"):
                tr_r = R.RandomGenerator([32, 32]) if split == "train" else None
                tr_m = D.RandomGenerator([32, 32]) if split == "train" else None
                a, b = RC(d, os.path.join(d, "lists"), split, tr_r), MC(d, os.path.join(d, "lists"), split, tr_m)
                assert len(a) == len(b)
                for i in range(len(a)):
                    random.seed(7 + i), np.random.seed(7 + i)
                    sa = a[i]
                    random.seed(7 + i), np.random.seed(7 + i)
                    sb = b[i]
                    assert sa["case_name"] == sb["case_name"]
                    for k in ("image", "label"):
                        xa, xb = torch.as_tensor(sa[k]), torch.as_tensor(sb[k])
                        assert xa.dtype == xb.dtype and torch.equal(xa, xb), (RC.__name__, split, i, k)
        print("dataset classes agree with the reference's on every split")


if __name__ == "__main__":
    main()
