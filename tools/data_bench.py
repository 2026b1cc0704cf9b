"""Host throughput of the ACDC input pipeline (cenet_amd/data.py): samples/s of RandomGenerator([224,224]) on synthetic
256x216 slices through make_train_loader with 0 / 8 / 16 / 32 / 64 worker processes and DevicePrefetcher (3 timed epochs of
4096 samples each after a warm-up epoch)."""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np
import torch

from cenet_amd import data as D


class Synth(torch.utils.data.Dataset):
    def __init__(self, n, transform):
        rng = np.random.default_rng(0)
        self.img = rng.random((64, 256, 216)).astype(np.float32)
        self.lab = rng.integers(0, 4, (64, 256, 216)).astype(np.uint8)
        self.n, self.transform = n, transform

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        s = self.transform({"image": self.img[i % 64], "label": self.lab[i % 64]})
        s["case_name"] = str(i)
        return s


ds = Synth(4096, D.RandomGenerator([224, 224]))
print("host cores:", os.cpu_count())
for nw in (0, 8, 16, 32, 64):
    if nw > (os.cpu_count() or 1):
        continue
    if nw == 0:
        ds.n = 256
    else:
        ds.n = 4096
    dl = D.make_train_loader(ds, batch_size=32, num_workers=nw, seed=0)
    it = D.DevicePrefetcher(dl, "cuda:0") if torch.cuda.is_available() else dl
    for _ in it:  # first epoch starts the workers
        pass
    t0 = time.perf_counter()
    cnt = 0
    for _ in range(1 if nw == 0 else 3):
        for b in it:
            cnt += b["image"].shape[0]
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"workers {nw:2d}: {cnt / dt:8.1f} samples/s")
    del it, dl
