"""Input pipeline of the ACDC preset: the reference's dataset classes and training-time augmentation
with the same names, arguments and random-number consumption, plus what the reference lacks to keep an MI355X fed — worker
processes with per-worker seeds, pinned staging buffers and a copy stream that uploads batch i+1 while batch i trains.

Mirrors src/datasets/dataset_acdc.py:
  * `random_rot_flip` (:15-22), `random_rotate` (:25-29), `RandomGenerator` (:32-48) — numpy / scipy on the host exactly
    as the reference does (`np.rot90`, `np.flip`, `ndimage.rotate(order=0, reshape=False)`, `zoom(order=3)` for the image and
    `zoom(order=0)` for the label); with the same seeds of `random` and `np.random` the samples are bit-identical
    (tests/test_data.py against vectors produced by the reference's own classes, oracle/gen_golden_data.py);
  * `ACDCdataset` (:51-78) and `ACDCdatasetFast` (:81-114) — same constructor and `__getitem__` contract
    (`{'image', 'label', 'case_name'}`; the transform only on split 'train').
The reference builds `DataLoader(db_train, batch_size, shuffle=True)` with `num_workers=0` (src/main_acdc.py:140): one host
process does all the resampling, 30 samples/s measured (tools/data_bench.py), against ~750 images/s per GPU for the step.
`make_train_loader` + `DevicePrefetcher` are the replacement; with `num_workers=0` the loader is the reference's.
"""
from __future__ import annotations

import os
import random
from typing import Iterable, Iterator, Optional, Sequence

import numpy as np
import torch
from scipy import ndimage
from scipy.ndimage import zoom
from torch.utils.data import DataLoader, Dataset


def _apply_pair(fn, image, label):
    return fn(image), fn(label)


def random_rot_flip(image, label):
    """dataset_acdc.py:15-22 — k quarter turns, then a flip along a random axis.  Draw order (np.random): k, then axis."""
    k = np.random.randint(0, 4)
    image, label = _apply_pair(lambda a: np.rot90(a, k), image, label)
    axis = np.random.randint(0, 2)
    return _apply_pair(lambda a: np.flip(a, axis=axis).copy(), image, label)


def random_rotate(image, label):
    """dataset_acdc.py:25-29 — whole-degree rotation in [-20, 20) (one np.random draw), nearest sampling for image and
    label alike, output shape kept."""
    angle = np.random.randint(-20, 20)
    return _apply_pair(lambda a: ndimage.rotate(a, angle, order=0, reshape=False), image, label)


def _resize_pair(image, label, size):
    """Cubic spline for the image, nearest for the label, scipy's default (non-grid) coordinate mapping (dataset_acdc.py:43-44)."""
    h, w = image.shape
    if h == size[0] and w == size[1]:
        return image, label
    factors = (size[0] / h, size[1] / w)
    return zoom(image, factors, order=3), zoom(label, factors, order=0)


class RandomGenerator(object):
    """dataset_acdc.py:32-48.  sample {'image' [H,W], 'label' [H,W]} -> {'image' float32 [1,h,w], 'label' int64 [h,w]}.
    `random` draws: one to choose the quarter-turn branch; only if that fails a second one to choose the rotation branch."""

    def __init__(self, output_size: Sequence[int]):
        self.output_size = output_size

    def __call__(self, sample):
        image, label = sample['image'], sample['label']
        if random.random() > 0.5:
            image, label = random_rot_flip(image, label)
        elif random.random() > 0.5:
            image, label = random_rotate(image, label)
        image, label = _resize_pair(image, label, self.output_size)
        image_t = torch.from_numpy(image.astype(np.float32)).unsqueeze(0)
        label_t = torch.from_numpy(label.astype(np.float32)).long()
        return {'image': image_t, 'label': label_t}


class ACDCdataset(Dataset):
    """dataset_acdc.py:51-78: `<list_dir>/<split>.txt` names .npz files holding 'img' and 'label'; train / valid entries
    are slices under `<base_dir>/<split>/`, test entries whole volumes under `<base_dir>/`."""

    def __init__(self, base_dir, list_dir, split, transform=None):
        self.transform = transform
        self.split = split
        with open(os.path.join(list_dir, self.split + '.txt')) as f:
            self.sample_list = f.readlines()
        self.data_dir = base_dir

    def __len__(self):
        return len(self.sample_list)

    def _path(self, idx):
        name = self.sample_list[idx].strip('\n')
        if self.split == "train" or self.split == "valid":
            return os.path.join(self.data_dir, self.split, name)
        return self.data_dir + "/{}".format(name)

    def _load(self, idx):
        with np.load(self._path(idx)) as data:
            return data['img'], data['label']

    def __getitem__(self, idx):
        image, label = self._load(idx)
        sample = {'image': image, 'label': label}
        if self.transform and self.split == "train":
            sample = self.transform(sample)
        sample['case_name'] = self.sample_list[idx].strip('\n')
        return sample


class ACDCdatasetFast(ACDCdataset):
    """dataset_acdc.py:81-114: everything decoded into memory once."""

    def __init__(self, base_dir, list_dir, split, transform=None):
        super().__init__(base_dir, list_dir, split, transform)
        self.all_data = [super(ACDCdatasetFast, self)._load(i) for i in range(len(self.sample_list))]

    def _load(self, idx):
        return self.all_data[idx]


def _seed_worker(worker_id: int):
    """Each worker process draws its augmentations from its own stream: torch hands every worker base_seed + id, which is
    folded into `random` and `np.random` (a forked worker would otherwise repeat the parent's numpy stream)."""
    s = torch.initial_seed() % (2 ** 32)
    random.seed(s)
    np.random.seed(s)


def make_train_loader(dataset: Dataset, batch_size: int, num_workers: int = 0, seed: Optional[int] = None,
                      drop_last: bool = False, sampler=None) -> DataLoader:
    """`DataLoader(db_train, batch_size=..., shuffle=True)` of src/main_acdc.py:140 when num_workers == 0; otherwise the same
    batches produced by worker processes into pinned memory (persistent workers, two batches ahead each).  Pass a
    `DistributedSampler` for one process per GPU."""
    g = None
    if seed is not None:
        g = torch.Generator()
        g.manual_seed(seed)
    kw = {}
    if num_workers > 0:
        kw = dict(worker_init_fn=_seed_worker, persistent_workers=True, prefetch_factor=2)
    return DataLoader(dataset, batch_size=batch_size, shuffle=sampler is None, sampler=sampler, num_workers=num_workers,
                      pin_memory=torch.cuda.is_available(), drop_last=drop_last, generator=g, **kw)


class DevicePrefetcher:
    """Iterates a loader of dict batches and returns them with their tensors already on `device`: batch i+1 is copied on a
    separate stream (from the loader's pinned buffers, non-blocking) while the caller trains on batch i; the consumer's
    stream waits on the copy's event only when it takes the batch.  Non-tensor fields (case names) pass through."""

    def __init__(self, loader: Iterable, device, float_labels: bool = True):
        self.loader, self.device = loader, torch.device(device)
        self.float_labels = float_labels  # the loss kernels take class ids as float (core.py:179-188 casts per call)
        self.stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None

    def __len__(self):
        return len(self.loader)

    def _upload(self, batch):
        out = {}
        for k, v in batch.items():
            if torch.is_tensor(v):
                v = v.to(self.device, non_blocking=True)
                if k == "label" and self.float_labels:
                    v = v.float()
            out[k] = v
        return out

    def __iter__(self) -> Iterator[dict]:
        it = iter(self.loader)
        if self.stream is None:
            for batch in it:
                yield self._upload(batch)
            return
        nxt, ev = None, None

        def fetch():
            nonlocal nxt, ev
            try:
                b = next(it)
            except StopIteration:
                nxt, ev = None, None
                return
            with torch.cuda.stream(self.stream):
                nxt = self._upload(b)
                ev = torch.cuda.Event()
                ev.record(self.stream)

        fetch()
        while nxt is not None:
            cur, cur_ev = nxt, ev
            torch.cuda.current_stream(self.device).wait_event(cur_ev)
            for v in cur.values():
                if torch.is_tensor(v):
                    v.record_stream(torch.cuda.current_stream(self.device))
            fetch()  # the next upload overlaps the caller's work on `cur`
            yield cur
