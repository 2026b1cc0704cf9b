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

Round 5, the MI355X-native form of the same pipeline: `DeviceSlices` keeps the whole training set in HBM (ACDC: 1 312 slices,
0.35 GB of 288), `DeviceAugmenter` draws the reference's random numbers on the host IN THE REFERENCE'S ORDER and runs the
resampling of a whole batch as three kernel launches (csrc/augment.hip, cenet_augment_acdc), `DeviceTrainLoader` is the
`DataLoader(db_train, batch_size, shuffle=True)` of main_acdc.py:140 over them — same permutation from the same torch seed, same
augmentation decisions from the same `random` / `np.random` seeds, no worker processes, no host copies of pixels.
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


# ---------------------------------------------------------------------------------------------------------------------------
# Device-resident training set + device-side augmentation (csrc/augment.hip)
# ---------------------------------------------------------------------------------------------------------------------------
_SPLINE_POLE = np.sqrt(3.0) - 2.0  # the cubic B-spline prefilter's pole (scipy ni_splines)


class DeviceSlices:
    """All (image [H, W] float32, label [H, W] uint8 class ids) training slices, sizes may differ, packed into two device pools."""

    def __init__(self, samples: Sequence, device, names: Optional[Sequence[str]] = None):
        imgs, labs, shapes = [], [], []
        for img, lab in samples:
            img, lab = np.asarray(img), np.asarray(lab)
            assert img.ndim == 2 and img.shape == lab.shape, (img.shape, lab.shape)
            assert lab.min() >= 0 and lab.max() < 256, "labels are class ids"
            imgs.append(np.ascontiguousarray(img, dtype=np.float32).reshape(-1))
            labs.append(np.ascontiguousarray(lab).astype(np.uint8).reshape(-1))
            shapes.append(img.shape)
        self.shapes = np.asarray(shapes, dtype=np.int64).reshape(-1, 2)
        self.offsets = np.concatenate([[0], np.cumsum(self.shapes[:, 0] * self.shapes[:, 1])]).astype(np.int64)
        self.device = torch.device(device)
        self.pool_img = torch.from_numpy(np.concatenate(imgs)).to(self.device)
        self.pool_lab = torch.from_numpy(np.concatenate(labs)).to(self.device)
        self.max_h, self.max_w = int(self.shapes[:, 0].max()), int(self.shapes[:, 1].max())
        self.max_side = max(self.max_h, self.max_w)  # (a quarter turn swaps the sides)
        self.names = list(names) if names is not None else None

    @classmethod
    def from_dataset(cls, ds: "ACDCdataset", device):
        """every slice of an ACDCdataset / ACDCdatasetFast (train split; its host transform, if any, is not used)"""
        return cls([ds._load(i) for i in range(len(ds))], device, names=[n.strip('\n') for n in ds.sample_list])

    def __len__(self):
        return len(self.shapes)


class DeviceAugmenter:
    """RandomGenerator (dataset_acdc.py:32-48) for a batch of DeviceSlices indices -> {'image' [B, 1, h, w] float32, 'label'
    [B, h, w] float32 class ids} on the device.  `draw` consumes `random` and `np.random` exactly as the reference's
    __getitem__ calls would for the same samples in the same order (one `random.random()`; then k and axis from np.random, or a
    second `random.random()` and the angle), so a seeded run makes the reference's augmentation decisions."""

    def __init__(self, slices: DeviceSlices, output_size: Sequence[int]):
        self.slices, self.output_size = slices, (int(output_size[0]), int(output_size[1]))
        self.stride = slices.max_side * slices.max_side
        self._stage = {}

    def draw(self, indices: Sequence[int]):
        OH, OW = self.output_size
        tab = np.zeros((len(indices), 8), dtype=np.int64)
        dp = np.zeros((len(indices), 10), dtype=np.float64)
        for n, idx in enumerate(indices):
            H, W = (int(v) for v in self.slices.shapes[idx])
            mode = k = axis = 0
            if random.random() > 0.5:  # dataset_acdc.py:38-39 -> :15-22
                mode, k, axis = 1, int(np.random.randint(0, 4)), int(np.random.randint(0, 2))
            elif random.random() > 0.5:  # :40-41 -> :25-29; the matrix and offset scipy.ndimage.rotate(reshape=False) builds
                mode = 2
                ang = np.deg2rad(np.random.randint(-20, 20))
                c, s_ = np.cos(ang), np.sin(ang)
                m = np.array([[c, s_], [-s_, c]], dtype=np.float64)
                ctr = (np.array([H, W], dtype=np.float64) - 1) / 2
                off = ctr - m @ ctr
                dp[n, 0:4] = m.reshape(-1)
                dp[n, 4:6] = off
            Ha, Wa = (W, H) if (mode == 1 and k & 1) else (H, W)
            resize = int(Ha != OH or Wa != OW)  # :42-45
            tab[n] = (self.slices.offsets[idx], H, W, mode, k, axis, resize, 0)
            dp[n, 6], dp[n, 7] = _SPLINE_POLE ** (Ha - 1), _SPLINE_POLE ** (Wa - 1)
            dp[n, 8] = (Ha - 1) / (OH - 1)  # scipy zoom, grid_mode=False: input coordinate = output index * (in - 1) / (out - 1)
            dp[n, 9] = (Wa - 1) / (OW - 1)
        return tab, dp

    def apply(self, tab: np.ndarray, dp: np.ndarray) -> dict:
        from . import kern
        sl, dev = self.slices, self.slices.device
        B = tab.shape[0]
        OH, OW = self.output_size
        if self._stage.get("B", 0) < B:
            self._stage = {"B": B, "img": torch.empty(B * self.stride, dtype=torch.float64, device=dev),
                           "lab": torch.empty(B * self.stride, dtype=torch.uint8, device=dev)}
        pin = dev.type == "cuda"
        tab_t, dp_t = torch.from_numpy(tab), torch.from_numpy(dp)
        if pin:
            tab_t, dp_t = tab_t.pin_memory(), dp_t.pin_memory()
        tab_d, dp_d = tab_t.to(dev, non_blocking=True), dp_t.to(dev, non_blocking=True)
        image = torch.empty(B, 1, OH, OW, dtype=torch.float32, device=dev)
        label = torch.empty(B, OH, OW, dtype=torch.float32, device=dev)
        kern.augment_acdc(sl.pool_img, sl.pool_lab, tab_d, dp_d, self._stage["img"], self._stage["lab"], self.stride,
                          sl.max_side, sl.max_side, image, label, B, OH, OW)
        return {"image": image, "label": label}

    def __call__(self, indices: Sequence[int]) -> dict:
        return self.apply(*self.draw(indices))


class DeviceTrainLoader:
    """`DataLoader(db_train, batch_size=B, shuffle=True)` (main_acdc.py:140) over DeviceSlices: the index order is torch's own
    RandomSampler / BatchSampler (and one draw for the iterator's base seed, as DataLoader makes), so the same torch seed gives
    the reference's batches; every batch is {'image', 'label', 'case_name'} with the tensors already on the device."""

    def __init__(self, slices: DeviceSlices, output_size: Sequence[int], batch_size: int, shuffle: bool = True,
                 drop_last: bool = False, generator: Optional[torch.Generator] = None, sampler=None):
        """sampler: an index sampler over range(len(slices)) instead of the shuffle — one process per GPU passes
        `DistributedSampler(range(len(slices)), num_replicas=world, rank=rank)` (and calls its set_epoch), exactly as with a DataLoader;
        every rank keeps the whole set resident (0.35 GB) and augments its own shard."""
        from torch.utils.data import BatchSampler, RandomSampler, SequentialSampler
        self.slices, self.aug = slices, DeviceAugmenter(slices, output_size)
        self.generator = generator
        src = range(len(slices))
        if sampler is None:
            sampler = RandomSampler(src, generator=generator) if shuffle else SequentialSampler(src)
        self.sampler = sampler
        self.batch_sampler = BatchSampler(sampler, batch_size, drop_last)

    def __len__(self):
        return len(self.batch_sampler)

    def __iter__(self) -> Iterator[dict]:
        torch.empty((), dtype=torch.int64).random_(generator=self.generator)  # _BaseDataLoaderIter's base-seed draw
        for idx in self.batch_sampler:
            batch = self.aug(idx)
            if self.slices.names is not None:
                batch["case_name"] = [self.slices.names[i] for i in idx]
            yield batch
