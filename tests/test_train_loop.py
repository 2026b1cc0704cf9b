"""cenet_amd.train.train_acdc — the reference's training loop (src/main_acdc.py:200-290) on the captured step: loader ->
DevicePrefetcher -> static input buffers -> one hipGraph replay per iteration -> poly LR -> validation on the device.
GPU: the replayed loop equals the eager loop on CHANGING batches (a ragged last batch included), iteration by iteration."""
import copy

import numpy as np
import pytest
import torch

from backend import use_hip
from oracle import cenet_oracle as O


class _Toy(torch.utils.data.Dataset):
    """in-memory slices with a different image / label per index (what ACDCdataset + RandomGenerator hand the loader)"""

    def __init__(self, n, seed):
        x, lab = O.synthetic_batch(n, 1, 4, seed=seed)
        self.x, self.lab = x, lab

    def __len__(self):
        return self.x.shape[0]

    def __getitem__(self, i):
        return {"image": self.x[i], "label": self.lab[i].long(), "case_name": f"case{i}"}


def _net(dev):
    from cenet_amd.networks import CENet
    torch.manual_seed(77)
    net = CENet(input_channels=1, num_classes=4, scale_factors=[1.0, 0.5], diffatt_num_heads=[4, 4, 4], out_up_block="upcn")
    net.backbone.reset_drop_path(0.0)  # (stochastic depth draws random masks: both loops must see the same network)
    return net.to(dev)


@pytest.mark.gpu
@pytest.mark.parametrize("bf16", [False, True], ids=["f32", "bf16"])
def test_replayed_loop_equals_the_eager_loop_on_changing_batches(bf16):
    from cenet_amd import kern, train
    dev = use_hip()
    ds = _Toy(14, seed=21)  # batch 4: three full batches + a ragged one of 2 per epoch
    loader = torch.utils.data.DataLoader(ds, batch_size=4, shuffle=False)
    xv, lv = O.synthetic_batch(3, 1, 4, seed=5)
    val = lambda: [(xv[i:i + 1].to(dev), lv[i:i + 1].to(dev)) for i in range(3)]  # noqa: E731
    ref_net = _net(dev)
    old = kern.set_compute_bf16(bf16)
    try:
        runs = []
        for graph in (False, True):
            net = copy.deepcopy(ref_net)
            h = train.train_acdc(net, loader, max_epochs=2, base_lr=0.01, device=dev, graph=graph, val_batches=val, log_every=1,
                                 log=lambda s: None)
            runs.append((h, net))
    finally:
        kern.set_compute_bf16(old)
    (he, ne), (hg, ng) = runs
    assert len(he["loss"]) == len(hg["loss"]) == 8 and he["lr"] == hg["lr"]
    le, lg = np.array([v for _, v in he["loss"]]), np.array([v for _, v in hg["loss"]])
    # the same kernels on the same data in the same order: the two loops differ only through the order of float atomics
    tol = 2e-2 if bf16 else 2e-4
    assert np.abs(le - lg).max() < tol, (le, lg)
    assert len(set(np.round(le, 4))) > 4  # the batches (and therefore the losses) do change from iteration to iteration
    assert abs(he["val_dice"][-1] - hg["val_dice"][-1]) < (5e-2 if bf16 else 5e-3)
    pe, pg = he["arena"].params, hg["arena"].params
    cos = torch.nn.functional.cosine_similarity(pe - pe.mean(), pg - pg.mean(), dim=0).item()
    assert cos > 0.99999
    # the replayed loop really replayed: one captured graph, eight iterations counted by the optimizer
    assert hg["optimizer"]._steps == 8 and he["optimizer"]._steps == 8


@pytest.mark.gpu
def test_loop_fed_by_the_device_loader_equals_the_host_pipeline():
    """train_acdc over DeviceTrainLoader (slices resident in HBM, augmentation on the device) against the same loop over the
    host pipeline (DataLoader(shuffle=True) of RandomGenerator samples, main_acdc.py:136-140) from the same seeds of torch,
    `random` and `np.random`: same batches in the same order, so the same losses."""
    import random

    from cenet_amd import data as D, train
    dev = use_hip()
    rng = np.random.default_rng(3)
    shapes = [(216, 256), (154, 224), (224, 224), (256, 208)] * 2
    samples = [(rng.random(s).astype(np.float32), rng.integers(0, 4, s).astype(np.uint8)) for s in shapes]

    class Host(torch.utils.data.Dataset):
        def __len__(self):
            return len(samples)

        def __getitem__(self, i):
            return D.RandomGenerator([224, 224])({"image": samples[i][0].copy(), "label": samples[i][1].copy()})

    ref_net = _net(dev)
    losses = []
    for device_side in (False, True):
        torch.manual_seed(31)
        random.seed(4)
        np.random.seed(4)
        if device_side:
            loader = D.DeviceTrainLoader(D.DeviceSlices(samples, dev), (224, 224), batch_size=4, shuffle=True)
        else:
            loader = torch.utils.data.DataLoader(Host(), batch_size=4, shuffle=True)
        h = train.train_acdc(copy.deepcopy(ref_net), loader, max_epochs=2, base_lr=0.01, device=dev, graph=True, log_every=1,
                             log=lambda s: None)
        losses.append(np.array([v for _, v in h["loss"]]))
    assert len(losses[0]) == len(losses[1]) == 4
    assert np.abs(losses[0] - losses[1]).max() < 2e-4, losses
