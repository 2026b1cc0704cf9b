"""The reference's training loop (src/main_acdc.py:200-290) on the MI355X-native step: loader -> DevicePrefetcher -> STATIC input
buffers -> one hipGraph replay per iteration (cenet_amd.graph.GraphedStep) -> poly learning rate -> per-epoch validation on the
device (cenet_amd.evaluate.validate).

What differs from the reference loop and why:
  * the batch is copied into two static device tensors and the captured step reads those: a hipGraph replays fixed addresses
    (main_acdc.py:238-240 moves a fresh tensor to the GPU per iteration); the copy is one device-to-device kernel per tensor on
    the training stream, the host-to-device upload already happened on the prefetcher's copy stream;
  * a ragged last batch of an epoch (DataLoader(drop_last=False), main_acdc.py:139) runs eagerly — BatchNorm statistics and the
    Dice sums depend on the batch, so it cannot be padded;
  * the loss is read back once per `log_every` iterations instead of every iteration (main_acdc.py:264-265: a device sync per
    step); the running epoch loss is accumulated on the device;
  * the learning rate for the NEXT step is uploaded by FusedSGD.prepare() before each replay (no host value is baked into the
    capture);
  * the epoch log line prints the reference's quantity — the sum of the batch losses over the number of training images
    (main_acdc.py:264-267) — but accumulated on the device and read back once per epoch;
  * NOT here: the reference's `best.pth` / last-epoch checkpoint saves and its test inference after training
    (main_acdc.py:272-289) — cenet_amd.checkpoint.save_weights / save_training_state and cenet_amd.evaluate are the pieces a caller
    combines for that (tests/test_checkpoint.py, tests/test_evaluate.py); the loop returns the arena and the optimizer for it.
`train_acdc(..., graph=False)` is the same loop with eager launches: tests/test_train_loop.py holds the two against each other
on changing batches."""
from __future__ import annotations

import argparse
from typing import Callable, Iterable, List, Optional

import torch

from . import graph as G
from . import losses, optim
from .data import DevicePrefetcher


def _as_batches(loader) -> Iterable[dict]:
    return loader


def train_acdc(net, tr_loader, *, num_classes: int = 4, max_epochs: int = 1, base_lr: float = 0.05, momentum: float = 0.9,
               weight_decay: float = 1e-4, loss_type: str = "dice,ce", loss_weights: str = "0.5,0.5", device="cuda:0",
               graph: bool = True, val_batches: Optional[Callable[[], Iterable]] = None, log_every: int = 20,
               log: Callable[[str], None] = print) -> dict:
    """main_acdc.py:200-290.  tr_loader yields {'image' [B,1,H,W] float, 'label' [B,H,W]} batches (host or device tensors);
    val_batches() yields (image [1,1,H,W], label [1,H,W]) pairs for the per-epoch validation (main_acdc.py:218-231).
    -> {'loss': per-iteration losses read back at log points, 'epoch_loss': [...], 'val_dice': [...], 'lr': [...]}"""
    from . import evaluate
    dev = torch.device(device)
    net = net.to(dev).train()
    crit = losses.Criterion(num_classes, argparse.Namespace(loss_type=loss_type, loss_weights=loss_weights))
    arena = optim.ParamArena(net, optim.cenet_segments())
    opt = optim.FusedSGD(arena, lr=base_lr, momentum=momentum, weight_decay=weight_decay)
    sched = optim.PolyLR(opt, max_iterations=max_epochs * len(tr_loader))
    hist = {"loss": [], "epoch_loss": [], "val_dice": [], "lr": []}
    static = {}
    stepper: List[Optional[G.GraphedStep]] = [None]

    def body(x, lab):
        opt.zero_grad()
        loss = crit(net(x), lab)
        loss.backward()
        opt.step(sync_hyper=False)
        return loss

    def one_step(x, lab):
        if graph and (not static or x.shape == static["x"].shape):
            if not static:
                static["x"], static["lab"] = torch.empty_like(x), torch.empty_like(lab)
            static["x"].copy_(x, non_blocking=True)
            static["lab"].copy_(lab, non_blocking=True)
            if stepper[0] is None:
                # (the capture's warm-up runs are REAL optimizer steps: keep the model where it is with a snapshot)
                snap = (arena.params.clone(), opt.buf.clone(), opt._steps,
                        {k: v.clone() for k, v in net.state_dict().items() if "running_" in k or "num_batches" in k})
                stepper[0] = G.GraphedStep(lambda: body(static["x"], static["lab"]), optimizer=opt, warmup=2)
                arena.params.copy_(snap[0])
                opt.buf.copy_(snap[1])
                opt._steps = snap[2]
                net.load_state_dict(snap[3], strict=False)
                arena.refresh_shadow()
            return stepper[0]()
        opt.prepare()
        return body(x, lab)

    it = 0
    for epoch in range(max_epochs):
        net.train()
        run = torch.zeros((), device=dev)
        n_img = 0
        for batch in DevicePrefetcher(_as_batches(tr_loader), dev):
            x, lab = batch["image"].float(), batch["label"].float()
            loss = one_step(x, lab)
            hist["lr"].append(sched.get_last_lr()[0])
            sched.step()
            it += 1
            run += loss.detach().float()  # main_acdc.py:264: the plain batch loss ...
            n_img += x.shape[0]
            if it % log_every == 0:
                hist["loss"].append((it, loss.item()))
                log(f"iteration {it} : loss : {hist['loss'][-1][1]:f} lr_: {hist['lr'][-1]:f}")
        hist["epoch_loss"].append(run.item() / max(n_img, 1))  # ... over len(db_train) (main_acdc.py:266), as the reference logs it
        if val_batches is not None:
            hist["val_dice"].append(evaluate.validate(net, val_batches()))
            net.train()
        lr_last = hist["lr"][-1] if hist["lr"] else base_lr  # (an empty loader takes no step)
        log(f"epoch:{epoch:03d}/{max_epochs}, loss:{hist['epoch_loss'][-1]:0.5f}, lr:{lr_last:0.6f}"
            + (f", vl_DCS:{hist['val_dice'][-1] * 100:0.3f}" if hist["val_dice"] else ""))
    hist["arena"], hist["optimizer"] = arena, opt
    return hist
