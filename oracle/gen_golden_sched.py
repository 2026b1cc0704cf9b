"""TEST INFRASTRUCTURE — golden vectors for the optimiser plumbing of the train step.

Imports the reference's own `get_optimizer` / `get_scheduler` (src/utils/core.py:12-41) in THIS container and records
  * the learning-rate sequence of `get_scheduler(opt, args(scheduler='poly'), max_iterations)` as the train loop reads it
    (`scheduler.get_last_lr()` BEFORE `scheduler.step()`, main_acdc.py:256-257) for two (base_lr, max_iterations) pairs;
  * five SGD(momentum .9, wd 1e-4) + poly steps of a 37-element parameter vector under a fixed gradient sequence
    (parameters after every step) — what `optim.FusedSGD` + `optim.PolyLR` must reproduce.
Writes tests/golden/sched_poly.npz."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.gen_golden import OUT, load_reference_losses  # noqa: E402


def main():
    core = load_reference_losses()
    rec = {}
    for tag, base_lr, max_it, n in (("a", 0.05, 40, 40), ("b", 0.01, 30000, 64)):
        p = torch.nn.Parameter(torch.zeros(3))
        args = argparse.Namespace(optimizer="sgd", base_lr=base_lr, weight_decay=1e-4, scheduler="poly")
        opt = core.get_optimizer(torch.nn.ParameterList([p]), args)
        sch = core.get_scheduler(opt, args, max_it)
        lrs = []
        for _ in range(n):
            opt.step()
            lrs.append(sch.get_last_lr()[0])  # main_acdc.py:256
            sch.step()                         # main_acdc.py:257
        rec[f"{tag}.base_lr"], rec[f"{tag}.max_it"] = np.float64(base_lr), np.int64(max_it)
        rec[f"{tag}.lrs"] = np.array(lrs, dtype=np.float64)
    g = torch.Generator().manual_seed(99)
    p0 = torch.randn(37, generator=g)
    grads = torch.randn(5, 37, generator=g)
    p = torch.nn.Parameter(p0.clone())
    args = argparse.Namespace(optimizer="sgd", base_lr=0.05, weight_decay=1e-4, scheduler="poly")
    opt = core.get_optimizer(torch.nn.ParameterList([p]), args)
    sch = core.get_scheduler(opt, args, 8)
    traj = []
    for i in range(5):
        opt.zero_grad()
        p.grad = grads[i].clone()
        opt.step()
        sch.step()
        traj.append(p.detach().clone().numpy())
    rec["sgd.p0"], rec["sgd.grads"], rec["sgd.traj"] = p0.numpy(), grads.numpy(), np.stack(traj)
    np.savez_compressed(os.path.join(OUT, "sched_poly.npz"), **rec)
    print("[golden] sched_poly", rec["a.lrs"][:4], rec["b.lrs"][-2:])


if __name__ == "__main__":
    main()
