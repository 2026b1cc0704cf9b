"""`optim.FusedSGD` + `optim.PolyLR` against the reference's own `get_optimizer('sgd')` /
`get_scheduler('poly')` (src/utils/core.py:12-41) — vectors produced by oracle/gen_golden_sched.py from the reference."""
import os

import numpy as np
import pytest
import torch

from backend import dev  # noqa: F401  (fixture: host SIMT checker on CPU, the HIP library with -m gpu)
from conftest import GOLDEN


def _golden():
    return np.load(os.path.join(GOLDEN, "sched_poly.npz"))


class _FakeOpt:
    def __init__(self, lr):
        self.lr = lr

    def set_lr(self, lr):
        self.lr = lr


@pytest.mark.parametrize("tag", ["a", "b"])
def test_poly_lr_sequence_matches_reference_lambdalr(tag):
    """lr read by the train loop before each scheduler.step() (main_acdc.py:256-257), incl. the final lr -> 0 step"""
    from cenet_amd import optim
    z = _golden()
    opt = _FakeOpt(float(z[f"{tag}.base_lr"]))
    sch = optim.PolyLR(opt, max_iterations=int(z[f"{tag}.max_it"]))
    got = []
    for _ in range(len(z[f"{tag}.lrs"])):
        got.append(sch.get_last_lr()[0])
        sch.step()
    np.testing.assert_allclose(np.array(got), z[f"{tag}.lrs"], rtol=1e-12, atol=1e-15)


def test_fused_sgd_poly_trajectory_matches_reference(dev):
    """five SGD(momentum .9, wd 1e-4) + poly steps: parameters after every step == torch.optim.SGD + LambdaLR"""
    from cenet_amd import optim
    z = _golden()
    m = torch.nn.Module()
    m.p = torch.nn.Parameter(torch.from_numpy(z["sgd.p0"]).clone().to(dev))
    arena = optim.ParamArena(m)
    opt = optim.FusedSGD(arena, lr=0.05, momentum=0.9, weight_decay=1e-4)
    sch = optim.PolyLR(opt, max_iterations=8)
    grads = torch.from_numpy(z["sgd.grads"]).to(dev)
    for i in range(5):
        opt.zero_grad()
        m.p.grad.copy_(grads[i])
        opt.step()
        sch.step()
        np.testing.assert_allclose(m.p.detach().cpu().numpy(), z["sgd.traj"][i], rtol=2e-6, atol=2e-7)


def test_param_groups_is_persistent_and_drives_the_step(dev):
    """the torch idiom `for g in opt.param_groups: g['lr'] = x` must take effect (warm-up code, torch schedulers)"""
    from cenet_amd import optim
    m = torch.nn.Module()
    m.p = torch.nn.Parameter(torch.ones(8, device=dev))
    arena = optim.ParamArena(m)
    opt = optim.FusedSGD(arena, lr=0.5, momentum=0.0, weight_decay=0.0)
    assert opt.param_groups is opt.param_groups
    for g in opt.param_groups:
        g["lr"] = 0.25
    assert opt.lr == 0.25
    opt.zero_grad()
    m.p.grad.fill_(1.0)
    opt.step()
    np.testing.assert_allclose(m.p.detach().cpu().numpy(), np.full(8, 0.75, dtype=np.float32))
    # LambdaLR from torch drives it too (it only needs param_groups[...]['lr'] and 'initial_lr')
    opt.param_groups[0]["lr"] = 0.5
    sch = torch.optim.lr_scheduler.LambdaLR(_as_torch_optimizer(opt), lambda s: (1 - s / 4) ** 0.9)
    sch.step()
    assert abs(opt.lr - 0.5 * (1 - 1 / 4) ** 0.9) < 1e-12


def _as_torch_optimizer(fused):
    """the minimal torch.optim.Optimizer view LambdaLR needs: shares the SAME param_groups list"""
    class _View(torch.optim.Optimizer):
        def __init__(self, groups):
            self.param_groups = groups
            self.defaults = {}
            self._optimizer_step_pre_hooks, self._optimizer_step_post_hooks = {}, {}

        def step(self):
            pass
    return _View(fused.param_groups)


def test_set_to_none_grads_return_to_the_arena(dev):
    """after `p.grad = None` (zero_grad(set_to_none=True) idiom) ops.grad_buf hands the ARENA slot back, zero-filled"""
    from cenet_amd import ops, optim
    m = torch.nn.Module()
    m.p = torch.nn.Parameter(torch.ones(70, device=dev))
    m.q = torch.nn.Parameter(torch.ones(3, device=dev))
    arena = optim.ParamArena(m)
    arena.grads.fill_(7.0)  # stale values of a previous step
    m.p.grad = None
    g = ops.grad_buf(m.p)
    off, n = arena.index["p"]
    assert g.data_ptr() == arena.grads.data_ptr() + 4 * off
    assert float(g.abs().max()) == 0.0
    assert float(arena.grads[arena.index["q"][0]]) == 7.0  # other slots untouched
