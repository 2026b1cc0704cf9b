"""Checkpoint I/O (cenet_amd/checkpoint.py): reference-compatible weight files, resumable training state."""
import argparse
import json
import os

import pytest
import torch

from cenet_amd import checkpoint
from cenet_amd.networks import CENet
from oracle.golden_cases import MODEL_CONFIGS

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_weight_file_is_a_reference_state_dict(tmp_path):
    """keys, order and shapes of the written file = the reference's 801-entry schema; a reference-keyed dict loads strictly"""
    net = CENet(**MODEL_CONFIGS["acdc"]["kw"])
    p = str(tmp_path / "best.pth")
    checkpoint.save_weights(net, p)
    sd = torch.load(p, weights_only=True)
    ref = json.load(open(os.path.join(GOLDEN, "schema_acdc.json")))
    assert list(sd.keys()) == list(ref.keys()) and len(sd) == 801
    assert all(list(sd[k].shape) == ref[k] for k in ref)
    other = CENet(**MODEL_CONFIGS["acdc"]["kw"])
    with torch.no_grad():
        for v in sd.values():
            if v.is_floating_point():
                v.mul_(0.5).add_(0.25)
    torch.save(sd, p)
    checkpoint.load_weights(other, p)
    for k, v in other.state_dict().items():
        assert torch.equal(v.cpu(), sd[k]), k


@pytest.mark.gpu
def test_resume_reproduces_the_next_step(tmp_path):
    """train 2 steps, save, train a 3rd; a fresh model + optimizer restored from the file must take the same 3rd step"""
    from backend import use_hip
    from cenet_amd import losses, optim
    dev = use_hip()
    kw = MODEL_CONFIGS["acdc"]["kw"]
    crit = losses.Criterion(4, argparse.Namespace(loss_type="dice,ce", loss_weights="0.5,0.5"))
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 1, 224, 224, generator=g).to(dev)
    lab = torch.randint(0, 4, (2, 224, 224), generator=g).float().to(dev)

    def make():
        torch.manual_seed(11)
        net = CENet(**kw).to(dev).train()
        net.backbone.reset_drop_path(0.0)
        arena = optim.ParamArena(net, optim.cenet_segments())
        opt = optim.FusedSGD(arena, lr=0.05, momentum=0.9, weight_decay=1e-4)
        return net, opt, optim.PolyLR(opt, max_iterations=100)

    def step(net, opt, sched):
        opt.zero_grad()
        loss = crit(net(x), lab)
        loss.backward()
        opt.step()
        sched.step()
        return loss.item()

    net, opt, sched = make()
    step(net, opt, sched), step(net, opt, sched)
    p = str(tmp_path / "state.pth")
    checkpoint.save_training_state(p, net, opt, sched, extra={"epoch": 7})
    l3 = step(net, opt, sched)
    w3 = net.state_dict()["out.out.1.conv.conv.weight"].clone()
    net2, opt2, sched2 = make()
    extra = checkpoint.load_training_state(p, net2, opt2, sched2)
    assert extra == {"epoch": 7} and sched2.last_epoch == 2 and abs(opt2.lr - opt.param_groups[0]["lr"]) < 1.0
    l3b = step(net2, opt2, sched2)
    assert abs(l3 - l3b) < 1e-5, (l3, l3b)
    torch.testing.assert_close(net2.state_dict()["out.out.1.conv.conv.weight"], w3, rtol=1e-4, atol=1e-6)
    assert abs(opt2.lr - opt.lr) < 1e-12
