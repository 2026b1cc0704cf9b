"""Pins the CPU oracle (oracle/cenet_oracle.py) against golden vectors emitted by the unmodified reference
(oracle/gen_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import cenet_oracle as O
from oracle.golden_cases import CASES, MODEL_CONFIGS, NONFINITE_CASE, config_from_kwargs
from oracle.gen_golden_keys import PROBE_BUFFERS, PROBE_KEYS

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
TOL = dict(rtol=2e-4, atol=2e-5)


def load_case(name):
    z = np.load(os.path.join(GOLDEN, f"mod_{name}.npz"))
    sd = {"m." + k[3:]: torch.from_numpy(z[k]).clone() for k in z.files if k.startswith("sd.")}
    ins = [torch.from_numpy(z[f"in{i}"]).clone() for i in range(8) if f"in{i}" in z.files]
    return z, sd, ins


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_module_golden(case):
    z, sd, ins = load_case(case["name"])
    # eval-mode output
    with torch.no_grad():
        out_eval = case["oracle"]({k: v.clone() for k, v in sd.items()}, ins, False)
    np.testing.assert_allclose(out_eval.numpy(), z["out_eval"], **TOL)
    # train-mode output, input grads, parameter grads, BN buffers after the step
    sd_t = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running_" not in k else v.clone())
            for k, v in sd.items()}
    ins_t = [t.clone().requires_grad_(True) for t in ins]
    out = case["oracle"](sd_t, ins_t, True)
    np.testing.assert_allclose(out.detach().numpy(), z["out"], **TOL)
    (out * torch.from_numpy(z["cot"])).sum().backward()
    for i, t in enumerate(ins_t):
        np.testing.assert_allclose(t.grad.numpy(), z[f"gin{i}"], rtol=1e-3, atol=1e-4)
    for k in z.files:
        if k.startswith("gsd."):
            g = sd_t["m." + k[4:]].grad
            assert g is not None, k
            ref = z[k]
            np.testing.assert_allclose(g.numpy(), ref, rtol=1e-3, atol=1e-4 * max(1.0, float(np.abs(ref).max())))
        if k.startswith("after."):
            np.testing.assert_allclose(sd_t["m." + k[6:]].detach().numpy(), z[k], rtol=1e-4, atol=1e-6)


def test_nonfinite_scores_golden():
    """multihead_diffattn.py:106: the reference passes the scores through torch.nan_to_num, so a forward whose q.k products
    overflow fp32 stays finite.  The oracle restates that line; pinned on a case generated from the reference with q_proj / k_proj
    scaled until the scores are +-inf / NaN (oracle/gen_golden.py gen_nonfinite_case).  Without the nan_to_num the same forward is
    NaN — asserted too, so the case really exercises the line."""
    z, sd, ins = load_case(NONFINITE_CASE["name"])
    q = torch.nn.functional.linear(ins[0], sd["m.q_proj.weight"])
    k = torch.nn.functional.linear(ins[0], sd["m.k_proj.weight"])
    assert not torch.isfinite(q @ k.transpose(-1, -2)).all()
    with torch.no_grad():
        out = NONFINITE_CASE["oracle"]({k_: v.clone() for k_, v in sd.items()}, ins, False)
    assert torch.isfinite(out).all()
    np.testing.assert_allclose(out.numpy(), z["out_eval"], **TOL)
    import unittest.mock as mock
    with mock.patch.object(torch, "nan_to_num", lambda t, *a, **kw: t), torch.no_grad():
        raw = NONFINITE_CASE["oracle"]({k_: v.clone() for k_, v in sd.items()}, ins, False)
    assert not torch.isfinite(raw).all()


def test_loss_golden():
    z = np.load(os.path.join(GOLDEN, "loss_dice_ce.npz"))
    for K in (4, 9, 2):
        logits = torch.from_numpy(z[f"K{K}.logits"]).clone().requires_grad_(True)
        labels = torch.from_numpy(z[f"K{K}.labels"])
        loss = O.criterion(logits, labels, K)
        loss.backward()
        assert abs(loss.item() - float(z[f"K{K}.loss"])) < 1e-6
        assert abs(O.dice_loss(logits.detach(), labels, K).item() - float(z[f"K{K}.dice_loss"])) < 1e-6
        np.testing.assert_allclose(logits.grad.numpy(), z[f"K{K}.grad"], rtol=1e-4, atol=1e-8)


def test_boundary_loss_golden():
    """BoundaryDoULoss (core.py:83-131) and its 'boundary,ce' combination vs vectors produced by the reference class
    (oracle/gen_golden_boundary.py); K = 9 has an absent class."""
    z = np.load(os.path.join(GOLDEN, "loss_boundary.npz"))
    for K in (4, 9, 2):
        names, weights = str(z[f"K{K}.spec"][0]).split(","), [float(w) for w in str(z[f"K{K}.spec"][1]).split(",")]
        logits = torch.from_numpy(z[f"K{K}.logits"]).clone().requires_grad_(True)
        labels = torch.from_numpy(z[f"K{K}.labels"])
        loss = O.criterion(logits, labels, K, loss_type=names, weights=weights)
        loss.backward()
        assert abs(loss.item() - float(z[f"K{K}.loss"])) < 1e-6
        np.testing.assert_allclose(logits.grad.numpy(), z[f"K{K}.grad"], rtol=1e-4, atol=1e-8)


@pytest.mark.parametrize("name", list(MODEL_CONFIGS))
def test_schema_matches_reference(name):
    ref = json.load(open(os.path.join(GOLDEN, f"schema_{name}.json")))
    mine = {k: list(v) for k, v in O.state_dict_schema(config_from_kwargs(MODEL_CONFIGS[name]["kw"])).items()}
    assert mine == ref
    assert len(mine) == 801


@pytest.mark.slow
@pytest.mark.parametrize("name", list(MODEL_CONFIGS))
def test_model_golden(name):
    mc = MODEL_CONFIGS[name]
    kw = mc["kw"]
    cfg = config_from_kwargs(kw)
    K = kw["num_classes"]
    z = np.load(os.path.join(GOLDEN, f"model_{name}.npz"))
    sd = O.make_state_dict(cfg, seed=int(z["fill_seed"]))
    x, lab = O.synthetic_batch(mc["batch"], kw["input_channels"], K, seed=int(z["x_seed"]))
    with torch.no_grad():
        le = O.cenet_forward({k: v.clone() for k, v in sd.items()}, x, cfg, training=False)
    np.testing.assert_allclose(le[:, :, ::9, ::9].numpy(), z["logits_eval_sub"], rtol=1e-3, atol=1e-3)
    assert abs(le.double().sum().item() - float(z["logits_eval_sum"])) < 1e-3 * float(z["logits_eval_abs"])
    pred = O.predict(le)[:, ::5, ::5].numpy()
    assert (pred != z["pred_eval_sub"]).mean() < 1e-3
    assert abs(O.mean_class_dice(le, lab, K) - float(z["dice_eval"])) < 1e-4
    # one training step (stochastic depth off), then a second one
    params = {k: v for k, v in sd.items() if v.is_floating_point() and "running_" not in k}
    for v in params.values():
        v.requires_grad_(True)
    mom = {}
    for step, (lk, pk) in enumerate((("loss", "p1."), ("loss2", "p2."))):
        for v in params.values():
            v.grad = None
        lt = O.cenet_forward(sd, x, cfg, training=True, drop_masks=None)
        loss = O.criterion(lt, lab, K)
        loss.backward()
        assert abs(loss.item() - float(z[lk])) < 2e-4, (loss.item(), float(z[lk]))
        if step == 0:
            np.testing.assert_allclose(lt.detach()[:, :, ::9, ::9].numpy(), z["logits_train_sub"], rtol=1e-3, atol=1e-3)
            for k in PROBE_KEYS:
                g = params[k].grad.reshape(-1)
                ref_n = float(z["g." + k + ".norm"])
                assert abs(g.double().norm().item() - ref_n) <= 2e-3 * ref_n + 1e-7, k
                np.testing.assert_allclose(g[:16].numpy(), z["g." + k + ".head"], rtol=5e-3, atol=2e-3 * ref_n / max(1, g.numel()) ** 0.5 + 1e-7)
            for k in PROBE_BUFFERS:
                np.testing.assert_allclose(sd[k].reshape(-1)[:16].numpy(), z["b." + k], rtol=1e-3, atol=1e-5)
        with torch.no_grad():  # torch.optim.SGD(lr=.01, momentum=.9, weight_decay=1e-4) — core.py:19-21
            for k, v in params.items():
                g = v.grad + 1e-4 * v
                mom[k] = g.clone() if k not in mom else mom[k].mul_(0.9).add_(g)
                v.sub_(0.01 * mom[k])
        for k in PROBE_KEYS:
            np.testing.assert_allclose(params[k].detach().reshape(-1)[:16].numpy(), z[pk + k + ".head"], rtol=1e-3, atol=2e-5)


def load_drop_masks(z):
    """{(stage, i): (mask_attn[B], mask_mlp[B])} from model_acdc_droppath.npz"""
    out = {}
    for k in z.files:
        if k.startswith("mask."):
            _, s, i = k.split(".")
            m = torch.from_numpy(z[k])
            out[(int(s), int(i))] = (m[0], m[1])
    return out


@pytest.mark.slow
def test_model_droppath_golden():
    """stochastic depth with INJECTED keep masks (pvtv2.py:145-149; timm DropPath semantics): the oracle's drop_masks path
    against the reference run with the same masks (oracle/gen_golden_droppath.py)."""
    mc = MODEL_CONFIGS["acdc"]
    kw = mc["kw"]
    cfg = config_from_kwargs(kw)
    K = kw["num_classes"]
    z = np.load(os.path.join(GOLDEN, "model_acdc_droppath.npz"))
    sd = O.make_state_dict(cfg, seed=int(z["fill_seed"]))
    x, lab = O.synthetic_batch(mc["batch"], kw["input_channels"], K, seed=int(z["x_seed"]))
    params = {k: v for k, v in sd.items() if v.is_floating_point() and "running_" not in k}
    for v in params.values():
        v.requires_grad_(True)
    lt = O.cenet_forward(sd, x, cfg, training=True, drop_masks=load_drop_masks(z))
    loss = O.criterion(lt, lab, K)
    loss.backward()
    assert abs(loss.item() - float(z["loss"])) < 2e-4
    np.testing.assert_allclose(lt.detach()[:, :, ::9, ::9].numpy(), z["logits_train_sub"], rtol=1e-3, atol=1e-3)
    for k in PROBE_KEYS:
        g = params[k].grad.reshape(-1)
        ref_n = float(z["g." + k + ".norm"])
        assert abs(g.double().norm().item() - ref_n) <= 2e-3 * ref_n + 1e-7, k
