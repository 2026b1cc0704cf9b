"""Parity of the HIP product modules (cenet_amd.networks.*) against the reference golden vectors.

Every case of oracle/golden_cases.py is replayed through the product module built with the SAME constructor
arguments as the reference class: eval output, train output, input gradients, parameter gradients and BN
running buffers after one training forward.  `sim` runs the kernel sources on the host SIMT checker (CPU);
`hip` (marker gpu) runs the real gfx950 kernels through the C ABI.  Tolerance: fp32, 1e-3 relative on values
(north_star: logits within 1e-3).
"""
import importlib
import os
from functools import partial

import numpy as np
import pytest
import torch
import torch.nn as nn

from backend import dev  # noqa: F401
from oracle.golden_cases import CASES, NONFINITE_CASE

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def build_product(case, device):
    modname, clsname, kw = case["ref"]
    kw = dict(kw)
    if kw.get("norm_layer") == "LN6":
        kw["norm_layer"] = partial(nn.LayerNorm, eps=1e-6)
    mod = importlib.import_module("cenet_amd." + modname)
    m = getattr(mod, clsname)(**kw)
    return m.to(device)


def call(case, m, ins):
    name = case["name"]
    if name.startswith("patch_embed"):
        return m(ins[0])[0]
    if name == "pvt_block_sr2":
        return m(ins[0], 8, 8)
    if name == "pvt_block_sr1":
        return m(ins[0], 4, 5)
    if len(ins) == 2:
        return m(ins[0], ins[1])
    return m(ins[0])


def close(a, b, rtol=1e-3, atol=None, what=""):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a
    scale = max(1.0, float(np.abs(b).max()))
    np.testing.assert_allclose(a, b, rtol=rtol, atol=(atol if atol is not None else 2e-4 * scale), err_msg=what)


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_module_parity(dev, case):
    z = np.load(os.path.join(GOLDEN, f"mod_{case['name']}.npz"))
    m = build_product(case, dev)
    sd = {k[3:]: torch.from_numpy(z[k]).to(dev) for k in z.files if k.startswith("sd.")}
    missing = m.load_state_dict(sd, strict=True)
    ins = [torch.from_numpy(z[f"in{i}"]).to(dev) for i in range(4) if f"in{i}" in z.files]
    m.eval()
    with torch.no_grad():
        close(call(case, m, ins), z["out_eval"], what="eval output")
    m.load_state_dict(sd, strict=True)
    m.train()
    ins_t = [t.clone().requires_grad_(True) for t in ins]
    out = call(case, m, ins_t)
    close(out, z["out"], what="train output")
    out.backward(torch.from_numpy(z["cot"]).to(dev))
    for i, t in enumerate(ins_t):
        close(t.grad, z[f"gin{i}"], rtol=2e-3, what=f"grad input {i}")
    params = dict(m.named_parameters())
    bufs = dict(m.named_buffers())
    # BatchNorm1d over a batch of TWO (ccu_b2) is ill-conditioned in fp32: the reference's own fp32 gradient is
    # 3.4e-3 away from an fp64 evaluation of the same formula, so that case gets a matching absolute tolerance.
    loose = 8e-3 if case["name"] == "ccu_b2" else None
    for k in z.files:
        if k.startswith("gsd."):
            g = params[k[4:]].grad
            assert g is not None, f"no gradient accumulated for {k[4:]}"
            close(g, z[k], rtol=2e-3, atol=loose, what=k)
        if k.startswith("after."):
            close(bufs[k[6:]].float(), z[k].astype(np.float32), rtol=1e-4, atol=1e-5, what=k)


def test_nonfinite_scores_follow_the_reference_in_parity_mode(dev):
    """multihead_diffattn.py:95,106: the reference scales q first and passes the scores through torch.nan_to_num, so a forward whose
    q.k products overflow fp32 (+-inf / NaN scores) stays finite.  Parity (fp32) mode reproduces it inside the flash forward
    (cenet_attn_t.finite_scores: q pre-scaled in the staged tile, NaN -> 0 and +-inf -> +-FLT_MAX on every score before the online
    softmax): the reference-made fixture tests/golden/mod_diffattn_nonfinite.npz is matched like every other module golden.  The
    bf16 throughput kernels keep the documented behaviour ("Non-finite attention scores"): they do not hide such an
    input — the output is non-finite there."""
    from cenet_amd import kern
    z = np.load(os.path.join(GOLDEN, f"mod_{NONFINITE_CASE['name']}.npz"))
    m = build_product(NONFINITE_CASE, dev)
    sd = {k[3:]: torch.from_numpy(z[k]).to(dev) for k in z.files if k.startswith("sd.")}
    m.load_state_dict(sd, strict=True)
    m.eval()
    x = torch.from_numpy(z["in0"]).to(dev)
    assert np.isfinite(z["out_eval"]).all()
    with torch.no_grad():
        out = m(x)
    assert torch.isfinite(out).all()
    close(out, z["out_eval"], what="out_eval (non-finite scores)")
    old = kern.set_compute_bf16(True)
    try:
        with torch.no_grad():
            out_b = m(x.to(torch.bfloat16))
    finally:
        kern.set_compute_bf16(old)
    assert not torch.isfinite(out_b.float()).all()
