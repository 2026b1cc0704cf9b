"""Spatial-reduction attention with MORE THAN 64 keys (pvtv2.py:92-105 at 512x512 inputs: 256 keys of head dimension 64 under
4 096 .. 16 384 queries, BASELINE config 5): the backward runs as dQ over the resident keys plus dK / dV per 64-key block
(attn_diff.hip: sra_bwd_kernel<4, true, false> / <1, false, true>; the saved log-sum-exp makes the key blocks independent).
Ragged query tiles and key blocks, against fp32 PyTorch on the same bf16-rounded inputs; `sim` runs the same kernel source on
the host SIMT checker (CPU), `hip` on the GPU.  (Full-size case: tests/test_attention_presets.py, 16 384 queries x 256 keys.)"""
import pytest
import torch

from cenet_amd import ops


def _rel(a, b):
    return float((a.detach().float() - b.detach().float()).norm() / (b.detach().float().norm() + 1e-12))


def _bf(*shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g).bfloat16()


@pytest.mark.parametrize("backend", [pytest.param("sim"), pytest.param("hip", marks=pytest.mark.gpu)])
@pytest.mark.parametrize("N,heads,Nk", [(200, 2, 150), (97, 1, 256), (130, 1, 65)])
def test_sr_attention_more_than_64_keys(backend, N, heads, Nk):
    """65 .. 256 keys (pvtv2.py:92-105 at 512x512 inputs: 256 keys): the backward runs as dQ over the resident keys plus dK / dV
    per 64-key block (sra_bwd_kernel<4, true, false> / <1, false, true>); ragged query tiles and key blocks; vs fp32 PyTorch.
    `sim` runs the same kernels on the host SIMT checker."""
    from backend import use_hip, use_sim
    from cenet_amd import _lib
    dev = use_sim() if backend == "sim" else use_hip()
    try:
        B, Cn = 2, 64 * heads
        q, kv = _bf(B, N, Cn, seed=N).to(dev), _bf(B, Nk, 2 * Cn, seed=N + 1).to(dev)
        go = _bf(B, N, Cn, seed=3).to(dev)
        qs, kvs = q.clone().requires_grad_(True), kv.clone().requires_grad_(True)
        o = ops.sr_attention(qs, kvs, heads)
        o.backward(go)
        qf, kvf = q.float().cpu().requires_grad_(True), kv.float().cpu().requires_grad_(True)
        qh = qf.view(B, N, heads, 64).transpose(1, 2)
        kh = kvf[..., :Cn].reshape(B, Nk, heads, 64).transpose(1, 2)
        vh = kvf[..., Cn:].reshape(B, Nk, heads, 64).transpose(1, 2)
        ref = (torch.softmax(qh @ kh.transpose(-1, -2) * 64 ** -0.5, dim=-1) @ vh).transpose(1, 2).reshape(B, N, Cn)
        ref.backward(go.float().cpu())
        assert _rel(o.cpu(), ref) < 2e-2
        assert _rel(qs.grad.cpu(), qf.grad) < 3e-2 and _rel(kvs.grad.cpu(), kvf.grad) < 3e-2
    finally:
        _lib._LIB, _lib._HOSTSIM = None, False
