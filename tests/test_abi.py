"""The C-ABI library loads and exports every entry point that include/cenet_hip.h declares (no compute, no GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "cenet_hip.h")).read()
    return sorted(set(re.findall(r"\b(?:int|const char\*)\s+(cenet_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_entry_points():
    syms = declared_symbols()
    assert len(syms) >= 55
    assert "cenet_gemm_f32" in syms and "cenet_flash_attn_fwd_f32" in syms and "cenet_sgd_step_f32" in syms


def test_hip_library_exports_every_declared_symbol():
    from cenet_amd import build
    lib_path = build.build_hip(verbose=False)  # hipcc cross-compiles for gfx950 without a GPU
    lib = ctypes.CDLL(lib_path)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, f"libcenet_hip.so does not export: {missing}"


def test_product_path_refuses_cpu_tensors():
    """No CPU fallback: the product ops raise on host tensors instead of silently computing elsewhere."""
    import torch
    from cenet_amd import _lib, ops
    assert not _lib.is_hostsim()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.linear(torch.randn(4, 8), torch.randn(3, 8))


def test_missing_library_fails_loudly(monkeypatch):
    from cenet_amd import _lib
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libcenet_hip.so")
    monkeypatch.setattr(_lib, "_LIB", None)
    with pytest.raises(RuntimeError, match="no CPU/PyTorch fallback"):
        _lib.lib()
