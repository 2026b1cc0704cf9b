"""ctypes loader for libcenet_hip.so — the hand-written gfx950 kernel library.

There is NO fallback: if the library is missing (or a tensor is not on the GPU) the product path raises.
(`tests/hostsim` may point `_LIB` at a g++ build of the same kernel sources running on a SIMT interpreter to
check kernel logic without a GPU; nothing in the package does that.)
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcenet_hip.so")

_LIB = None
_HOSTSIM = False  # set only by tests/hostsim


class MatT(C.Structure):
    """cenet_mat_t (include/cenet_hip.h)."""
    _fields_ = [("ptr", C.c_void_p), ("sb", C.c_long), ("sb2", C.c_long), ("skb", C.c_long), ("sr", C.c_long),
                ("sc", C.c_long), ("sk_outer", C.c_long), ("kinner", C.c_int), ("mode", C.c_int), ("kfast", C.c_int), ("patch_is_row", C.c_int), ("transposed", C.c_int),
                ("KH", C.c_int), ("KW", C.c_int), ("Pw", C.c_int), ("Hs", C.c_int), ("Ws", C.c_int),
                ("stride", C.c_int), ("pad", C.c_int), ("dil", C.c_int),
                ("sci", C.c_long), ("sy", C.c_long), ("sx", C.c_long)]


class EpiT(C.Structure):
    """cenet_epi_t (include/cenet_hip.h)."""
    _fields_ = [("C", C.c_void_p), ("scb", C.c_long), ("scb2", C.c_long), ("scr", C.c_long), ("scc", C.c_long),
                ("bias", C.c_void_p), ("bias_on_row", C.c_int), ("act", C.c_int), ("slope", C.c_float),
                ("bscale", C.c_void_p), ("R", C.c_void_p), ("srb", C.c_long), ("srb2", C.c_long), ("srr", C.c_long), ("src", C.c_long),
                ("atomic", C.c_int), ("alpha", C.c_float),
                ("cmode", C.c_int), ("cKH", C.c_int), ("cKW", C.c_int), ("cPw", C.c_int), ("cHs", C.c_int),
                ("cWs", C.c_int), ("cstride", C.c_int), ("cpad", C.c_int),
                ("csci", C.c_long), ("csy", C.c_long), ("csx", C.c_long), ("bscale_rows", C.c_int), ("asum", C.c_void_p)]


def lib():
    """Return the loaded kernel library, loading it on first use. Raises if it has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP kernel library is required (no CPU/PyTorch fallback exists). "
                "Build it with `python -m cenet_amd.build` (hipcc --offload-arch=gfx950).")
        _LIB = C.CDLL(LIB_PATH)
    return _LIB


def is_hostsim() -> bool:
    return _HOSTSIM


def check(rc: int, name: str):
    if rc != 0:
        raise RuntimeError(f"libcenet_hip: {name} failed with error code {rc}")
