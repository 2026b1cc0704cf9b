"""Test helper: run the same kernel tests either on the host SIMT checker ("sim", CPU) or on the real
MI355X library ("hip", marked gpu)."""
import ctypes
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "hostsim"))

from cenet_amd import _lib  # noqa: E402

_SIM = None


def use_sim():
    global _SIM
    if _SIM is None:
        from build_sim import build_sim
        # CENET_SIM_SANITIZE=1: the UndefinedBehaviorSanitizer build of the same sources (sanitizers run on the CPU build only)
        _SIM = ctypes.CDLL(build_sim(sanitize=os.environ.get("CENET_SIM_SANITIZE") == "1"))
    _lib._LIB = _SIM
    _lib._HOSTSIM = True
    return torch.device("cpu")


def use_hip():
    _lib._LIB = None
    _lib._HOSTSIM = False
    _lib.lib()
    return torch.device("cuda:0")


BACKENDS = [pytest.param("sim", id="sim"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]


@pytest.fixture(params=BACKENDS)
def dev(request):
    d = use_sim() if request.param == "sim" else use_hip()
    yield d
    _lib._LIB = None
    _lib._HOSTSIM = False
