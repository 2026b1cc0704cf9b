"""Margins of tests/test_training_pin.py's trajectory test over a few repetitions (run-to-run noise of the bf16 product:
float atomics).   python tools/pin_margins.py [reps]"""
import os
import sys

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import torch

import test_training_pin as T
from backend import use_hip

z = T.W.golden("acdc")
dev = use_hip()
l32, d32, dp32, arena = T._product_run(z, dev, False, T.STEPS)
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    l16, d16, dp16, _ = T._product_run(z, dev, True, T.STEPS)
    worst = max(abs(a - b) / abs(b) for a, b in zip(l16, l32))
    cos = {name: torch.nn.functional.cosine_similarity(dp16[s:e].double(), dp32[s:e].double(), dim=0).item() for name, s, e in arena.segments}
    print(f"rep {rep}: worst loss rel {worst:.4f}  |dice16 - dice32| {abs(d16 - d32):.5f}  cos " + " ".join(f"{k}={v:.4f}" for k, v in cos.items()), flush=True)
