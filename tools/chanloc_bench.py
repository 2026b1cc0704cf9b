"""Drives the channel-local fused kernels at the decoder shapes of the ACDC preset (B = 32) a few times each — a target for
rocprofv3 kernel traces / SQ counter passes:   python tools/chanloc_bench.py [reps]"""
import sys

import torch

sys.path.insert(0, ".")
from cenet_amd import kern, ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda:0")
BF = torch.bfloat16
torch.manual_seed(0)
for (C, H) in ((512, 7), (320, 14), (128, 28)):
    x = torch.randn(32, C, H, H, device=dev).to(BF).requires_grad_(True)
    w = (0.3 * torch.randn(C, 1, 3, 3, device=dev)).requires_grad_(True)
    gamma, beta = torch.ones(C, device=dev).requires_grad_(True), torch.zeros(C, device=dev).requires_grad_(True)
    rm, rv, nbt = torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.zeros((), dtype=torch.long, device=dev)
    cot = torch.randn(32, C, 2 * H, 2 * H, device=dev).to(BF)
    for p in (w, gamma, beta):
        p.grad = torch.zeros_like(p)
    for _ in range(reps):
        y = ops.eucb_front(x, w, gamma, beta, rm, rv, nbt, 1e-5, 0.2, 0.1)
        y.backward(cot)
torch.cuda.synchronize()
print("ok")
