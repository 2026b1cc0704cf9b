"""FETCH_SIZE calibration for dword-per-lane streaming reads: copies a known number of bytes with copy_batched_kernel."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from cenet_amd import kern
dev = torch.device("cuda:0")
n = 128 * 1024 * 1024  # 512 MiB of fp32
x = torch.randn(n, device=dev); y = torch.empty_like(x)
for _ in range(3):
    kern.copy_batched(x, 0, y, 0, 1, n)
torch.cuda.synchronize()
print("bytes read per launch:", n * 4)
