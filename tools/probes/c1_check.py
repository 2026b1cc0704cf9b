import os, sys
sys.path.insert(0, os.getcwd())
import torch, torch.nn.functional as F
from cenet_amd import kern, ops
dev = torch.device("cuda:0")
for (B, Co, k, H, W) in ((4, 32, 5, 224, 224), (4, 32, 1, 224, 224), (2, 32, 5, 512, 512)):
    x = torch.randn(B, 1, H, W, device=dev).bfloat16()
    w = (torch.randn(Co, 1, k, k, device=dev) * 0.2)
    wp = torch.nn.Parameter(w.clone())
    y = ops.conv2d_nchw(x, wp, None, stride=1, pad=k // 2)
    g = torch.randn_like(y)
    y.backward(g)
    xr = x.float(); wr = w.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, padding=k // 2)
    yr.backward(g.float())
    rel = lambda a, b: ((a.float() - b.float()).norm() / b.float().norm()).item()
    print(B, Co, k, H, W, "fwd rel", rel(y, yr), "wgrad rel", rel(wp.grad, wr.grad), "cos", torch.nn.functional.cosine_similarity(wp.grad.flatten(), wr.grad.flatten(), dim=0).item())
