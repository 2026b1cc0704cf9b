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

def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for k in (5, 1):
    x = torch.randn(32, 1, 224, 224, device=dev).bfloat16()
    w = torch.randn(32, 1, k, k, device=dev) * 0.2
    y = torch.empty(32, 32, 224, 224, device=dev, dtype=torch.bfloat16)
    g = torch.randn_like(y)
    dw = torch.zeros_like(w)
    print("k", k, "fwd us", timeit(lambda: kern.conv_c1_fwd(x, w, y, 32, 32, 224, 224, k)), "wgrad us", timeit(lambda: kern.conv_c1_wgrad(x, g, dw, 32, 32, 224, 224, k)))
