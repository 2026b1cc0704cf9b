"""Fused PVT-MLP kernels against the chain of launches they replace, at the stage-1 / stage-2 shapes of the benched step
(B = 32): HIP-event times of forward and forward+backward.   python tools/mlp_bench.py"""
import os
import sys

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch

from cenet_amd import kern, ops
from test_pvt_mlp import _chain, _params

dev = torch.device("cuda:0")
BF = torch.bfloat16


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for C, HD, H, W in [(64, 512, 56, 56), (128, 1024, 28, 28)][:int(os.environ.get("MLP_NCFG", "2"))]:
    B = 32
    p = _params(C, HD, dev)
    x = (torch.randn(B, H * W, C, device=dev) * 1.5).to(BF)
    bscale = (torch.rand(B, device=dev) < 0.9).float() / 0.9
    y = torch.empty_like(x)
    w1, w2 = kern.wq(p["w1"], x), kern.wq(p["w2"], x)

    saved = (torch.empty_like(x), torch.empty(B * H * W, device=dev), torch.empty(B * H * W, device=dev),
             torch.empty(B, H * W, HD, dtype=BF, device=dev), torch.empty(B, H * W, HD, dtype=BF, device=dev))

    def fused():
        kern.pvt_mlp_fwd(x, p["ln_g"], p["ln_b"], 1e-6, w1, p["b1"], p["wd"], p["bd"], w2, p["b2"], bscale, y, B, H, W, C, HD)

    def fused_saving():
        kern.pvt_mlp_fwd(x, p["ln_g"], p["ln_b"], 1e-6, w1, p["b1"], p["wd"], p["bd"], w2, p["b2"], bscale, y, B, H, W, C, HD, saved)

    def chain():
        with torch.no_grad():
            return _chain(x, p, H, W, bscale)

    ref = chain()
    fused()
    d = (y.float() - ref.float()).abs()
    print(f"C{C} HD{HD} {H}x{W}: max|d| {d.max().item():.4f} mean|d| {d.mean().item():.2e} (|ref| mean {ref.float().abs().mean().item():.3f})")
    print(f"   forward: fused {timeit(fused):8.1f} us   fused + saved tensors {timeit(fused_saving):8.1f} us   chain {timeit(chain):8.1f} us")
    gy0 = torch.randn_like(x)
    gu, dh, dx = torch.empty_like(saved[3]), torch.empty_like(saved[3]), torch.empty_like(x)
    gr = [torch.zeros_like(p[k]) for k in ("wd", "bd", "ln_g", "ln_b")]
    db2 = torch.zeros_like(p["b2"])
    fused_saving()

    def bwd_raw():
        kern.pvt_mlp_bwd(gy0, bscale, w1, w2, p["wd"], p["bd"], saved[3], x, p["ln_g"], saved[1], saved[2], gu, dh, dx,
                         gr[0], gr[1], gr[2], gr[3], db2, B, H, W, C, HD)

    print(f"   backward kernels (K1 + K2 + fold) alone: {timeit(bwd_raw):8.1f} us")
    if os.environ.get("MLP_RAW_ONLY"):
        continue
    if hasattr(ops, "pvt_mlp"):
        xg = x.clone().requires_grad_(True)
        gy = torch.randn_like(x)

        def fb_fused():
            out = ops.pvt_mlp(xg, H, W, p["ln_g"], p["ln_b"], 1e-6, p["w1"], p["b1"], p["wd"], p["bd"], p["w2"], p["b2"], bscale)
            out.backward(gy)
            ops.wgrad_flush()

        def fb_chain():
            out = _chain(xg, p, H, W, bscale)
            out.backward(gy)
            ops.wgrad_flush()

        print(f"   fwd+bwd: fused {timeit(fb_fused):8.1f} us   chain {timeit(fb_chain):8.1f} us")
