"""Micro-benchmark of the attention problems of one CENet step (B=32, bf16 mode): fwd and bwd time per call."""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch

from cenet_amd import kern, ops

dev = torch.device("cuda:0")
kern.set_compute_bf16((sys.argv[1] if len(sys.argv) > 1 else "bf16") == "bf16")


def t(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


B = 32
print("differential attention (DSEB): N, heads, hd -> fwd ms, bwd ms")
for N, H, hd in [(3136, 4, 16), (784, 4, 32), (196, 4, 80)]:
    E = 2 * H * hd
    q, k, v = (torch.randn(B, N, E, device=dev, requires_grad=True) for _ in range(3))
    if not kern.flash_supported(hd, 2 * hd):
        print(N, H, hd, "materialised path"); continue
    U = ops.diff_attention_heads(q, k, v, H)
    g = torch.randn_like(U)
    tf = t(lambda: ops.diff_attention_heads(q.detach(), k.detach(), v.detach(), H))
    tfb = t(lambda: ops.diff_attention_heads(q, k, v, H).backward(g))
    print(f"  N={N} H={H} hd={hd}: fwd {tf:.3f}  bwd {tfb - tf:.3f}")
print("spatial-reduction attention (PVT): N, Nk, C, heads")
for N, C, heads in [(3136, 64, 1), (784, 128, 2), (196, 320, 5), (49, 512, 8)]:
    q = torch.randn(B, N, C, device=dev, requires_grad=True)
    kv = torch.randn(B, 49, 2 * C, device=dev, requires_grad=True)
    o = ops.sr_attention(q, kv, heads)
    g = torch.randn_like(o)
    tf = t(lambda: ops.sr_attention(q.detach(), kv.detach(), heads))
    tfb = t(lambda: ops.sr_attention(q, kv, heads).backward(g))
    print(f"  N={N} C={C} heads={heads}: fwd {tf:.3f}  bwd {tfb - tf:.3f}")
print("non-local attention: C, N")
for C, N in [(64, 3136), (128, 784), (320, 196)]:
    if not kern.flash_supported(C, C):
        print(" ", C, N, "materialised path"); continue
    th, ph, gx = (torch.randn(B, C, N, device=dev, requires_grad=True) for _ in range(3))
    y = ops.nonlocal_attention(th, ph, gx)
    g = torch.randn_like(y)
    tf = t(lambda: ops.nonlocal_attention(th.detach(), ph.detach(), gx.detach()))
    tfb = t(lambda: ops.nonlocal_attention(th, ph, gx).backward(g))
    print(f"  C={C} N={N}: fwd {tf:.3f}  bwd {tfb - tf:.3f}")
