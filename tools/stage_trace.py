"""Per-stage launch census of one training step (B = 32, bf16): every C-ABI call (kern._call entries, GEMMs with the kernel
instance they dispatched to, attention entries) is bracketed with HIP events and tagged with the SURVEY §8d stage whose
forward or backward it belongs to (module hooks).  Prints, for the stages named on the command line (default: all), the
entries sorted by time.   python tools/stage_trace.py [dec4 block1 ...]"""
import argparse
import collections
import os
import sys

sys.path.insert(0, os.getcwd())
import torch

import bench
from cenet_amd import kern, losses, optim

dev = torch.device("cuda:0")
kern.set_compute_bf16(True)
net = bench.make_model(dev)
x, lab = bench.synthetic(32, dev, 0)
crit = losses.Criterion(4, argparse.Namespace(loss_type="dice,ce", loss_weights="0.5,0.5"))
arena = optim.ParamArena(net, optim.cenet_segments())
opt = optim.FusedSGD(arena, lr=0.01, momentum=0.9, weight_decay=1e-4)


def step():
    opt.zero_grad()
    crit(net(x), lab).backward()
    opt.step()


for _ in range(2):
    step()
torch.cuda.synchronize()
cur = ["other"]
log = []
mods = dict(net.named_modules())
for name in bench.STAGES:
    m = mods[name]
    for sub in (list(m) if isinstance(m, (torch.nn.ModuleList, torch.nn.Sequential)) else [m]):
        sub.register_forward_pre_hook(lambda mod, inp, n=name: cur.__setitem__(0, n + " fwd"))
        sub.register_forward_hook(lambda mod, inp, out, n=name: cur.__setitem__(0, "other"))
        sub.register_full_backward_pre_hook(lambda mod, g, n=name: cur.__setitem__(0, n + " bwd"))
        sub.register_full_backward_hook(lambda mod, gi, go, n=name: cur.__setitem__(0, "other"))


def bracket(fn, label):
    def f(*a, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*a, **kw)
        e1.record()
        log.append((cur[0], label(*a, **kw), e0, e1))
        return r
    return f


def _lab(name, *a):
    n = name.replace("cenet_", "").replace("_f32", "")
    if os.environ.get("TRACE_SHAPES") and (os.environ.get("TRACE_ALL") or n.startswith(("dwconv3x3", "bn_", "bilinear", "copy_batched", "layernorm", "srm", "diffattn", "dseb", "adaptive", "gate", "ccu", "scale", "patch", "mix", "chan"))):
        n += " " + ",".join(str(v) for v in a if isinstance(v, int) and not isinstance(v, bool))
    return n


kern._call = bracket(kern._call, _lab)
kern.gemm = bracket(kern.gemm, lambda A, B, C, M, N, K, **kw: f"gemm {kern.last_gemm_kernel()} M{M} N{N} K{K} nb{kw.get('nbatch', 1)}"
                    f" nkb{kw.get('nkb', 1)}{' at' if kw.get('atomic') else ''}")
kern.flash_fwd = bracket(kern.flash_fwd, lambda a, bf=False: f"flash_fwd Nq{a.Nq} Nk{a.Nk} D{a.D} H{a.H}")
kern.flash_bwd = bracket(kern.flash_bwd, lambda a, bf=False: f"flash_bwd Nq{a.Nq} Nk{a.Nk} D{a.D} H{a.H}")
kern.diffattn_heads = bracket(kern.diffattn_heads, lambda a, backward=False: f"dattn {'bwd' if backward else 'fwd'} N{a.N} hd{a.hd} H{a.H}")
kern.sra_attn_bwd = bracket(kern.sra_attn_bwd, lambda q, kv, o, g, lse, dq, dkv, B, H, Nq, Nk, sc: f"sra_bwd Nq{Nq} Nk{Nk} H{H}")
kern.attn64 = bracket(kern.attn64, lambda a, backward=False: f"attn64 {'bwd' if backward else 'fwd'} N{a.N} H{a.H}")
step()
torch.cuda.synchronize()
want = sys.argv[1:]
st = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
tot = collections.defaultdict(lambda: [0, 0.0])
for stage, label, e0, e1 in log:
    ms = e0.elapsed_time(e1)
    st[stage][label][0] += 1
    st[stage][label][1] += ms
    tot[stage][0] += 1
    tot[stage][1] += ms
print(f"all bracketed calls: {sum(v[0] for v in tot.values())} calls, {sum(v[1] for v in tot.values()):.3f} ms (single stream)")
fam = collections.defaultdict(lambda: [0, 0.0])
for stage, label, e0, e1 in log:
    k = label.split(" ")[0]
    fam[k][0] += 1
    fam[k][1] += e0.elapsed_time(e1)
for k, (n, ms) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"   {ms:7.3f} ms {n:4d}x {ms / n * 1e3:7.1f} us  {k}")
if os.environ.get("TRACE_ALL"):  # every distinct call (with shapes) over the whole step, largest first
    lab = collections.defaultdict(lambda: [0, 0.0])
    for stage, label, e0, e1 in log:
        lab[label][0] += 1
        lab[label][1] += e0.elapsed_time(e1)
    print("-- by call and shape")
    for k, (n, ms) in sorted(lab.items(), key=lambda kv: -kv[1][1])[:int(os.environ["TRACE_ALL"])]:
        print(f"   {ms:7.3f} ms {n:4d}x {ms / n * 1e3:7.1f} us  {k}")
print(f"{'stage':36s} {'calls':>6s} {'ms':>8s}")
for stage, (n, ms) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"{stage:36s} {n:6d} {ms:8.3f}")
for stage in sorted(st):
    if want and not any(w in stage for w in want):
        continue
    if not want:
        continue
    print(f"\n== {stage}: {tot[stage][0]} calls, {tot[stage][1]:.3f} ms")
    for label, (n, ms) in sorted(st[stage].items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"  {ms:7.3f} ms {n:4d}x {ms / n * 1e3:7.1f} us  {label}")
