"""Per-stage launch SEQUENCE of one training step on the host SIMT checker (no GPU): every C-ABI call in issue order, tagged with
the SURVEY 8d stage whose forward / backward it belongs to.  The launch count per stage does not depend on the batch, so a
batch of 2 on the checker answers "which launches make up dec4's backward" in a minute.
    python tools/launch_census.py [-v] [stage ...]        (-v: the ordered list, else counts per entry point)"""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import bench  # noqa: E402
from backend import use_sim  # noqa: E402
from cenet_amd import kern, losses, optim  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("-v", action="store_true")
ap.add_argument("-B", type=int, default=2)
ap.add_argument("stages", nargs="*")
a = ap.parse_args()
dev = use_sim()
kern.set_compute_bf16(True)
net = bench.make_model(dev)
x, lab = bench.synthetic(a.B, dev, 0)
crit = losses.Criterion(4, argparse.Namespace(loss_type="dice,ce", loss_weights="0.5,0.5"))
arena = optim.ParamArena(net, optim.cenet_segments())
opt = optim.FusedSGD(arena, lr=0.01, momentum=0.9, weight_decay=1e-4)
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


def wrap(fn, label):
    def f(*args, **kw):
        log.append((cur[0], label(*args, **kw)))
        return fn(*args, **kw)
    return f


def _lab(name, *args):
    n = name.replace("cenet_", "")
    return n + " " + ",".join(str(v) for v in args if isinstance(v, int) and not isinstance(v, bool))[:60]


kern._call = wrap(kern._call, _lab)
kern.gemm = wrap(kern.gemm, lambda A, B, C, M, N, K, **kw: f"gemm M{M} N{N} K{K} nb{kw.get('nbatch', 1)} nkb{kw.get('nkb', 1)}"
                 f"{' at' if kw.get('atomic') else ''}")
kern.flash_fwd = wrap(kern.flash_fwd, lambda t, bf=False: f"flash_fwd Nq{t.Nq} Nk{t.Nk} D{t.D} H{t.H}")
kern.flash_bwd = wrap(kern.flash_bwd, lambda t, bf=False: f"flash_bwd Nq{t.Nq} Nk{t.Nk} D{t.D} H{t.H}")
kern.diffattn_heads = wrap(kern.diffattn_heads, lambda t, backward=False: f"dattn {'bwd' if backward else 'fwd'} N{t.N} hd{t.hd} H{t.H}")
kern.sra_attn_bwd = wrap(kern.sra_attn_bwd, lambda q, kv, o, g, lse, dq, dkv, B, H, Nq, Nk, sc: f"sra_bwd Nq{Nq} Nk{Nk} H{H}")
kern.sra_attn_fwd = wrap(kern.sra_attn_fwd, lambda q, kv, o, lse, B, H, Nq, Nk, sc: f"sra_fwd Nq{Nq} Nk{Nk} H{H}")
kern.attn64 = wrap(kern.attn64, lambda t, backward=False: f"attn64 {'bwd' if backward else 'fwd'} N{t.N} H{t.H}")
kern.wgrad_group = wrap(kern.wgrad_group, lambda probs, *r, **k: f"wgrad_group {len(probs)} problems")
opt.zero_grad()
crit(net(x), lab).backward()
opt.step()
tot = collections.Counter(s for s, _ in log)
print("launch-ish calls:", len(log))
for s, n in tot.items():
    print(f"  {s:40s} {n}")
for s in a.stages:
    for d in ("fwd", "bwd"):
        key = [k for k in tot if k.endswith(f"{s} {d}")]
        for k in key:
            print(f"== {k}: {tot[k]}")
            if a.v:
                for st, l in log:
                    if st == k:
                        print("   ", l)
            else:
                for l, n in collections.Counter(l.split(" ")[0] for st, l in log if st == k).most_common():
                    print(f"    {n:3d} {l}")
