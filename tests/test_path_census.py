"""Path census: WHICH kernels a preset's training step takes at its BASELINE batch (bf16 throughput mode, the benched configuration).

The fused fast paths of cenet_amd/ops/ are shape-gated (`pvt_mlp_supported`, `split_dwconv_bn_supported`, the channel-local
gates `cfam_front_supported` / `cfam_mid_supported` / `eucb_front_supported` / `pool_branch_supported`, `res_tail_*_supported`,
`kern.diffattn_heads_supported` ...): a gate that silently stops matching (a refactor, a changed default batch, an alignment
change) would send a stage back to its launch chain and nothing but the bench would notice.  Here one training step per preset runs
with every C-ABI call counted by entry point (kern._call, plus the attention / GEMM entries that have their own wrappers), and the
counts of the fast-path entries are held to the frozen census below; entries of the launch chains those paths replace must NOT appear
where the fused path is expected.  The census was taken on the MI355X with `python tests/test_path_census.py` (prints the table).
"""
import argparse
import collections
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def census(preset: str):
    import bench
    from cenet_amd import kern, losses, optim
    dev = torch.device("cuda:0")
    cfg = bench.CONFIGS[preset]
    old = kern.set_compute_bf16(True)
    counts = collections.Counter()
    saved = {}

    def count(name, fn, label):
        saved[name] = fn

        def f(*a, **k):
            counts[label(*a, **k)] += 1
            return fn(*a, **k)
        setattr(kern, name, f)

    try:
        net = bench.make_model(dev, cfg)
        x, lab = bench.synthetic(cfg["batch"], dev, 0, cfg)
        crit = losses.Criterion(cfg["classes"], argparse.Namespace(loss_type="dice,ce", loss_weights="0.5,0.5"))
        arena = optim.ParamArena(net, optim.cenet_segments())
        opt = optim.FusedSGD(arena, lr=0.01, momentum=0.9, weight_decay=1e-4)

        def step():
            opt.zero_grad()
            crit(net(x), lab).backward()
            opt.step()

        step()  # (first step: lazy buffers, merged parameters)
        torch.cuda.synchronize()
        count("_call", kern._call, lambda name, *a: name.replace("cenet_", ""))
        count("gemm", kern.gemm, lambda *a, **k: "gemm")
        count("diffattn_heads", kern.diffattn_heads, lambda a, backward=False: f"dattn_pairs_{'bwd' if backward else 'fwd'}_hd{a.hd}")
        count("attn64", kern.attn64, lambda a, backward=False: f"attn64_{'bwd' if backward else 'fwd'}")
        count("flash_fwd", kern.flash_fwd, lambda a, bf=False: "flash_fwd")
        count("flash_bwd", kern.flash_bwd, lambda a, bf=False: "flash_bwd")
        count("sra_attn_bwd", kern.sra_attn_bwd, lambda *a, **k: "sra_bwd_resident")
        step()
        torch.cuda.synchronize()
    finally:
        for name, fn in saved.items():
            setattr(kern, name, fn)
        kern.set_compute_bf16(old)
    return counts


# entry point -> launches per training step, as counted on the MI355X at the preset's BASELINE batch (names as kern._call sees them:
# the `_f32` twin name also stands for its bf16 twin).  Fast paths must be taken exactly this often; `TOTAL` bounds the number of
# C-ABI calls of the step from above (a fused path that falls back to its launch chain adds calls).
# ACDC / Synapse (224 x 224): stages at 56 / 28 / 14 / 7; HAM (512 x 512): 128 / 64 / 32 / 16.
_COMMON_224 = {
    "pvt_mlp_fwd_bf16": 7, "pvt_mlp_bwd_bf16": 7,              # stage 1 + 2 MLP halves (3 + 4 blocks): ops.pvt_mlp_supported
    "eucb_fwd_f32": 3, "eucb_bwd_acc_f32": 3,                  # up3 / up2 / up1 up to their 1x1 conv: ops.eucb_front_supported
    "cfam_front_fwd_f32": 2, "cfam_front_bwd_acc_f32": 2,      # dec4 (7 x 7), dec3 (14 x 14): channel-local gates, B H W <= 8192
    "cfam_mid_fwd_f32": 2, "cfam_mid_bwd_acc_f32": 2,
    "dwbn_fwd_f32": 2, "dwbn_bwd_acc_f32": 2,                  # ops.split_dwconv_bn_supported
    "dwact_fwd_f32": 2, "dwact_bwd_acc_f32": 2,
    "pool_branch_fwd_f32": 4, "pool_branch_bwd_acc_f32": 4,    # every decoder level: ops.pool_branch_supported
    "srm_conv_gelu_fwd_f32": 4, "srm_conv_bn_bwd_acc_f32": 4,  # fused SRM tail at every level
    "res_tail_img_fwd_bf16": 1, "res_tail_img_bwd_bf16": 1,    # the head's image branch: ops.res_tail_img_pool_supported
    "attn64_fwd": 2, "attn64_bwd": 2,                          # Non-local blocks of dec1 / dec2 on the single-softmax pair kernels
    "layernorm_bwd_add_part_bf16": 28, "layernorm_bwd_add_part_scaled_bf16": 18,  # LayerNorm backward on partial rows (no atomics)
    "scale_batch_f32": 0,                                      # DropPath scales ride in the producers (ops._prescaled_put)
    "bn_apply_f32": 0, "nearest2x_fwd_f32": 0, "adaptive_avgpool_fwd_f32": 0,  # launch-chain entries the fused paths replace
}
EXPECT = {
    "acdc": dict(_COMMON_224, **{
        "dattn_pairs_fwd_hd16": 1, "dattn_pairs_bwd_hd16": 1,  # DSEB1 (56 x 56, 2C = 128, 4 heads)
        "dattn_pairs_fwd_hd32": 1, "dattn_pairs_bwd_hd32": 1,  # DSEB2
        "flash_fwd": 0, "flash_bwd": 0, "softmax_rows_fwd_f32": 4,  # DSEB3 (head dim 80) + Non-local 14 x 14 / 7 x 7: materialised
        "TOTAL": 662}),
    "synapse": dict(_COMMON_224, **{
        "dattn_pairs_fwd_hd8": 1, "dattn_pairs_bwd_hd8": 1,    # DSEB1 (16 heads)
        "dattn_pairs_fwd_hd16": 1, "dattn_pairs_bwd_hd16": 1,  # DSEB2 (8 heads)
        "flash_fwd": 1, "flash_bwd": 1,                        # DSEB3 (head dim 40: tiled kernels)
        "TOTAL": 665}),
    "ham512": {
        "eucb_fwd_f32": 2, "eucb_bwd_acc_f32": 2, "nearest2x_fwd_f32": 1,   # up1 at 128 x 128 -> 256 x 256 exceeds the fused kernel's plane
        "pool_branch_fwd_f32": 3, "pool_branch_bwd_acc_f32": 3,
        "cfam_front_fwd_f32": 1, "cfam_mid_fwd_f32": 1, "dwbn_fwd_f32": 1, "dwact_fwd_f32": 1,  # dec4 at 16 x 16 only (B H W = 2048)
        "dattn_pairs_fwd_hd32": 1, "dattn_pairs_bwd_hd32": 1, "dattn_pairs_fwd_hd64": 1, "dattn_pairs_bwd_hd64": 1,
        "attn64_fwd": 2, "attn64_bwd": 2,
        "sra_bwd_resident": 16, "flash_fwd": 16,               # 256 keys under 4 096 ... 16 384 queries: resident-key backward
        "res_tail_fwd_bf16": 1, "res_tail_bwd_bf16": 1,         # three-channel input: the non-image form of the head's tail
        "TOTAL": 804},
}


@pytest.mark.gpu
@pytest.mark.parametrize("preset", ["acdc", "synapse", "ham512"])
def test_fast_paths_taken_at_the_baseline_batch(preset):
    got = census(preset)
    exp = dict(EXPECT[preset])
    total = exp.pop("TOTAL")
    bad = {k: (got.get(k, 0), v) for k, v in exp.items() if got.get(k, 0) != v}
    assert not bad, f"{preset}: (taken, expected) per entry point: {bad}"
    assert sum(got.values()) <= total, f"{preset}: {sum(got.values())} C-ABI calls per step, census {total}"


if __name__ == "__main__":
    for p in sys.argv[1:] or ["acdc", "synapse", "ham512"]:
        c = census(p)
        print(f"== {p}: {sum(c.values())} calls")
        for k, v in sorted(c.items()):
            print(f"   {v:4d}  {k}")
