"""bench.py — training throughput of the CENet hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W [--config acdc|synapse|ham512]
                                                   (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" = the reference's train-step body (src/main_acdc.py:237-257) on one synthetic batch already resident in
HBM: zero_grad -> CENet forward -> Dice+CE (0.5/0.5) -> backward -> gradient all-reduce (RCCL) -> fused SGD.
Default workload = BASELINE.json configs[1]: ACDC 224x224, 4 classes, batch 32 per GPU, random-init
PVTv2-b2 CENet.  --config synapse / ham512 run C4 / C5 of the same table.

Prints ONE JSON line on rank 0 (contract in the task statement).  After the timed region, rank 0 (N == 1) re-runs a few
INSTRUMENTED steps and adds:
  roofline         the kernel instance with the largest summed duration among the step's launches (every GEMM launch is
                   bracketed with HIP events on the stream it is launched on and grouped by the instance name the C-ABI
                   reports, spelled as rocprofv3 prints it; the attention kernels are bracketed per call): algorithmic
                   bytes and FLOPs of all its launches in one step / their summed duration, against the binding roof.
                   `traffic` = HBM bytes PER LAUNCH from the PMC passes recorded in profiles/r03_traffic.json (null when
                   that file has no entry for the kernel); `top_kernels` = the three largest groups, same fields.
  whole_step       the step's algorithmic FLOPs / bytes over the timed ms_per_step against the MFMA and HBM roofs.
  roofline_stages  every stage of SURVEY.md §8d (patch_embed1-4, block1-4, dec4..dec1, up3..up1, DSEB3..DSEB1, out.rb /
                   out.up / out.out): forward + backward time measured with HIP events in module hooks (weight gradients on
                   the main stream for this pass), against max(3 x fwd FLOPs / MFMA peak, 3 x boundary bytes / HBM peak).
  cpu_baseline     the oracle timed on the host cores.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: FP32 matrix peak (v_mfma_f32_16x16x4_f32)
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 MFMA (same guide)
PEAK_HBM_GBS = 8000.0

# SURVEY.md §8d C2 / C4 / C5 (the reference's flag presets: acdc.sh:41-77, synapse.sh:42-81, skin.sh:45-100)
CONFIGS = {
    "acdc": dict(name="ACDC 224x224 4-class", in_ch=1, classes=4, size=224, batch=32, heads=[4, 4, 4], scales=[1.0, 0.5]),
    "synapse": dict(name="Synapse 224x224 9-class", in_ch=1, classes=9, size=224, batch=24, heads=[16, 8, 8],
                    scales=[0.8, 0.4]),
    "ham512": dict(name="HAM10000 512x512 2-class", in_ch=3, classes=2, size=512, batch=8, heads=[2, 2, 2],
                   scales=[1.0, 0.75, 0.5]),
}

# SURVEY.md §8d per-stage algorithmic work per image at 224^2 (forward GFLOP; boundary elements in + out)
STAGES = {
    "backbone.patch_embed1": (0.059, 351232), "backbone.patch_embed2": (0.116, 301056),
    "backbone.patch_embed3": (0.145, 163072), "backbone.patch_embed4": (0.145, 87808),
    "backbone.block1": (1.671, 3 * 401408), "backbone.block2": (2.102, 4 * 200704),
    "backbone.block3": (2.871, 6 * 125440), "backbone.block4": (0.945, 3 * 50176),
    "decoder.dec4": (0.400, 50176), "decoder.up3": (0.066, 87808), "decoder.skip_enhancer3": (0.821, 188160),
    "decoder.dec3": (0.669, 125440), "decoder.up2": (0.069, 163072), "decoder.skip_enhancer2": (1.092, 301056),
    "decoder.dec2": (0.717, 200704), "decoder.up1": (0.059, 301056), "decoder.skip_enhancer1": (5.498, 602112),
    "decoder.dec1": (2.929, 401408), "out.rb": (2.653, 451584), "out.up": (0.462, 602112), "out.out": (1.856, 852992),
}


# of which N x N attention contractions (forward GFLOP per image at 224 x 224, the reference's operator list: DSEB 4 E N^2, Non-local
# 4 C N^2, spatial-reduction attention 4 C Nq Nk per block): these grow with the FOURTH power of the input size, the rest with its
# square (x27 against x5.2 at 512 x 512)
ATTN_GF = {
    "backbone.block1": 0.118, "backbone.block2": 0.0787, "backbone.block3": 0.0738, "backbone.block4": 0.0148,
    "decoder.dec4": 0.0049, "decoder.dec3": 0.049, "decoder.dec2": 0.315, "decoder.dec1": 2.517,
    "decoder.skip_enhancer3": 0.098, "decoder.skip_enhancer2": 0.629, "decoder.skip_enhancer1": 5.035,
}


def stage_gflop(name: str, size: int) -> float:
    """forward GFLOP per image of a SURVEY 8d stage at `size` x `size` input"""
    sf = (size / 224.0) ** 2
    ga = ATTN_GF.get(name, 0.0)
    return (STAGES[name][0] - ga) * sf + ga * sf * sf


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="acdc")
    ap.add_argument("--batch", type=int, default=0, help="images per GPU (default: the preset's 32 / 24 / 8)")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="bf16",
                    help="storage type of activations and MFMA operands (accumulation, statistics, parameters stay fp32)")
    ap.add_argument("--graph", choices=["auto", "on", "off"], default="auto",
                    help="replay the step as one captured hipGraph (auto: N == 1 times a few steps both ways and keeps the faster)")
    ap.add_argument("--dist-launch", choices=["auto", "eager", "split", "segmented"], default="auto",
                    help="N > 1: force one launch form of the step instead of timing the three (see main)")
    ap.add_argument("--no-f32", action="store_true", help="skip the extra fp32 parity-mode measurement (N=1, bf16 runs)")
    ap.add_argument("--no-overlap", action="store_true", help="keep the weight-gradient kernels on the main stream")
    ap.add_argument("--bf16-buckets", action="store_true", help="N > 1: all-reduce the gradient segments as bf16 (half the bytes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true", help="skip the instrumented passes (roofline, roofline_stages)")
    ap.add_argument("--cpu-steps", type=int, default=3)
    return ap.parse_args()


def make_model(dev, cfg=None):
    from cenet_amd.networks import CENet
    cfg = cfg or CONFIGS["acdc"]
    torch.manual_seed(1234)
    net = CENet(input_channels=cfg["in_ch"], num_classes=cfg["classes"], scale_factors=cfg["scales"],
                diffatt_num_heads=cfg["heads"], encoder="pvt_v2_b2", enc_pretrain=False, skip_mode="cat", dec_up_block="eucb",
                out_merge_mode="cat", out_up_block="upcn", out_up_ks=3)  # (the network is size-agnostic: 512^2 needs no flag)
    return net.to(dev).train()


def synthetic(B, dev, seed, cfg=None):
    cfg = cfg or CONFIGS["acdc"]
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, cfg["in_ch"], cfg["size"], cfg["size"], generator=g)
    lab = torch.randint(0, cfg["classes"], (B, cfg["size"], cfg["size"]), generator=g).float()
    return x.to(dev), lab.to(dev)


def peak_tflops():
    from cenet_amd import kern
    return PEAK_BF16_MFMA_TFLOPS if kern.get_compute_bf16() else PEAK_F32_MFMA_TFLOPS


# ---------------------------------------------------------------------------------------------------------------------------
# instrumented passes (after the timed region; never inside it)
# ---------------------------------------------------------------------------------------------------------------------------
class _Trace:
    """brackets every kern.gemm / attention-kernel call of a step with HIP events on the launching stream"""

    def __init__(self):
        self.rows = []  # (name, e0, e1, flops, bytes)
        self.valu = {}  # attention kernels: vector-issue cycles (per SIMD-lane-group, see work()) summed over the bracketed calls
        self.valu_cycles_last = 0.0

    def install(self):
        from cenet_amd import kern
        self._saved = {n: getattr(kern, n) for n in ("gemm", "diffattn_heads", "attn64", "flash_fwd", "flash_bwd", "wgrad_group")}
        tr = self

        def wgrad_group(probs, device, **kw):
            """the grouped weight gradients, one call per kernel launch — the partition is the LIBRARY's (kern.wgrad_group_plan:
            launch index and tile per problem), K-slice kernel and fold kernel bracketed separately (phases 1 / 2)"""
            plan = kern.wgrad_group_plan(probs)
            for li in sorted({p[0] for p in plan}):
                chunk = [t for t, p in zip(probs, plan) if p[0] == li]
                _, bm, bn, ns = next(p for p in plan if p[0] == li)
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
                ev[0].record()
                need = tr._saved["wgrad_group"](chunk, device, phases=(1, 2),
                                                between=lambda: (ev[1].record(), ev[2].record()))
                ev[3].record()
                fl = sum(2.0 * t[8] * t[9] * t[10] * t[11] for t in chunk)
                # operands once (bf16) + the fp32 gradient tile read and written once
                by = sum(2.0 * t[10] * t[11] * (t[8] + t[9]) + 8.0 * t[8] * t[9] for t in chunk)
                b = "true" if chunk[0][12] else "false"
                tr.rows.append((f"gemm_group_kernel<{b}, {b}, {bm}, {bn}, {ns}>", ev[0], ev[1], fl, by))
                if need:  # partial tiles written by the slices and read back by the fold
                    tr.rows.append((f"gemm_group_fold_kernel<{bm}, {bn}>", ev[2], ev[3], 0.0, 2.0 * 4.0 * need))

        def gemm(A, B, Cout, M, N, K, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            tr._saved["gemm"](A, B, Cout, M, N, K, **kw)
            e1.record()
            nb, nkb = kw.get("nbatch", 1), kw.get("nkb", 1)
            es = 2 if A.bf16 else 4
            ce = 4 if kw.get("atomic") else es
            by = nb * nkb * (M * K + K * N) * es + nb * M * N * ce * (2 if kw.get("R") is not None else 1)
            tr.rows.append((kern.last_gemm_kernel(), e0, e1, 2.0 * M * N * K * nb * nkb, float(by), f"M{M} N{N} K{K} batch{nb * nkb}"))

        def work(a, backward):
            """algorithmic FLOPs / bytes of one attention call: QK^T and PV (2 MAC each) forward, 2.5x that backward;
            q, k, v, o once (plus their gradients backward), 2-byte elements"""
            if hasattr(a, "Nq"):
                fl = 2.0 * a.B * a.H * a.Nq * a.Nk * (a.D + a.Dv)
                el = a.B * a.H * (a.Nq * (a.D + a.Dv) + a.Nk * (a.D + a.Dv))
            elif a.hd == 64:  # single-softmax form: H heads of dimension 64
                fl = 2.0 * a.B * a.H * a.N * a.N * 128
                el = a.B * a.H * a.N * 64 * 4
            else:  # differential pairs: 2H softmax heads of dim hd over H value heads of dim 2 hd
                fl = 2.0 * a.B * 2 * a.H * a.N * a.N * 3 * a.hd
                el = a.B * a.N * (2 * a.H * a.hd * 2 + a.H * 2 * a.hd + 2 * a.H * 2 * a.hd)
            # vector-instruction ISSUE cycles per softmax score (MI355X_MICROARCH.md issue costs: 4 per wave64 VALU instruction, 8
            # per transcendental; 64 scores per instruction).  Forward: scale / subtract (fma 4) + exp (8) + row-sum add (4) + half
            # a max3 (2) + half a cvt_pk (2) = 20.  Backward, per kernel (dQ and dK/dV each recompute p): fma + exp (12) +
            # dS = p (dP - delta) (8) + two half cvt_pk (4) = 24, two kernels = 48.
            scores = a.B * a.H * a.Nq * a.Nk if hasattr(a, "Nq") else a.B * a.H * a.N * a.N * (1 if a.hd == 64 else 2)
            tr.valu_cycles_last = scores * (48.0 if backward else 20.0) / 64.0
            return (2.5 * fl, 2.0 * el * 2) if backward else (fl, el * 2.0)

        def wrap(name, label, backward):
            def f(*a, **kw):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                r = tr._saved[name](*a, **kw)
                e1.record()
                bw = backward if backward is not None else bool(kw.get("backward", a[1] if len(a) > 1 else False))
                fl, by = work(a[0], bw)
                lab = label(*a, **kw)
                tr.rows.append((lab, e0, e1, fl, by))
                tr.valu[lab] = tr.valu.get(lab, 0.0) + tr.valu_cycles_last
                return r
            return f
        kern.gemm = gemm
        kern.wgrad_group = wgrad_group
        kern.diffattn_heads = wrap("diffattn_heads", lambda a, backward=False: "dattn_bwd_dq+dkv_kernel" if backward else "dattn_fwd_kernel", None)
        kern.attn64 = wrap("attn64", lambda a, backward=False: "dattn_bwd_dq+dkv_kernel<32, true>" if backward else "dattn_fwd_kernel<32, 1, 2, true>", None)
        kern.flash_fwd = wrap("flash_fwd", lambda a, bf=False: "flashc_fwd_kernel" if bf else "flash_fwd_kernel", False)
        kern.flash_bwd = wrap("flash_bwd", lambda a, bf=False: "flashc_bwd_dq+dkv_kernel" if bf else "flash_bwd_dq+dkv_kernel", True)

    def remove(self):
        from cenet_amd import kern
        for n, f in self._saved.items():
            setattr(kern, n, f)

    @staticmethod
    def bracket_overhead_ms(n=200):
        """what an EMPTY event bracket reads on the parked stream (marker-to-marker dispatch latency).  Reported, NOT subtracted:
        a bracketed kernel hides part of it (the live average of the 17.6 us dominant kernel reads 20.7 us, the empty bracket
        4.6 us), so the live averages bound the profiler's kernel durations from above by at most this much"""
        _hold_stream(5.0)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for a, b in ev:
            a.record()
            b.record()
        torch.cuda.synchronize()
        v = sorted(a.elapsed_time(b) for a, b in ev)
        return v[len(v) // 2]

    def groups(self, steps):
        torch.cuda.synchronize()
        self.overhead_ms = self.bracket_overhead_ms()
        g = {}
        self.shapes = {}  # kernel instance -> problem shape -> [ms, launches, flops, bytes] per step (GEMM instances only)
        for row in self.rows:
            name, e0, e1, fl, by = row[:5]
            ms = e0.elapsed_time(e1) / steps
            t = g.setdefault(name, [0.0, 0, 0.0, 0.0])
            t[0] += ms
            t[1] += 1
            t[2] += fl / steps
            t[3] += by / steps
            if len(row) > 5:
                u = self.shapes.setdefault(name, {}).setdefault(row[5], [0.0, 0, 0.0, 0.0])
                u[0] += ms
                u[1] += 1
                u[2] += fl / steps
                u[3] += by / steps
        return g


HOLD_MS = [60.0]  # set from the measured eager step time (main): long enough for the host to queue a whole step


def _hold_stream(ms=None):
    """Parks the stream behind a spin kernel so that the host queues the whole instrumented step before the GPU starts it:
    the intervals between two events then are kernel durations (as rocprofv3 reports them), not host launch gaps."""
    torch.cuda.synchronize()
    ms = HOLD_MS[0] if ms is None else ms
    torch.cuda._sleep(int(ms * 1e-3 * 2.0e9))  # cycles of the ~2 GHz shader clock (shows up as `spin_kernel` in a profile)


def roofline_block(body, steps=2):
    """dominant kernel instance of the step (largest summed duration among the bracketed launches)"""
    from cenet_amd import ops
    tr = _Trace()
    tr.install()
    old = ops.set_wgrad_overlap(False)  # one stream: an interval then never contains a wait for the other stream
    try:
        for _ in range(steps):
            _hold_stream()
            body()
        g = tr.groups(steps)
    finally:
        tr.remove()
        ops.set_wgrad_overlap(old)
    # GEMM groups are exact kernel symbols; an attention-backward call launches two symbols (dQ and dK/dV kernels of similar
    # length) on one entry, so its group competes with half its time
    def weight(kv):
        return kv[1][0] * (0.5 if "+" in kv[0] else 1.0)
    peak = peak_tflops()
    traffic_rec = {}
    traffic_tag = None
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):  # the latest round's PMC passes
        try:
            traffic_rec = json.load(open(f))
            traffic_tag = os.path.basename(f).split("_")[0]
            break
        except (OSError, ValueError):
            continue

    meta = traffic_rec.pop("_meta", None) or {}
    from cenet_amd.build import source_sha16
    # the profiler's average kernel durations of the latest committed `rocprofv3 --kernel-trace --stats` summary of the bench
    # command (profiles/rNN_c_kernel_stats_noroofline.csv, tools/refresh_profiles.sh): name -> (average ms, file)
    prof_avg = {}
    import csv
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_c_kernel_stats_noroofline.csv")), reverse=True):
        try:
            for r in csv.DictReader(open(f)):
                nm = r["Name"].replace("void ", "").split("(")[0]
                prof_avg[nm] = (float(r["AverageNs"]) * 1e-6, os.path.basename(f))
            break
        except (OSError, ValueError, KeyError):
            continue
    # the committed summary describes THIS build only if it was taken on the same kernel sources: tools/refresh_profiles.sh writes
    # the PMC record (with the source hash in its _meta) and the kernel statistics in one run under one round tag.  A summary of
    # another build is reported beside the live numbers but never replaces them (`frac` must not go stale).
    prof_tag = next(iter(prof_avg.values()))[1].split("_")[0] if prof_avg else None
    prof_current = bool(prof_avg) and prof_tag == traffic_tag and meta.get("kernel_src_sha16") == source_sha16()

    def entry(name, ms, n, fl, by):
        n //= steps
        out = {"kernel": name, "launches_per_step": n, "avg_launch_ms": round(ms / max(n, 1), 5), "total_ms_per_step": round(ms, 3)}
        if fl > 0 or by > 0:
            t_mfma, t_hbm = fl / (peak * 1e12), by / (PEAK_HBM_GBS * 1e9)
            if t_mfma >= t_hbm:
                a = fl / (ms * 1e-3) / 1e12
                out.update({"bound": "mfma", "achieved": round(a, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(a / peak, 4)})
            else:
                a = by / (ms * 1e-3) / 1e9
                out.update({"bound": "hbm", "achieved": round(a, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                            "frac": round(a / PEAK_HBM_GBS, 4)})
            out["algorithmic_bytes_per_launch"] = int(by / max(n, 1))
            out["algorithmic_flops_per_launch"] = int(fl / max(n, 1))
            # `frac` above is the LIVE one (HIP-event intervals, which include up to one marker-to-marker latency).  Where the
            # committed profiler summary has this instance, the fraction from ITS average duration is reported too and is the one
            # that follows from profiles/: frac_rocprof = algorithmic work per launch / rocprof average / peak
            if name in prof_avg:
                pms, pfile = prof_avg[name]
                out["live_avg_launch_ms"] = out["avg_launch_ms"]
                out["rocprof_avg_launch_ms"] = round(pms, 5)
                out["rocprof_source"] = "profiles/" + pfile
                pa = (fl if out["bound"] == "mfma" else by) / max(n, 1) / (pms * 1e-3) / (1e12 if out["bound"] == "mfma" else 1e9)
                out["frac_live"] = out["frac"]
                out["achieved_live"] = out["achieved"]
                out["frac_rocprof"] = round(pa / out["peak"], 4)
                out["achieved_rocprof"] = round(pa, 2 if out["bound"] == "mfma" else 1)
                out["rocprof_is_this_build"] = prof_current
                if prof_current:  # `frac` follows from profiles/ only when profiles/ describes the kernels that just ran
                    out["achieved"], out["frac"] = out["achieved_rocprof"], out["frac_rocprof"]
                    out["frac_source"] = "rocprofv3 average duration (profiles/, same kernel sources)"
                else:
                    out["frac_source"] = "live HIP events (the committed profile is of other kernel sources)"
        if name in tr.valu:
            # the pair / flash attention kernels are bound by vector-instruction ISSUE, not by the matrix pipe (DESIGN.md section 8,
            # profiles/r04_attn_sq_counters.txt): graded against that roof too — issue cycles of the softmax algebra alone over
            # 1024 SIMDs at the 2.4 GHz maximum clock
            roof_ms = tr.valu[name] / steps / (1024 * 2.4e9) * 1e3
            out["valu_issue"] = {"bound": "valu", "roof_ms_per_step": round(roof_ms, 4), "frac": round(roof_ms / ms, 4),
                                 "peak": "1024 SIMDs x 1 wave64 vector instruction per 4 cycles (8 per transcendental) x 2.4 GHz"}
        rec = traffic_rec.get(name)
        # HBM bytes per LAUNCH from the PMC passes of tools/refresh_profiles.sh (2 x FETCH_SIZE + WRITE_SIZE, separate passes);
        # a committed measurement of this build on another box of the pool, not taken in this run
        out["traffic"] = rec["hbm_bytes_per_launch"] if rec else None
        if rec:
            out["traffic_unit"] = "bytes per launch"
            out["traffic_source"] = rec.get("source")
        return out
    ranked = sorted(g.items(), key=weight, reverse=True)
    name, (ms, n, fl, by) = ranked[0]
    out = entry(name, ms, n, fl, by)
    # the dominant instance serves many differently shaped problems: its work, time and (live) fraction per problem shape, so that
    # the headline fraction is not read as one problem's
    if name in tr.shapes:
        peak_v = out.get("peak", PEAK_HBM_GBS)
        per = []
        for shp, (sms, sn, sfl, sby) in sorted(tr.shapes[name].items(), key=lambda kv: -kv[1][0]):
            k = sn // steps
            a = ((sfl / 1e12) if out.get("bound") == "mfma" else (sby / 1e9)) / (sms * 1e-3)
            per.append({"shape": shp, "launches_per_step": k, "live_avg_launch_ms": round(sms / max(k, 1), 5),
                        "algorithmic_bytes_per_launch": int(sby / max(k, 1)), "algorithmic_flops_per_launch": int(sfl / max(k, 1)),
                        "frac_live": round(a / peak_v, 4)})
        out["shapes"] = per
    out["method"] = ("HIP events around every launch of this instance in %d instrumented steps (one stream, the stream parked behind a "
                     "spin kernel while the host queues the step, so intervals are kernel durations plus at most the "
                     "marker-to-marker latency an empty bracket reads: %.4f ms)" % (steps, tr.overhead_ms))
    out["empty_bracket_ms"] = round(tr.overhead_ms, 5)
    # the PMC passes were taken on the kernel sources that hash to traffic_src_sha16; kernel_src_sha16 = the sources benched now
    out["kernel_src_sha16"], out["traffic_src_sha16"] = source_sha16(), meta.get("kernel_src_sha16")
    out["top_kernels"] = [entry(k, *v) for k, v in ranked[:3]]
    # the grouped weight-gradient launches (round 2's dominant kernel family, now ~10 launches per step), whatever their rank
    out["weight_gradient_kernels"] = [entry(k, *v) for k, v in ranked if k.startswith("gemm_group")]
    out["next_kernels_ms_per_step"] = {k: round(v[0], 3) for k, v in ranked[1:6]}
    out["attention_kernels"] = [entry(k, *v) for k, v in ranked if k in tr.valu]
    return out


def stage_block(net, body, B, size, steps=2):
    """forward + backward time of every SURVEY §8d stage (module hooks + HIP events, weight gradients on the main stream)"""
    from cenet_amd import ops
    old = ops.set_wgrad_overlap(False)
    ev = {k: {"f": [], "b": []} for k in STAGES}
    # the two stages fed by the network input: nothing upstream needs a gradient, so their full-backward hook fires before
    # their kernels run.  Their interval is closed by the NEXT backward pre-hook that fires (autograd runs out.rb between
    # out.up and decoder.dec1) or, for patch_embed1 (last of the pass), by an event recorded when loss.backward() returns.
    nograd = ("backbone.patch_embed1", "out.rb")
    order = []  # (stage, start event) of every backward pre-hook, in firing order
    handles = []
    mods = dict(net.named_modules())

    def hook(name, m):
        st = {}

        def fpre(mod, inp):
            st["f0"] = torch.cuda.Event(enable_timing=True)
            st["f0"].record()

        def fpost(mod, inp, outp):
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            ev[name]["f"].append((st["f0"], e))

        def bpre(mod, gout):
            st["b0"] = torch.cuda.Event(enable_timing=True)
            st["b0"].record()
            order.append((name, st["b0"]))

        def bpost(mod, gin, gout):
            if name in nograd:
                return
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            ev[name]["b"].append((st["b0"], e))
        handles.extend([m.register_forward_pre_hook(fpre), m.register_forward_hook(fpost),
                        m.register_full_backward_pre_hook(bpre), m.register_full_backward_hook(bpost)])
    for name in STAGES:
        m = mods[name]
        # ModuleList stages (block1-4) and the Sequential containers of the output head (its forward calls their children
        # one by one) are hooked per child; a child that is never called (the MaxPool2d folded into ops.maxpool2_scale) adds 0
        for sub in (list(m) if isinstance(m, (torch.nn.ModuleList, torch.nn.Sequential)) else [m]):
            hook(name, sub)

    wg_ev = []  # the grouped weight-gradient launches (recorded during backward, issued when it ends): their own line

    def after_backward():
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        order.append(("<end of backward>", e))
        ops.wgrad_flush()
        e2 = torch.cuda.Event(enable_timing=True)
        e2.record()
        wg_ev.append((e, e2))
    try:
        for _ in range(steps):
            _hold_stream()
            ops.wgrad_hold(True)  # no end-of-backward callback in this pass: after_backward() issues the flush itself
            body(after_backward=after_backward)
        torch.cuda.synchronize()
    finally:
        ops.wgrad_hold(False)
        for h in handles:
            h.remove()
        ops.set_wgrad_overlap(old)
    for i, (name, e0) in enumerate(order[:-1]):
        if name in nograd:
            ev[name]["b"].append((e0, order[i + 1][1]))
    peak = peak_tflops()
    scale_f = (size / 224.0) ** 2
    out = {}
    for name, (gf, elems) in STAGES.items():
        tf = sum(a.elapsed_time(b) for a, b in ev[name]["f"]) / steps
        tb = sum(a.elapsed_time(b) for a, b in ev[name]["b"]) / steps
        es = 2 if peak == PEAK_BF16_MFMA_TFLOPS else 4
        fl = 3.0 * stage_gflop(name, size) * 1e9 * B  # fwd + bwd = 3 x fwd; attention terms grow with size^4
        by = 3.0 * elems * es * B * scale_f
        t_m, t_h = fl / (peak * 1e12) * 1e3, by / (PEAK_HBM_GBS * 1e9) * 1e3
        bound_ms = max(t_m, t_h)
        key = name if name.startswith("out.") else name.split(".", 1)[1].replace("skip_enhancer", "DSEB")
        out[key] = {
            "fwd_ms": round(tf, 3), "bwd_ms": round(tb, 3), "bound": "mfma" if t_m >= t_h else "hbm",
            "bound_ms": round(bound_ms, 4), "frac": round(bound_ms / max(tf + tb, 1e-9), 4)}
    if wg_ev:
        # not a SURVEY stage: the Linear / 1x1-conv weight gradients of ALL stages, reduced by the grouped launches after the
        # backward pass (their time is therefore in none of the per-stage bwd_ms above)
        out["grouped_weight_gradients"] = {"ms": round(sum(a.elapsed_time(b) for a, b in wg_ev) / steps, 3)}
    return out


def cpu_baseline(steps):
    """The oracle (plain PyTorch fp32 restatement, parity-pinned to the reference) timed on the host cores:
    BASELINE.json configs[0] = ACDC 224x224 4-class, batch 4, forward + Dice/CE + backward + SGD."""
    from oracle import cenet_oracle as O
    cfg = O.CENetConfig()
    n = min(os.cpu_count() or 8, 32)  # a batch-4 step stops scaling (and then regresses) beyond ~32 host threads
    torch.set_num_threads(n)
    sd = O.make_state_dict(cfg, seed=1)
    params = {k: v.requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running_" not in k}
    mom = {}
    x, lab = O.synthetic_batch(4, 1, 4, seed=1234)

    def step():
        for v in params.values():
            v.grad = None
        loss = O.criterion(O.cenet_forward(sd, x, cfg, training=True), lab, 4)
        loss.backward()
        with torch.no_grad():
            for k, v in params.items():
                g = v.grad + 1e-4 * v
                mom[k] = g.clone() if k not in mom else mom[k].mul_(0.9).add_(g)
                v.sub_(0.01 * mom[k])
    step()  # warm-up
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    dt = (time.perf_counter() - t0) / steps
    return {"value": round(4.0 / dt, 4), "unit": "images/s", "cores": n, "kind": "port",
            "sample": f"{steps} train steps of ACDC 224x224 4-class batch=4 fp32 (oracle/cenet_oracle.py), {dt:.2f} s/step"}


class _Watchdog:
    """N > 1: a collective that never completes (a rank died, a link hung) blocks every other rank inside the GPU queue, and the
    driver would wait for its own time limit.  A daemon thread watches a heartbeat the main thread touches at every phase of the run
    (set-up, each candidate launch form, warm-up, the timed loop's final synchronise, the instrumented passes); when the heartbeat
    is older than the limit (CENET_WATCHDOG_S, default 600 s; 0 disables) the process prints what it was doing and EXITS NON-ZERO
    with os._exit — never by re-executing itself (a process that has touched the GPU must not exec)."""

    def __init__(self):
        self.t = time.monotonic()
        self.what = "start"
        self.limit = 0.0
        self.thread = None
        self.done = False

    def beat(self, what):
        self.t = time.monotonic()
        self.what = what

    def start(self, limit_s, rank):
        import threading
        if limit_s <= 0 or self.thread is not None:
            return
        self.limit = limit_s
        self.beat("process group up")

        def run():
            while not self.done:
                time.sleep(min(5.0, self.limit / 4))
                if not self.done and time.monotonic() - self.t > self.limit:
                    print(f"[bench] watchdog: rank {rank} made no progress for {self.limit:.0f} s in phase '{self.what}' "
                          "(hung collective?) - exiting with status 3", file=sys.stderr, flush=True)
                    os._exit(3)
        self.thread = threading.Thread(target=run, name="bench-watchdog", daemon=True)
        self.thread.start()

    def stop(self):
        self.done = True


WATCHDOG = _Watchdog()


def _time_steps(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def dist_identity(dev, world):
    """What makes an N > 1 line self-proving: the collective backend, its version, and how many DISTINCT GPUs the ranks really
    sit on (every rank contributes its device's PCI identity / UUID; a launch that put two ranks on one device, or one that fell
    back to another backend, shows here).  Collective: every rank calls it."""
    pr = torch.cuda.get_device_properties(dev)
    ident = str(getattr(pr, "uuid", "")) or "-"
    pci = tuple(getattr(pr, k, -1) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
    mine = (os.uname().nodename, ident, pci, pr.name)
    allv = [None] * world
    dist.all_gather_object(allv, mine)
    backend = dist.get_backend()
    rec = {"backend": backend, "world_size": world, "distinct_devices": len({(v[0], v[1], v[2]) for v in allv}),
           "hosts": len({v[0] for v in allv}), "device_names": sorted({v[3] for v in allv})}
    if backend == "nccl":
        try:
            rec["nccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())  # (RCCL reports its NCCL API level)
        except Exception as e:  # pragma: no cover
            rec["nccl_version"] = f"unavailable ({type(e).__name__})"
    return rec


def main():
    a = parse()
    cfg = CONFIGS[a.config]
    B = a.batch or cfg["batch"]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    # (test aids: CENET_DEVICE pins every rank to one device and CENET_DIST_BACKEND=gloo carries the collectives, so that the
    # N > 1 code path can run with two ranks on a one-GPU box — RCCL refuses two ranks on one device; tests/test_bench_dist.py)
    local = int(os.environ.get("CENET_DEVICE", local))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # CENET_FORCE_DIST=1 runs the whole RCCL path (process group, hooks, side stream) even with a single rank, so the
    # distributed code can be exercised on a 1-GPU box (tests/test_bench_dist.py)
    use_dist = world > 1 or os.environ.get("CENET_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        import datetime
        wd_s = float(os.environ.get("CENET_WATCHDOG_S", "600"))
        dist.init_process_group(os.environ.get("CENET_DIST_BACKEND", "nccl"), rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=max(60.0, 2.0 * wd_s)))
        WATCHDOG.start(wd_s, rank)
    dist_record = dist_identity(dev, world) if use_dist else None

    from cenet_amd import kern, losses, optim, parallel
    kern.set_compute_bf16(a.dtype == "bf16")
    from cenet_amd import ops
    ops.set_wgrad_overlap(not a.no_overlap)  # weight gradients on a second HIP stream, beside the data-gradient chain
    net = make_model(dev, cfg)
    arena = optim.ParamArena(net, optim.cenet_segments())
    reducer = parallel.GradReducer(arena, force=use_dist, bf16_buckets=a.bf16_buckets) if use_dist else None
    if reducer is not None:
        reducer.broadcast_state(net)
        parallel.attach(net, reducer)
    opt = optim.FusedSGD(arena, lr=0.01, momentum=0.9, weight_decay=1e-4, grad_scale=1.0 / world)
    sched = optim.PolyLR(opt, max_iterations=100000)
    crit = losses.Criterion(cfg["classes"], argparse.Namespace(loss_type="dice,ce", loss_weights="0.5,0.5"))
    x, lab = synthetic(B, dev, seed=1234 + rank, cfg=cfg)

    comm = [True]  # (False: rank 0's instrumented passes at N > 1 — the other ranks are not in the step, so no collective)

    def body(sync_hyper=True, after_backward=None):
        opt.zero_grad()
        loss = crit(net(x), lab)
        loss.backward()
        if after_backward is not None:  # (instrumented passes only)
            after_backward()
        if reducer is not None and comm[0]:
            reducer.finish()
        opt.step(sync_hyper=sync_hyper)
        return loss

    def fwd_bwd():
        opt.zero_grad()
        loss = crit(net(x), lab)
        loss.backward()
        return loss

    graphed = None
    launch_note = None
    dist_launch = "eager"
    # N == 1, auto: eager launches are host-bound (~1.3 k C-ABI calls of ~20 us of Python + ctypes each) and replay is not,
    # so which one is faster depends on the host; a few steps of each decide.  (N > 1: next block.)
    # The weight-gradient stream pays off for EAGER launches (its kernels fill the gaps the host leaves) and costs under
    # replay: a captured second stream becomes extra hardware queues whose kernels share the chip with the main chain instead
    # of filling idle CUs, and the cross-queue dependencies cost ~6 % of the step (measured: 1 198 vs 1 273 images/s).  The
    # graphs are therefore captured with the weight gradients on the main stream; eager steps keep the second stream.
    overlap_eager = not a.no_overlap
    if not use_dist and a.graph != "off":
        from cenet_amd.graph import GraphedStep
        try:
            ops.set_wgrad_overlap(False)
            graphed = GraphedStep(lambda: body(sync_hyper=False), optimizer=opt, warmup=2)
        except Exception as e:  # capture is an optimisation: fall back to eager launches of the same kernels
            if a.graph == "on":
                raise
            print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eager", file=sys.stderr)
            graphed = None
            torch.cuda.synchronize()
        finally:
            ops.set_wgrad_overlap(overlap_eager)
        if graphed is not None and a.graph == "auto":
            for _ in range(2):
                body()
                graphed()
            t_e, t_g = _time_steps(body, 4), _time_steps(graphed, 4)
            launch_note = (f"auto: eager (weight gradients on a second stream) {t_e * 1e3:.1f} ms vs hipGraph replay (one stream) "
                           f"{t_g * 1e3:.1f} ms over 4 untimed steps each")
            HOLD_MS[0] = min(80.0, max(20.0, 1.25 * t_e * 1e3))
            if t_e <= t_g:
                graphed = None

    dist_times = {}
    if use_dist and a.graph != "off":
        # N > 1: three launch forms of the same step, timed for four untimed steps each on every rank; rank 0's pick is
        # broadcast (the collective sequence — one all-reduce per arena segment — is the same in all three):
        #   eager      ~1 300 launches per step, all-reduces started from backward hooks (host-bound; 8 ranks share one host)
        #   split      forward + backward as ONE hipGraph, the five all-reduces after it, SGD as a second graph
        #              (cenet_amd.graph.GraphedSplitStep: ~8 host calls, no backward / all-reduce overlap)
        #   segmented  the backward cut at the encoder stage outputs: graph 0 = forward + head/decoder backward, graphs 1-4 =
        #              one encoder stage each, segment k's all-reduce issued eagerly behind graph k so that it runs beside
        #              graphs k+1.. (cenet_amd.graph.SegmentedStep: ~15 host calls, overlap kept)
        from cenet_amd.graph import GraphedSplitStep, SegmentedStep
        cands = {}
        parallel.attach(net, None)  # no hooks inside the captures: the step objects drive the reducer themselves
        ops.set_wgrad_overlap(False)
        for name, make in (("split", lambda: GraphedSplitStep(fwd_bwd, opt, reducer.finish, warmup=2)),
                           ("segmented", lambda: SegmentedStep(net, lambda: crit(net(x), lab), opt, reducer.segment_ready,
                                                               reducer.finish, warmup=2))):
            try:
                cands[name] = make()
            except Exception as e:
                print(f"[bench] {name} hipGraph capture failed on rank {rank} ({type(e).__name__}: {e})", file=sys.stderr)
                torch.cuda.synchronize()
        ops.set_wgrad_overlap(overlap_eager)
        # all ranks or none (the timing loops below contain barriers): EVERY rank reaches this collective, also one whose
        # capture threw — with --graph on the failure is raised on all ranks only after they have agreed on it
        okf = torch.tensor([1.0 if "split" in cands else 0.0, 1.0 if "segmented" in cands else 0.0], device=dev)
        dist.all_reduce(okf, op=dist.ReduceOp.MIN)
        for i, name in enumerate(("split", "segmented")):
            if okf[i].item() < 1.0:
                cands.pop(name, None)
        if not cands and a.graph == "on":
            raise SystemExit("--graph on: no hipGraph form of the step could be captured on every rank")
        times = {}
        parallel.attach(net, reducer)
        if a.graph != "on":
            WATCHDOG.beat("timing the eager launch form")
            for _ in range(2):
                body()
            dist.barrier()
            times["eager"] = _time_steps(body, 4)
        parallel.attach(net, None)
        for name, cand in cands.items():
            WATCHDOG.beat(f"timing the {name} launch form")
            for _ in range(2):
                cand()
            dist.barrier()
            times[name] = _time_steps(cand, 4)
        dist_times = dict(times)
        order = ["eager", "split", "segmented"]
        best = min(times, key=times.get) if times else "eager"
        if a.dist_launch != "auto":
            if a.dist_launch != "eager" and a.dist_launch not in cands:
                raise SystemExit(f"--dist-launch {a.dist_launch}: that form could not be captured on every rank")
            best = a.dist_launch
        pick = torch.tensor([float(order.index(best))], device=dev)
        dist.broadcast(pick, 0)
        choice = order[int(pick.item())]
        launch_note = ("auto over 4 untimed steps each (rank 0 decides): " +
                       ", ".join(f"{k} {v * 1e3:.1f} ms" for k, v in times.items()) + f" -> {choice}")
        if choice == "eager":
            parallel.attach(net, reducer)
        else:
            graphed = cands[choice]
        dist_launch = {"eager": "eager", "split": "two hipGraphs + eager all-reduce",
                       "segmented": "six hipGraphs (backward cut per arena segment) + overlapped eager all-reduce"}[choice]

    def step():
        loss = graphed() if graphed is not None else body()
        sched.step()
        return loss

    WATCHDOG.beat("warm-up")
    for _ in range(a.warmup):
        step()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    WATCHDOG.beat("timed loop")
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    WATCHDOG.beat("timed loop done")
    if use_dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    final_loss = float(loss.item())

    if rank == 0:
        ms = dt / a.steps * 1e3
        out = {"metric": "training images/sec (224x224, 4-class)" if a.config == "acdc" else
               f"training images/sec ({cfg['size']}x{cfg['size']}, {cfg['classes']}-class)",
               "value": round(B * world * a.steps / dt, 3),
               "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
               "config": {"workload": f"{cfg['name']}, batch={B}/GPU, random-init PVTv2-b2 CENet "
                                      f"(heads {cfg['heads']}, scales {cfg['scales']}), "
                                      "fwd + Dice/CE + bwd + grad all-reduce + SGD(momentum .9, wd 1e-4)",
                          "preset": a.config, "batch_per_gpu": B, "global_batch": B * world, "parallelism": f"dp{world}", **({"grad_buckets": "bf16"} if a.bf16_buckets else {}),
                          "launch": ("eager" if graphed is None else (dist_launch if use_dist else "hipGraph replay")),
                          "final_loss": round(final_loss, 5)}}
        if launch_note:
            out["config"]["launch_choice"] = launch_note
        if dist_record is not None:
            out["config"]["rccl"] = dict(dist_record, launch_forms_ms={k: round(v * 1e3, 3) for k, v in dist_times.items()})
        if not a.no_roofline:
            # N > 1: rank 0's instrumented passes run AFTER the timed loop with the collectives switched off (the other ranks wait at
            # the final barrier; the dominant kernel and the stage times do not depend on the all-reduce), so that every line of a
            # scaling record carries its roofline
            WATCHDOG.beat("instrumented passes (roofline)")
            if use_dist:
                parallel.attach(net, None)
                comm[0] = False
            out["roofline"] = roofline_block(body)
            if use_dist:
                out["roofline"]["note"] = "rank 0, after the timed loop, collectives off (per-GPU kernel times do not depend on N)"
            out["roofline_stages"] = stage_block(net, body, B, cfg["size"])
            WATCHDOG.beat("instrumented passes done")
            # the whole step against both roofs (fwd + bwd = 3 x the forward contractions; stage-boundary tensors
            # three times, parameters read twice and their gradient written once, the optimizer's three streams)
            sf = (cfg["size"] / 224.0) ** 2
            fl = 3.0 * sum(stage_gflop(k, cfg["size"]) for k in STAGES) * 1e9 * B
            es = 2 if a.dtype == "bf16" else 4
            by = 3.0 * 9.42e6 * es * B * sf + arena.numel * (2 * es + 4 + 5 * 4)
            t = ms * 1e-3
            out["whole_step"] = {"flops": int(fl), "bytes": int(by), "tflops": round(fl / t / 1e12, 1),
                                 "frac_mfma": round(fl / t / (peak_tflops() * 1e12), 4),
                                 "gbs": round(by / t / 1e9, 1), "frac_hbm": round(by / t / (PEAK_HBM_GBS * 1e9), 4),
                                 "ideal_ms": round(max(fl / (peak_tflops() * 1e12), by / (PEAK_HBM_GBS * 1e9)) * 1e3, 3)}
        if world == 1 and a.dtype == "bf16" and not a.no_f32:
            # the same step in the fp32 PARITY mode (fp32 tensors end to end: the mode the 1e-3 logit / 1e-4 Dice tests run in)
            kern.set_compute_bf16(False)
            n32 = max(3, min(10, a.steps // 5))
            for _ in range(2):
                body()
            d32 = _time_steps(body, n32)
            out["parity_mode_f32"] = {"value": round(B / d32, 3), "unit": "images/s", "ms_per_step": round(d32 * 1e3, 3),
                                      "steps": n32}
            kern.set_compute_bf16(True)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.cpu_steps)
        line = json.dumps(out)
    if use_dist:
        WATCHDOG.beat("final barrier")
        dist.barrier()
        dist.destroy_process_group()
    WATCHDOG.stop()
    if rank == 0:
        # the JSON line is the LAST thing on stdout: flush what native libraries (the RCCL version banner) still hold in C stdio
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(line, flush=True)


if __name__ == "__main__":
    main()
