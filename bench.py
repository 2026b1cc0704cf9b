"""bench.py — training throughput of the CENet hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" = the reference's train-step body (src/main_acdc.py:237-257) on one synthetic batch already resident in
HBM: zero_grad -> CENet forward -> Dice+CE (0.5/0.5) -> backward -> gradient all-reduce (RCCL) -> fused SGD.
Workload = BASELINE.json configs[1]: ACDC 224x224, 4 classes, batch 32 per GPU, random-init PVTv2-b2 CENet.
Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel, timed live with
HIP events on the launch stream) and `cpu_baseline` (the oracle timed on the host cores, rank 0, N == 1 only).
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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU (BASELINE.json configs[1]: 32)")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="bf16",
                    help="operand precision of the MFMA contractions (accumulation, softmax and norm statistics stay fp32)")
    ap.add_argument("--graph", choices=["auto", "on", "off"], default="auto",
                    help="replay the step as one captured hipGraph (auto: fall back to eager launches if capture fails)")
    ap.add_argument("--no-f32", action="store_true", help="skip the extra fp32 parity-mode measurement (N=1, bf16 runs)")
    ap.add_argument("--no-overlap", action="store_true", help="keep the weight-gradient kernels on the main stream")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=3)
    return ap.parse_args()


def make_model(dev):
    from cenet_amd.networks import CENet
    torch.manual_seed(1234)
    net = CENet(input_channels=1, num_classes=4, scale_factors=[1.0, 0.5], diffatt_num_heads=[4, 4, 4],
                encoder="pvt_v2_b2", enc_pretrain=False, skip_mode="cat", dec_up_block="eucb", out_merge_mode="cat",
                out_up_block="upcn", out_up_ks=3)
    return net.to(dev).train()


def synthetic(B, dev, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 1, 224, 224, generator=g)
    lab = torch.randint(0, 4, (B, 224, 224), generator=g).float()
    return x.to(dev), lab.to(dev)


def dominant_kernel_probe(dev, B):
    """Times the single heaviest launch of the step in isolation with HIP events on the launch stream:
    gemm_f32_kernel<im2col> computing out.rb.0.conv2 forward (5x5, 32->32 channels @224x224, reference out.py:41-44).
    Algorithmic FLOPs per launch = 2 * B * Cout * Ho*Wo * Cin*k*k."""
    from cenet_amd import kern, ops
    dt = torch.bfloat16 if kern.get_compute_bf16() else torch.float32
    x = torch.randn(B, 32, 224, 224, device=dev).to(dt)
    w = torch.randn(32, 32, 5, 5, device=dev) * 0.03
    with torch.no_grad():
        for _ in range(2):
            ops.conv2d_nchw(x, w, None, stride=1, pad=2)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        e0.record()
        for _ in range(reps):
            ops.conv2d_nchw(x, w, None, stride=1, pad=2)
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    flops = 2.0 * B * 32 * 224 * 224 * 32 * 25
    achieved = flops / (ms * 1e-3) / 1e12
    peak = peak_tflops()
    # HBM bytes per launch from rocprofv3 PMC passes at B=32 (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE; profiles/
    # r01_pmc_roofline_kernel*.csv): fp32 mode = implicit-GEMM gather, bf16 mode = LDS-halo direct conv; algorithmic = 411 MB
    bf16 = peak != PEAK_F32_MFMA_TFLOPS
    traffic = (4.2e8 if bf16 else 1.157e9) if B == 32 else None
    name = ("conv_direct_bf16_kernel<32,32,5>" if bf16 else "gemm_kernel<float,32,256,im2col>") + \
        " (out.rb.0.conv2 fwd, 5x5 32->32 @224^2)"
    alg_bytes = 4.0 * (2 * B * 32 * 224 * 224 + 32 * 32 * 25)  # input + output + weights, fp32 in HBM
    # the binding roof depends on the operand mode: fp32 MFMA (157 TF) binds before HBM; at the bf16 MFMA rate (2.5 PF)
    # the same launch is HBM-bound (82 GFLOP / 2.5 PF = 33 us < 411 MB / 8 TB/s = 51 us)
    if flops / (peak * 1e12) >= alg_bytes / (PEAK_HBM_GBS * 1e9):
        return {"bound": "mfma", "kernel": name, "achieved": round(achieved, 3), "peak": peak, "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4), "traffic": traffic, "avg_launch_ms": round(ms, 4)}
    gbs = alg_bytes / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": name, "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": round(gbs / PEAK_HBM_GBS, 4), "traffic": traffic, "avg_launch_ms": round(ms, 4),
            "mfma_tflops": round(achieved, 2)}


def peak_tflops():
    from cenet_amd import kern
    return PEAK_BF16_MFMA_TFLOPS if kern.get_compute_bf16() else PEAK_F32_MFMA_TFLOPS


def _time(fn, reps=5):
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


def extra_kernel_probes(dev, B):
    """Two more live-timed launches: the heaviest single launch of the step (flash backward dK/dV of DSEB-56^2: 8 softmax
    heads, N=3136, hd=16, dv=32 — fp32 MFMA in every mode this round) and the largest plain GEMM (stage-1 Mlp fc1)."""
    from cenet_amd import kern, ops
    out = []
    N, H, hd = 3136, 4, 16
    E = 2 * H * hd
    dt = torch.bfloat16 if kern.get_compute_bf16() else torch.float32
    q, k, v = (torch.randn(B, N, E, device=dev).to(dt).requires_grad_(True) for _ in range(3))
    U = ops.diff_attention_heads(q, k, v, H)
    g = torch.randn_like(U)
    t_f = _time(lambda: ops.diff_attention_heads(q.detach(), k.detach(), v.detach(), H))
    t_fb = _time(lambda: ops.diff_attention_heads(q, k, v, H).backward(g))
    fl_f = 2.0 * B * 2 * H * N * N * (hd + 2 * hd)
    fl_b = 2.0 * B * 2 * H * N * N * (2 * hd + 2 * 2 * hd) + 2.0 * B * 2 * H * N * N * (2 * hd + 2 * hd)
    peak = peak_tflops()
    bf = kern.get_compute_bf16()
    kf = "flashc_fwd_kernel<32,32,2>" if bf else "flash_fwd_kernel<16,32>"
    kb = "flashc_bwd_dq+dkv_kernel<32,32,2>" if bf else "flash_bwd_dq+dkv_kernel<16,32>"
    out.append({"kernel": kf + " (DSEB-56^2 differential attention: 8 heads, N=3136, hd=16, dv=32)", "bound": "mfma",
                "achieved": round(fl_f / t_f / 1e9, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(fl_f / t_f / 1e9 / peak, 4), "avg_launch_ms": round(t_f, 3)})
    out.append({"kernel": kb + " (same problem, both backward launches)", "bound": "mfma",
                "achieved": round(fl_b / (t_fb - t_f) / 1e9, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(fl_b / (t_fb - t_f) / 1e9 / peak, 4), "avg_launch_ms": round(t_fb - t_f, 3)})
    R, K, Nn = B * 3136, 64, 512
    x = torch.randn(R, K, device=dev).to(dt)
    W = (torch.randn(Nn, K, device=dev) * 0.05).to(dt)
    y = torch.empty(R, Nn, device=dev, dtype=dt)
    t = _time(lambda: kern.gemm(kern.mat_plain(x, K, 1, kfast=1), kern.mat_plain(W, 1, K, kfast=1), y, R, Nn, K, scr=Nn, scc=1))
    by = (R * K + Nn * K + R * Nn) * 4.0
    out.append({"kernel": "gemm_kernel<128,128,plain> (stage-1 Mlp.fc1 fwd: 100352x64 @ 64x512)", "bound": "hbm",
                "achieved": round(by / t / 1e6, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(by / t / 1e6 / PEAK_HBM_GBS, 4),
                "avg_launch_ms": round(t, 4)})
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


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # CENET_FORCE_DIST=1 runs the whole RCCL path (process group, hooks, side stream) even with a single rank, so the
    # distributed code can be exercised on a 1-GPU box (tests/test_bench_dist.py)
    use_dist = world > 1 or os.environ.get("CENET_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=rank, world_size=world)

    from cenet_amd import kern, losses, optim, parallel
    kern.set_compute_bf16(a.dtype == "bf16")
    from cenet_amd import ops
    ops.set_wgrad_overlap(not a.no_overlap)  # weight gradients on a second HIP stream, beside the data-gradient chain
    net = make_model(dev)
    arena = optim.ParamArena(net, optim.cenet_segments())
    reducer = parallel.GradReducer(arena, force=use_dist) if use_dist else None
    if reducer is not None:
        reducer.broadcast_state(net)
        parallel.attach(net, reducer)
    opt = optim.FusedSGD(arena, lr=0.01, momentum=0.9, weight_decay=1e-4, grad_scale=1.0 / world)
    sched = optim.PolyLR(opt, max_iterations=100000)
    crit = losses.Criterion(4, argparse.Namespace(loss_type="dice,ce", loss_weights="0.5,0.5"))
    x, lab = synthetic(a.batch, dev, seed=1234 + rank)

    def body(sync_hyper=True):
        opt.zero_grad()
        loss = crit(net(x), lab)
        loss.backward()
        if reducer is not None:
            reducer.finish()
        opt.step(sync_hyper=sync_hyper)
        return loss

    graphed = None
    # auto: eager launches when the weight gradients overlap on their own stream (measured 45.7 ms eager vs 49.5 ms as a
    # replayed hipGraph, whose two branches the runtime interleaves less well), and with RCCL collectives inside the step
    # (N > 1; capture of multi-rank collectives could not be exercised on the 1-GPU development box).  Without the overlap
    # replay and eager launches are equal (the step is GPU-bound): capture then.
    if a.graph == "on" or (a.graph == "auto" and not use_dist and a.no_overlap):
        # the whole step (memset, ~1.8 k kernels, all-reduces, SGD) as ONE hipGraph replay per iteration
        from cenet_amd.graph import GraphedStep
        try:
            graphed = GraphedStep(lambda: body(sync_hyper=False), optimizer=opt, warmup=2)
        except Exception as e:  # capture is an optimisation: fall back to eager launches of the same kernels
            if a.graph == "on":
                raise
            print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eager", file=sys.stderr)
            graphed = None
            torch.cuda.synchronize()

    def step():
        loss = graphed() if graphed is not None else body()
        sched.step()
        return loss

    for _ in range(a.warmup):
        step()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    final_loss = float(loss.item())

    if rank == 0:
        ms = dt / a.steps * 1e3
        out = {"metric": "training images/sec (224x224, 4-class)", "value": round(a.batch * world * a.steps / dt, 3),
               "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
               "config": {"workload": "ACDC 224x224 4-class, batch=32/GPU, random-init PVTv2-b2 CENet, "
                                      "fwd + Dice/CE + bwd + grad all-reduce + SGD(momentum .9, wd 1e-4)",
                          "batch_per_gpu": a.batch, "global_batch": a.batch * world, "parallelism": f"dp{world}",
                          "launch": "hipGraph replay" if graphed is not None else "eager",
                          "final_loss": round(final_loss, 5)}}
        out["roofline"] = dominant_kernel_probe(dev, a.batch)
        out["roofline_extra"] = extra_kernel_probes(dev, a.batch)
        if world == 1 and a.dtype == "bf16" and not a.no_f32:
            # the same step in the fp32-operand PARITY mode (the mode the 1e-3 logit / 1e-4 Dice tests are run in)
            kern.set_compute_bf16(False)
            g32 = None
            if graphed is not None:
                from cenet_amd.graph import GraphedStep
                g32 = GraphedStep(lambda: body(sync_hyper=False), optimizer=opt, warmup=2)
            n32 = max(3, a.steps // 2)
            for _ in range(2):
                (g32() if g32 is not None else body())
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n32):
                (g32() if g32 is not None else body())
            torch.cuda.synchronize()
            d32 = (time.perf_counter() - t0) / n32
            out["parity_mode_f32"] = {"value": round(a.batch / d32, 3), "unit": "images/s", "ms_per_step": round(d32 * 1e3, 3),
                                      "steps": n32}
            kern.set_compute_bf16(True)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.cpu_steps)
        line = json.dumps(out)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
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
