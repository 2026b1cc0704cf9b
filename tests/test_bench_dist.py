"""bench.py contract + the RCCL code path on the real GPU.

Only one GPU is available to the tests, so the distributed path (process group, broadcast, backward-hook driven
segment all-reduces on the side stream, max-over-ranks timing) is exercised with a ONE-rank RCCL group launched exactly
the way the driver launches N ranks (`python -m torch.distributed.run ...`), forced on with CENET_FORCE_DIST=1.
The N>1 arithmetic (mean of shard gradients, per-rank BN buffers) is pinned by tests/test_parallel_gloo.py on CPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline"}


def _last_json(out: str):
    for line in reversed(out.strip().splitlines()):
        line = line.strip()
        if line.startswith("{") and line.endswith("}"):
            return json.loads(line)
    raise AssertionError("no JSON line in bench output:\n" + out[-2000:])


@pytest.mark.gpu
def test_bench_single_process_contract():
    p = subprocess.run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _last_json(p.stdout)
    assert REQUIRED <= set(d), REQUIRED - set(d)
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["value"] > 0 and abs(d["value"] - 32 * 1000.0 / d["ms_per_step"]) / d["value"] < 1e-3
    r = d["roofline"]
    assert r["bound"] in ("mfma", "hbm") and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert "workload" in d["config"] and "model" not in d["config"]


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--bf16-buckets"], ["--dist-launch", "segmented"], ["--dist-launch", "split"],
                                   ["--dist-launch", "eager"]])
def test_bench_under_torchrun_with_rccl_group(extra):
    env = dict(os.environ, CENET_FORCE_DIST="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", "29611", "bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-f32"] + extra
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    d = _last_json(p.stdout)
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["parallelism"] == "dp1"
    assert d["config"]["final_loss"] == d["config"]["final_loss"]  # not NaN
    r = d["config"]["rccl"]  # the record that makes an N > 1 line self-proving (here: one rank, the real RCCL backend)
    assert r["backend"] == "nccl" and r["world_size"] == 1 and r["distinct_devices"] == 1 and r["hosts"] == 1
    assert r["nccl_version"][0].isdigit() and set(r["launch_forms_ms"]) <= {"eager", "split", "segmented"}
    if "--dist-launch" in extra:
        want = {"segmented": "six hipGraphs", "split": "two hipGraphs", "eager": "eager"}[extra[-1]]
        assert d["config"]["launch"].startswith(want), d["config"]["launch"]


@pytest.mark.gpu
@pytest.mark.parametrize("preset,batch,size,classes", [("synapse", 24, 224, 9), ("ham512", 2, 512, 2)])
def test_bench_other_presets(preset, batch, size, classes):
    """SURVEY.md §8d C4 (Synapse at its full per-GPU batch) and C5 (HAM10000 at 512x512, three FEA scales; batch 2 here,
    8 in the preset; parity at this size: tests/test_ham_oracle.py; here the step must run and train)"""
    p = subprocess.run([sys.executable, "bench.py", "--config", preset, "--batch", str(batch), "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-f32"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _last_json(p.stdout)
    assert d["config"]["preset"] == preset and d["config"]["batch_per_gpu"] == batch
    assert f"{size}x{size}" in d["metric"] and f"{classes}-class" in d["metric"]
    assert d["value"] > 0 and d["config"]["final_loss"] == d["config"]["final_loss"] and d["config"]["final_loss"] < 5.0
    st = dict(d["roofline_stages"])
    assert st.pop("grouped_weight_gradients")["ms"] > 0  # (all stages' weight gradients, issued after the backward pass)
    assert len(st) == 21 and all(v["fwd_ms"] > 0 for v in st.values())
    assert d["roofline"]["bound"] in ("mfma", "hbm")


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--dist-launch", "segmented"]])
def test_bench_with_two_ranks_on_one_gpu(extra):
    """`--gpus 2` exactly as the driver launches it (torch.distributed.run, two processes), with both ranks pinned to cuda:0 and
    the collectives over gloo (RCCL refuses two ranks on one device; only a one-GPU box is available to the tests): the whole
    N > 1 control flow of bench.py — broadcast, capture of the split / segmented graph forms on every rank, the agreement
    all-reduce, the timing loops with their barriers, rank 0's pick and its broadcast, max-over-ranks timing — with world 2."""
    env = dict(os.environ, CENET_DIST_BACKEND="gloo", CENET_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29633", "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-f32", "--batch", "8"] + extra
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=1200, env=env)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    d = _last_json(p.stdout)
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 16
    assert d["value"] > 0 and abs(d["value"] - 16 * 1000.0 / d["ms_per_step"]) / d["value"] < 1e-3
    assert d["config"]["final_loss"] == d["config"]["final_loss"]
    # every line of a scaling record carries its roofline: rank 0's instrumented passes run after the timed loop, collectives off
    assert d["roofline"]["bound"] in ("mfma", "hbm") and d["roofline"]["frac"] > 0 and "rank 0" in d["roofline"]["note"]
    assert len(d["roofline_stages"]) == 22
    r = d["config"]["rccl"]
    # both ranks were pinned to cuda:0 on purpose: the record must SAY that the two ranks share one device, over gloo
    assert r["backend"] == "gloo" and r["world_size"] == 2 and r["distinct_devices"] == 1 and r["hosts"] == 1
    assert "nccl_version" not in r and all(v > 0 for v in r["launch_forms_ms"].values())
    if extra:
        assert d["config"]["launch"].startswith("six hipGraphs"), d["config"]["launch"]
    else:
        assert "->" in d["config"]["launch_choice"]

