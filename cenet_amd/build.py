"""Builds libcenet_hip.so (hipcc, gfx950) in-tree.  `python -m cenet_amd.build`."""
from __future__ import annotations

import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcenet_hip.so")


# per-file flags: the attention kernels never produce or consume NaNs on purpose (masked scores are -1e30), and without
# NaN semantics fmaxf on MFMA results needs no canonicalising v_max
# pvt_mlp.hip: its depthwise phase is bound by vector-instruction issue; SLP-packing scalar f32 math into v_pk_fma_f32 (two issue
# slots each) costs ~300 and/or/shift instructions per slab to assemble the 64-bit operands
EXTRA_FLAGS = {"attn_diff.hip": ["-fno-honor-nans"], "pvt_mlp.hip": ["-fno-slp-vectorize"]}


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def source_sha16() -> str:
    """hash of the kernel sources (csrc/*.hip, csrc/*.h, include/cenet_hip.h): identifies the BUILD a measurement belongs to
    independently of where and when hipcc ran (profiles/*_traffic.json, bench.py roofline.kernel_src_sha16)"""
    import hashlib
    h = hashlib.sha256()
    for f in sources() + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(HERE, "..", "include", "cenet_hip.h")]:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_hip(force: bool = False, verbose: bool = True) -> str:
    """hipcc --offload-arch=gfx950 -> cenet_amd/libcenet_hip.so (cross-compiles without a GPU)."""
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "cenet_hip.h")]
    if not force and not _stale(LIB, deps):
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "csrc", "obj"), exist_ok=True)
    for src in sources():
        obj = os.path.join(HERE, "csrc", "obj", os.path.basename(src) + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + deps[len(sources()):]):
            cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", obj,
                   "-Wno-unused-result"] + EXTRA_FLAGS.get(os.path.basename(src), [])
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    failed = [s for s, p in procs if p.wait() != 0]
    if failed:
        raise RuntimeError(f"hipcc failed for {failed}")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build_hip(force="--force" in sys.argv)
    print(LIB)
