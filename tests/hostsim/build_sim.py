"""TEST INFRASTRUCTURE: builds tests/hostsim/build/libcenet_sim.so — the kernel sources of cenet_amd/csrc
compiled with g++ against the SIMT interpreter in hipsim.h (no GPU needed)."""
from __future__ import annotations

import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "cenet_amd", "csrc")
OUT = os.path.join(HERE, "build")
LIB = os.path.join(OUT, "libcenet_sim.so")


def build_sim(force=False, sanitize=False):
    os.makedirs(OUT, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    deps = srcs + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "hipsim.*")) + \
        [os.path.join(ROOT, "include", "cenet_hip.h")]
    lib = LIB.replace(".so", "_asan.so") if sanitize else LIB
    if not force and os.path.exists(lib) and all(os.path.getmtime(d) <= os.path.getmtime(lib) for d in deps):
        return lib
    flags = ["-O3", "-std=c++17", "-fPIC", "-DCENET_HOSTSIM_BUILD", "-I", HERE, "-Wno-unused-function",
             "-Wno-attributes", "-fno-strict-aliasing"]
    if sanitize:
        flags += ["-fsanitize=undefined", "-fno-sanitize-recover=undefined"]
    objs, procs = [], []
    for s in srcs + [os.path.join(HERE, "hipsim.cpp")]:
        o = os.path.join(OUT, os.path.basename(s) + (".san.o" if sanitize else ".o"))
        objs.append(o)
        if force or not os.path.exists(o) or any(os.path.getmtime(d) > os.path.getmtime(o) for d in deps):
            procs.append((s, subprocess.Popen(["g++"] + flags + ["-x", "c++", "-c", s, "-o", o])))
    bad = [s for s, p in procs if p.wait() != 0]
    if bad:
        raise RuntimeError(f"g++ failed: {bad}")
    subprocess.check_call(["g++", "-shared", "-o", lib] + objs + (["-fsanitize=undefined"] if sanitize else []))
    return lib


if __name__ == "__main__":
    print(build_sim(force=True))
