#!/usr/bin/env python3
"""Builds libhyperpocket_hip.so (all HIP kernels + the C ABI) for gfx950 with hipcc, in-tree.

    python 3d-point-clouds-autocomplete_amd/build.py [--force]

The .so lands next to the Python host package (hyperpocket_amd/), is git-ignored, and travels
to the GPU box with the snapshot.  hipcc cross-compiles without a GPU.
"""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "hyperpocket_amd", "libhyperpocket_hip.so")
OBJ = os.path.join(HERE, "build")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fvisibility=hidden", "-Wall", "-Wno-unused-function", "-I" + CSRC]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = sorted(glob.glob(os.path.join(CSRC, "*.h")))
    os.makedirs(OBJ, exist_ok=True)
    jobs = []
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s)[:-4] + ".o")
        if force or _stale(o, [s] + hdrs):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        cmd = [HIPCC] + FLAGS + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(cc, jobs))
    objs = [os.path.join(OBJ, os.path.basename(s)[:-4] + ".o") for s in srcs]
    if force or jobs or _stale(OUT, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print("built", OUT)
