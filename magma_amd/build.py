"""Build libmagma_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m magma_amd.build [--force]

The shared library is the only compiled artefact; it is git-ignored but travels with the tree.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmagma_hip.so")
SOURCES = ["small.hip", "dense.hip", "dense_dag.hip", "capi.hip", "driver.hip", "lbfgsb.cpp"]
HEADERS = ["common.h", "kernels.h", "mfma_sweep.h", "dense_tile.h", "lbfgsb_core.h", "driver.h", os.path.join("..", "..", "include", "magma_hip.h")]
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics",
         "-ffp-contract=off", "-Wall", "-Wno-unused-function",
         "-mllvm", "-pragma-unroll-threshold=200000"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("MAGMA_EXTRA_FLAGS", "").split()
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            lang = ["-x", "hip"] if src.endswith(".hip") else []
            jobs.append([hipcc] + FLAGS + extra + lang + ["-c", s, "-o", o])
    if jobs:
        def run(cmd):
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                sys.stderr.write(r.stdout + r.stderr)
                raise subprocess.CalledProcessError(r.returncode, cmd)
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or not os.path.exists(LIB):
        cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs + ["-ldl", "-lpthread"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
