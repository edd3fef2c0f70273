"""Builds libinfidex_hip.so (HIP kernels + C ABI + C++ host engine) in-tree for gfx950.

hipcc cross-compiles without a GPU. The .so is git-ignored but travels to the GPU box with the snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libinfidex_hip.so")
LIB_EXP = os.path.join(HERE, "libinfidex_hip_experiments.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

# -ffp-contract=off: BM25 / fusion arithmetic must round like the reference's separate fp32 mul/add (no FMA fusion)
COMMON = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-value", "-Wno-unused-result"]


def sources():
    out = []
    for root, _, files in os.walk(CSRC):
        for f in files:
            if f.endswith((".hip", ".inc", ".h", ".cpp")):
                out.append(os.path.join(root, f))
    inc = os.path.join(os.path.dirname(HERE), "include")
    out += [os.path.join(inc, f) for f in os.listdir(inc)]
    return out


def _build(lib, obj_dev, obj_host, extra, force, verbose):
    if not force and os.path.exists(lib) and all(os.path.getmtime(s) <= os.path.getmtime(lib) for s in sources()):
        return lib
    cmds = [
        [HIPCC, "--offload-arch=gfx950", *COMMON, *extra, "-c", os.path.join(CSRC, "infidex_hip.hip"), "-o", obj_dev],
        [HIPCC, *COMMON, "-march=x86-64-v3", "-x", "c++", "-c", os.path.join(CSRC, "host", "engine.cpp"), "-o", obj_host],
        [HIPCC, "--offload-arch=gfx950", "-shared", "-o", lib, obj_dev, obj_host, "-lpthread"],
    ]
    for c in cmds:
        if verbose:
            print(" ".join(c), file=sys.stderr)
        subprocess.check_call(c)
    return lib


def build(force=False, verbose=False):
    """The product library: the production kernels only."""
    return _build(LIB, os.path.join(CSRC, "infidex_hip.o"), os.path.join(CSRC, "engine.o"), [], force, verbose)


def build_experiments(force=False, verbose=False):
    """libinfidex_hip_experiments.so = the product plus the alternative k_accumulate designs (stage1b/c/d.hip.inc, -DINFX_BUILD_EXPERIMENTS): loaded only by
    the A/B parity test and the profiling scripts (INFX_LIB points the Python plumbing at it)."""
    return _build(LIB_EXP, os.path.join(CSRC, "infidex_hip_exp.o"), os.path.join(CSRC, "engine.o"), ["-DINFX_BUILD_EXPERIMENTS"], force, verbose)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    if "--experiments" in sys.argv:
        print(build_experiments(force="--force" in sys.argv, verbose=True))
