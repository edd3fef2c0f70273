"""Builds libinfidex_hip.so (HIP kernels + C ABI + C++ host engine) in-tree for gfx950.

hipcc cross-compiles without a GPU. The .so is git-ignored but travels to the GPU box with the snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libinfidex_hip.so")
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


def build(force=False, verbose=False):
    if not force and os.path.exists(LIB) and all(os.path.getmtime(s) <= os.path.getmtime(LIB) for s in sources()):
        return LIB
    obj_dev = os.path.join(CSRC, "infidex_hip.o")
    obj_host = os.path.join(CSRC, "engine.o")
    cmds = [
        [HIPCC, "--offload-arch=gfx950", *COMMON, "-c", os.path.join(CSRC, "infidex_hip.hip"), "-o", obj_dev],
        [HIPCC, *COMMON, "-march=x86-64-v3", "-x", "c++", "-c", os.path.join(CSRC, "host", "engine.cpp"), "-o", obj_host],
        [HIPCC, "--offload-arch=gfx950", "-shared", "-o", LIB, obj_dev, obj_host, "-lpthread"],
    ]
    for c in cmds:
        if verbose:
            print(" ".join(c), file=sys.stderr)
        subprocess.check_call(c)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
