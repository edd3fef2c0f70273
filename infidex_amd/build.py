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


def _include_headers():
    inc = os.path.join(os.path.dirname(HERE), "include")
    return [os.path.join(inc, f) for f in os.listdir(inc)]


def device_sources():
    """What the device object is compiled from: the .hip / .inc files of csrc/ and the public headers."""
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".inc", ".h"))] + _include_headers()


def host_sources():
    """What the host object (csrc/host/engine.cpp) is compiled from."""
    host = os.path.join(CSRC, "host")
    return [os.path.join(host, f) for f in os.listdir(host) if f.endswith((".h", ".cpp"))] + _include_headers()


def sources():
    return sorted(set(device_sources() + host_sources()))


def _stale(target, deps):
    return not os.path.exists(target) or any(os.path.getmtime(s) > os.path.getmtime(target) for s in deps)


def _build(lib, obj_dev, obj_host, extra, force, verbose):
    if not force and not _stale(lib, sources()):
        return lib                                   # up to date, whether or not the intermediate objects travelled with the tree
    # one builder at a time: the ranks of a multi-process job (bench.py --gpus N, the two-rank tests) all call build() — two compilers writing the same
    # object file left a half-written library behind and the second rank died loading it (round 5, on a tree whose sources were newer than its .so)
    import fcntl
    with open(os.path.join(HERE, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not _stale(lib, sources()):
                return lib                           # another process built it while this one waited
            return _build_locked(lib, obj_dev, obj_host, extra, force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(lib, obj_dev, obj_host, extra, force, verbose):
    # each object is rebuilt only when one of ITS sources changed: a host-side change leaves the device object (the kernels) byte for byte as it was
    cmds = []
    if force or _stale(obj_dev, device_sources()):
        cmds.append([HIPCC, "--offload-arch=gfx950", *COMMON, *extra, "-c", os.path.join(CSRC, "infidex_hip.hip"), "-o", obj_dev])
    if force or _stale(obj_host, host_sources()):
        cmds.append([HIPCC, *COMMON, "-march=x86-64-v3", "-x", "c++", "-c", os.path.join(CSRC, "host", "engine.cpp"), "-o", obj_host])
    if cmds or _stale(lib, [obj_dev, obj_host]):
        cmds.append([HIPCC, "--offload-arch=gfx950", "-shared", "-o", lib + ".tmp", obj_dev, obj_host, "-lpthread"])
    for c in cmds:
        if verbose:
            print(" ".join(c), file=sys.stderr)
        subprocess.check_call(c)
    if os.path.exists(lib + ".tmp"):
        os.replace(lib + ".tmp", lib)               # the library appears complete or not at all
    return lib


def build(force=False, verbose=False):
    """The product library: the production kernels only."""
    return _build(LIB, os.path.join(CSRC, "infidex_hip.o"), os.path.join(CSRC, "engine.o"), [], force, verbose)


def build_variant(tag, defines, force=False, verbose=False):
    """libinfidex_hip_<tag>.so = the product compiled with extra -D flags: A/B measurements of one kernel decision on the GPU box (INFX_LIB selects it)."""
    return _build(os.path.join(HERE, f"libinfidex_hip_{tag}.so"), os.path.join(CSRC, f"infidex_hip_{tag}.o"), os.path.join(CSRC, "engine.o"), list(defines), force, verbose)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    for a in sys.argv[1:]:                                   # --variant=tag:-DX=1,-DY=2
        if a.startswith("--variant="):
            tag, _, defs = a[len("--variant="):].partition(":")
            print(build_variant(tag, [d for d in defs.split(",") if d], force="--force" in sys.argv, verbose=True))
