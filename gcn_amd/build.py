"""In-tree build of the native library (hipcc cross-compiles gfx950 without a GPU).

    python -m gcn_amd.build [--force]

Produces  gcn_amd/lib/libgcnspmm.so  and the five drop-in copies under
gcn_amd/dropin/ that carry the file names gcn6.py loads (pygcn/gcn6.py:21-25).
The .so files are git-ignored but travel with the gpurun snapshot.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
DROPIN = os.path.join(HERE, "dropin")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(LIBDIR, "libgcnspmm.so")
ARCH = "gfx950"
DROPIN_NAMES = ["flexspmm.so", "cuspmm.so", "tile.so", "permutate.so", "renumber.so"]

SOURCES = ["spmm_kernels.hip", "spmm_narrow.hip", "spmm_quad.hip", "spmm_group.hip", "spmm_panel.hip", "slicing.hip", "reorder_device.hip", "api_spmm.cpp", "api_reorder.cpp", "api_dropin.cpp", "reorder.cpp"]
HEADERS = ["spmm_kernels.h", "plan.h", "philox.h", "reorder.h", os.path.join(INCLUDE, "gcn_spmm.h")]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, /opt/rocm/bin/hipcc, PATH)")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(DROPIN, exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    hdrs = [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    if force or _stale(LIB, srcs + hdrs + [os.path.abspath(__file__)]):
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++20", "-fPIC", "-shared",
               "-fvisibility=default", "-Wall", "-Wno-unused-result",
               "-x", "hip", *srcs, "-o", LIB]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    for name in DROPIN_NAMES:
        dst = os.path.join(DROPIN, name)
        if force or _stale(dst, [LIB]):
            shutil.copy2(LIB, dst)
    return LIB


if __name__ == "__main__":
    path = build(force="--force" in sys.argv, verbose=True)
    print(path)
