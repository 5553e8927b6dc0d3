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
OBJDIR = os.path.join(LIBDIR, "obj")
ARCH = "gfx950"
DROPIN_NAMES = ["flexspmm.so", "cuspmm.so", "tile.so", "permutate.so", "renumber.so"]

SOURCES = ["spmm_kernels.hip", "spmm_narrow.hip", "spmm_quad.hip", "spmm_group.hip", "spmm_panel.hip", "slicing.hip", "reorder_device.hip", "rabbit_device.hip", "exchange.hip", "plan_policy.cpp", "plan_build.cpp", "api_spmm.cpp", "api_reorder.cpp", "api_dropin.cpp", "reorder.cpp"]
HEADERS = ["spmm_kernels.h", "plan.h", "plan_policy.h", "philox.h", "reorder.h", os.path.join(INCLUDE, "gcn_spmm.h")]


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


def _digest(paths):
    """content hash of a unit's inputs: the stamp beside every object (mtimes do not survive a snapshot copy)"""
    import hashlib
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _compile_one(args):
    src, obj, verbose = args
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++20", "-fPIC", "-fvisibility=default", "-Wall",
           "-Wno-unused-result", "-x", "hip", "-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return obj


def build(force=False, verbose=False):
    """One object per translation unit (no relocatable device code: the units share headers only), compiled in
    parallel and re-compiled only when the unit or a header changed; then one link."""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    os.makedirs(DROPIN, exist_ok=True)
    hdrs = [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    me = os.path.abspath(__file__)
    jobs, objs, stamps = [], [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJDIR, os.path.splitext(s)[0] + ".o")
        objs.append(obj)
        want = _digest([src, me] + hdrs)
        stamp = obj + ".sha"
        have = open(stamp).read().strip() if os.path.exists(stamp) else ""
        if force or not os.path.exists(obj) or have != want:
            jobs.append((src, obj, verbose))
            stamps.append((stamp, want))
    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), max(1, (os.cpu_count() or 2) - 1))) as ex:
            list(ex.map(_compile_one, jobs))
        for stamp, want in stamps:
            with open(stamp, "w") as f:
                f.write(want)
    if force or jobs or not os.path.exists(LIB):
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", LIB]
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
