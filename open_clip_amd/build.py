"""Builds ``libopenclip_hip.so`` (all HIP kernels + the C ABI of include/openclip_hip.h) for gfx950.

hipcc cross-compiles without a GPU; objects are cached under ``open_clip_amd/csrc/_build`` and the shared
library is written IN-TREE next to this file so that it travels to the GPU box with the repo snapshot.
"""
import concurrent.futures
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_build")
LIB = os.path.join(HERE, "libopenclip_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (needed to build libopenclip_hip.so)")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, dev: bool = False) -> str:
    """``dev=True`` builds ``libopenclip_hip_dev.so`` with -DOCN_DEV_BUILD: the same library plus the in-kernel developer knobs of
    include/openclip_hip_debug.h (ablation bits, per-tile timeline, cache-policy flips) that tools/ select through ``OCN_LIB_PATH``;
    the product library compiles none of them."""
    if dev:
        return _build(force, verbose, os.path.join(CSRC, "_build", "dev"), os.path.join(HERE, "libopenclip_hip_dev.so"), FLAGS + ["-DOCN_DEV_BUILD"])
    return _build(force, verbose, OBJ, LIB, FLAGS)


def _build(force, verbose, OBJ, LIB, FLAGS):
    """one library from every csrc/*.hip: object directory, output file and flags are arguments (no module state is touched: re-entrant)"""
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(os.path.dirname(HERE), "include", "openclip_hip.h"), os.path.abspath(__file__)]
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    jobs = []
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s)[:-4] + ".o")
        if force or _stale(o, [s] + hdrs):
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        cmd = [hipcc] + FLAGS + ["-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (s, r.stderr[-4000:]))
        return s

    if jobs:
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for s in ex.map(compile_one, jobs):
                if verbose:
                    print("compiled", os.path.basename(s))
    objs = [os.path.join(OBJ, os.path.basename(s)[:-4] + ".o") for s in srcs]
    if force or jobs or _stale(LIB, objs):
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr[-4000:])
        if verbose:
            print("linked", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, dev="--dev" in sys.argv))
