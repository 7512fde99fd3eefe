"""s2p_amd/build.py -- compiles the HIP sources into s2p_amd/lib/libs2p_hip.so (gfx950 only).

hipcc cross-compiles without a GPU, so this runs in the build container; the in-tree .so then
travels to the GPU box.  No JIT cache, no site-packages install.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libs2p_hip.so")
SOURCES = ["api.hip", "sgbm_kernels.hip", "census_kernels.hip", "warp_kernels.hip", "tri_kernels.hip", "fusion_kernels.hip", "raster_kernels.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-Wno-unused-value", "-fvisibility=hidden"]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _headers():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")] + [os.path.join(HERE, "..", "include", "s2p_hip.h")]


def needs_build():
    return _stale(LIB, [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + _headers())


def build(force=False, verbose=False):
    """One object per source (compiled side by side, rebuilt only when the source or a header changed), then one link."""
    if not force and not needs_build():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(LIBDIR, exist_ok=True)
    cc, hdrs = hipcc(), _headers()
    cflags = [f for f in FLAGS if f != "-shared"]
    extra = os.environ.get("S2P_HIP_EXTRA_FLAGS", "").split()          # probe builds (-DS2P_MGM_TRACE ...) keep their own objects
    objdir = os.path.join(HERE, "..", "build", "obj" + ("-%08x" % (hash(tuple(extra)) & 0xffffffff) if extra else ""))
    os.makedirs(objdir, exist_ok=True)

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        path = os.path.join(CSRC, src)
        if force or _stale(obj, [path] + hdrs):
            cmd = [cc] + cflags + extra + ["-c", path, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.run(cmd, check=True)
        return obj
    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fvisibility=hidden", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(LIB)
