"""s2p_amd/build.py -- compiles the HIP sources into s2p_amd/lib/libs2p_hip.so (gfx950 only).

hipcc cross-compiles without a GPU, so this runs in the build container; the in-tree .so then
travels to the GPU box.  No JIT cache, no site-packages install.

PROBE BUILDS.  Any extra compiler flag (S2P_HIP_EXTRA_FLAGS="-DS2P_MGM_PF=32 ...": the measurement and tuning switches listed in
csrc/probe_guard.hpp) makes the build a probe build: it is compiled with -DS2P_PROBE_BUILD="<flags>" (the library then says so in
s2p_hip_build_info() and in every error message), its objects live in a directory of their own, and the result is written to
build/variants/libs2p_hip_<S2P_HIP_VARIANT or a hash of the flags>.so -- NEVER to s2p_amd/lib/libs2p_hip.so, which only ever holds
the shipped configuration.  A probe library is selected at run time with S2P_HIP_LIB=<path> (tools/build_variants.sh).
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libs2p_hip.so")
VARIANTS = os.path.join(HERE, "..", "build", "variants")
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


def extra_flags():
    return os.environ.get("S2P_HIP_EXTRA_FLAGS", "").split()


def target(extra=None, variant=None):
    """Where a build with these extra flags goes: the shipped path for none, build/variants/ for a probe build."""
    extra = extra_flags() if extra is None else list(extra)
    if not extra:
        return LIB
    tag = variant or os.environ.get("S2P_HIP_VARIANT") or hashlib.sha1(" ".join(extra).encode()).hexdigest()[:10]
    return os.path.normpath(os.path.join(VARIANTS, "libs2p_hip_%s.so" % tag))


def needs_build():
    return _stale(target(), [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + _headers())


def build(force=False, verbose=False):
    """One object per source (compiled side by side, rebuilt only when the source or a header changed), then one link."""
    if not force and not needs_build():
        return target()
    from concurrent.futures import ThreadPoolExecutor
    extra = extra_flags()
    out = target(extra)
    assert bool(extra) == (os.path.abspath(out) != os.path.abspath(LIB)), "a probe build never lands in s2p_amd/lib/"
    os.makedirs(os.path.dirname(out), exist_ok=True)
    cc, hdrs = hipcc(), _headers()
    cflags = [f for f in FLAGS if f != "-shared"]
    if extra:                                                # the umbrella every probe switch requires (csrc/probe_guard.hpp)
        cflags.append('-DS2P_PROBE_BUILD="%s"' % " ".join(extra).replace('"', "'").replace("\\", ""))
    objdir = os.path.join(HERE, "..", "build", "obj" + ("-" + os.path.basename(out)[len("libs2p_hip_"):-3] if extra else ""))
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
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fvisibility=hidden", "-o", out] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
