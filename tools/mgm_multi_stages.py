"""tools/mgm_multi_stages.py [size ndisp] -- (GPU box) one 'mgm_multi' matcher call on a synthetic tile, resident buffers:
wall time per call and the library's per-stage event times (all pyramid levels summed)."""
import ctypes, sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from helpers import synth_pair
from s2p_amd import _lib as L, block_matching
size = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
nd = int(sys.argv[2]) if len(sys.argv) > 2 else 256
im1, im2 = synth_pair(5, size, size, lambda x, y: 0.3 * nd * np.sin(2 * np.pi * x / 512.) * np.cos(2 * np.pi * y / 512.))
import os
from s2p_amd.config import cfg
variants = [("mgm", {}), ("mgm_multi", {}), ("mgm_multi", {"hip_mgm_multi_subpix": 1}), ("mgm_multi", {"hip_mgm_multi_scales": 1}),
            ("mgm_multi", {"hip_mgm_multi_recursion": 0})]
for algo, over in variants:
    c = dict(cfg); c.update(over)
    kind, p = block_matching.matcher_params(algo, c)
    os.environ["S2P_MS_DEBUG"] = "1"
    L.census_sgm(im1, im2, -nd // 2, nd // 2 - 1, params=p, want_conf=False)
    del os.environ["S2P_MS_DEBUG"]
    ctx = L.context()
    lib = L.lib()
    for rep in range(3):
        L.census_sgm(im1, im2, -nd // 2, nd // 2 - 1, params=p, ctx=ctx, want_conf=False)
    lib.s2p_hip_timing_enable(ctx, 1)
    t = time.perf_counter(); n = 10
    for rep in range(n):
        L.census_sgm(im1, im2, -nd // 2, nd // 2 - 1, params=p, ctx=ctx, want_conf=False)
    dt = (time.perf_counter() - t) / n * 1e3
    out = {}
    for st in ("cost", "aggregate", "wta", "median", "speckle", "epilogue", "total"):
        ms, k = ctypes.c_double(), ctypes.c_int()
        if lib.s2p_hip_timing_get(ctx, st.encode(), ctypes.byref(ms), ctypes.byref(k)) == 0 and k.value:
            out[st] = (round(ms.value / n, 3), k.value // n)
    print("%s %s %dx%d x %d: %.2f ms per host call; stages (ms, launches per call): %s" % (algo, over, size, size, nd, dt, out), flush=True)
