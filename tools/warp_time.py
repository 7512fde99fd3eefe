"""tools/warp_time.py -- time the resampler (HIP events) on the reference tile and on a 1024-tile-sized window."""
import ctypes, sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from helpers import load_golden
from s2p_amd import _lib as L
g = load_golden("warp_tile")
ctx = L.context(0)
rng = np.random.default_rng(0)
big = rng.integers(0, 2000, (1200, 1200)).astype(np.uint16)
Hb = np.array([[0.999, -0.02, -40.0], [0.02, 0.999, -50.0], [0, 0, 1.0]])
for name, src, H, w, h in (("reference tile 613x553 -> 503x425", g["src"], g["H"], 503, 425), ("1200x1200 -> 1088x1024", big, Hb, 1088, 1024)):
    for _ in range(3): L.warp(src, H, w, h)
    L.check(L.lib().s2p_hip_timing_enable(ctx, 1)); L.check(L.lib().s2p_hip_timing_reset(ctx))
    t = time.perf_counter(); n = 20
    for _ in range(n): L.warp(src, H, w, h)
    wall = (time.perf_counter() - t) / n
    ms, k = ctypes.c_double(), ctypes.c_int()
    L.check(L.lib().s2p_hip_timing_get(ctx, b"warp", ctypes.byref(ms), ctypes.byref(k)))
    L.check(L.lib().s2p_hip_timing_enable(ctx, 0))
    print("%s: kernels %.3f ms, host call %.3f ms" % (name, ms.value / k.value, wall * 1e3))
