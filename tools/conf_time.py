"""tools/conf_time.py -- census matcher stage times with and without the confidence image (device-resident arrays are not
needed for stage timing: HIP events bracket the kernels only)."""
import ctypes, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from helpers import synth_pair
from s2p_amd import _lib as L
im1, im2 = synth_pair(7, 1024, 1024, lambda x, y: 40 * np.sin(2 * np.pi * x / 512.) * np.cos(2 * np.pi * y / 512.))
ctx = L.context(0)
for conf in (False, True):
    for _ in range(2): L.census_sgm(im1, im2, -64, 63, want_conf=conf, params=L.default_census_params(recursion=0))
    L.check(L.lib().s2p_hip_timing_enable(ctx, 1)); L.check(L.lib().s2p_hip_timing_reset(ctx))
    for _ in range(5): L.census_sgm(im1, im2, -64, 63, want_conf=conf, params=L.default_census_params(recursion=0))
    out = {}
    for s in ("cost", "aggregate", "wta", "total"):
        ms, n = ctypes.c_double(), ctypes.c_int()
        L.check(L.lib().s2p_hip_timing_get(ctx, s.encode(), ctypes.byref(ms), ctypes.byref(n)))
        out[s] = round(ms.value / max(n.value, 1), 4)
    L.check(L.lib().s2p_hip_timing_enable(ctx, 0))
    print("confidence image:", conf, out)
