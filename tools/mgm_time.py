"""tools/mgm_time.py -- the census matcher with MGM's two-predecessor recursion (recursion = 1; one band-pipelined
launch, and the front-by-front implementation kept as cross-check) against the default 8-path mode."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from helpers import synth_pair
from s2p_amd import _lib as L
ctx = L.context(0)
for (h, w, dmin, dmax) in ((425, 503, -45, 34), (1024, 1024, -64, 63)):
    im1, im2 = synth_pair(7, h, w, lambda x, y: 0.3 * (dmax - dmin) * np.sin(2 * np.pi * x / 512.) * np.cos(2 * np.pi * y / 512.))
    for rec, impl in ((0, ""), (1, "bands"), (1, "steps")):
        os.environ["S2P_MGM_IMPL"] = impl
        p = L.default_census_params(recursion=rec)
        for _ in range(2): L.census_sgm(im1, im2, dmin, dmax, params=p, want_conf=False)
        L.check(L.lib().s2p_hip_timing_enable(ctx, 1)); L.check(L.lib().s2p_hip_timing_reset(ctx))
        n = 5
        for _ in range(n): L.census_sgm(im1, im2, dmin, dmax, params=p, want_conf=False)
        out = {}
        for s in ("cost", "aggregate", "wta", "total"):
            ms, k = ctypes.c_double(), ctypes.c_int()
            L.check(L.lib().s2p_hip_timing_get(ctx, s.encode(), ctypes.byref(ms), ctypes.byref(k)))
            out[s] = round(ms.value / max(k.value, 1), 3)
        L.check(L.lib().s2p_hip_timing_enable(ctx, 0))
        print("%dx%d, %d disparities, recursion %d%s: %s ms" % (w, h, dmax - dmin + 1, rec, " (%s)" % impl if impl else "", out))
