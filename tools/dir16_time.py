"""tools/dir16_time.py -- stage times of the census / MGM matcher with 8 and with 16 directions (the knight's moves: 40 more
lattices under the same band kernel, 16 e-volumes in the WTA), one 1024 x 1024 x 128 tile at a time, and how far the two results are
apart.  (The oracle comparison at this size is tests/test_gpu_jobs.py::test_mgm_mode_equals_the_oracle_at_the_full_tile_shapes.)"""
import ctypes, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from helpers import synth_pair
from s2p_amd import _lib as L
ctx = L.context(0)
h, w, dmin, dmax = 1024, 1024, -64, 63
im1, im2 = synth_pair(7, h, w, lambda x, y: 0.3 * (dmax - dmin) * np.sin(2 * np.pi * x / 512.) * np.cos(2 * np.pi * y / 512.))
res = {}
for nd in (8, 16):
    for rec in (1, 2):
        p = L.default_census_params(recursion=rec, nb_dir=nd)
        for _ in range(2): r = L.census_sgm(im1, im2, dmin, dmax, params=p, want_conf=False)
        res[(nd, rec)] = r
        L.check(L.lib().s2p_hip_timing_enable(ctx, 1)); L.check(L.lib().s2p_hip_timing_reset(ctx))
        for _ in range(5): L.census_sgm(im1, im2, dmin, dmax, params=p, want_conf=False)
        out = {}
        for s in ("cost", "aggregate", "wta", "total"):
            ms, k = ctypes.c_double(), ctypes.c_int()
            L.check(L.lib().s2p_hip_timing_get(ctx, s.encode(), ctypes.byref(ms), ctypes.byref(k)))
            out[s] = round(ms.value / max(k.value, 1), 3)
        L.check(L.lib().s2p_hip_timing_enable(ctx, 0))
        print("%dx%d x %d, %2d directions, recursion %d: %s ms, %.1f %% valid" % (w, h, dmax - dmin + 1, nd, rec, out, 100 * np.isfinite(r["disp"]).mean()), flush=True)
a, b = res[(8, 2)]["disp"], res[(16, 2)]["disp"]
v = np.isfinite(a) & np.isfinite(b)
print("8 vs 16 directions (recursion 2): %.2f %% of the common pixels within 0.5 px, %.1f / %.1f %% valid" % (100 * np.mean(np.abs(a[v] - b[v]) <= 0.5), 100 * np.isfinite(a).mean(), 100 * np.isfinite(b).mean()))
