"""tools/pcie_rate.py -- host-buffer (PCIe-inclusive) rate of the *_host entry points, for DESIGN.md."""
import sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from helpers import synth_pair
from s2p_amd import _lib as L
im1, im2 = synth_pair(1000, 1024, 1024, lambda x, y: 40 * np.sin(2 * np.pi * x / 512.) * np.cos(2 * np.pi * y / 512.))
for name, fn in (("census", lambda: L.census_sgm(im1, im2, -64, 63, want_conf=False, params=L.default_census_params(recursion=0))), ("sgbm", lambda: L.sgbm(im1, im2, -64, 64, want_cost=False))):
    for _ in range(3): fn()
    t = time.perf_counter(); n = 20
    for _ in range(n): fn()
    dt = (time.perf_counter() - t) / n
    print("%s host-buffer call: %.3f ms/tile  (%.1f Gdisp/s incl. H2D 8 MB + D2H 5 MB, pageable numpy buffers)" % (name, dt * 1e3, 1024 * 1024 * 128 / dt / 1e9))
