"""tests/perf/raster_time.py -- DSM rasterisation: the HIP path (host arrays in and out) next to the CPU restatement
of plyflatten's C core (oracle/rasterize_oracle.c, one host core), on clouds of the size of one s2p tile and of a
whole-ROI DSM."""
import sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from s2p_amd import _lib as L
from oracle import pyoracle as po

for n, size, res, radius in ((120764, 477, 0.4, 0), (1_000_000, 1024, 0.5, 0), (4_000_000, 2048, 0.5, 0), (1_000_000, 1024, 0.5, 2)):
    rng = np.random.default_rng(1)
    cloud = np.column_stack([rng.uniform(0, size * res, n), rng.uniform(-size * res, 0, n), rng.normal(2300, 30, n),
                             rng.integers(0, 255, (n, 3)).astype(float)])
    for _ in range(2):
        r = L.plyflatten(cloud, 0.0, 0.0, res, size, size, radius=radius)
    t = time.perf_counter(); k = 5
    for _ in range(k):
        r = L.plyflatten(cloud, 0.0, 0.0, res, size, size, radius=radius)
    g = (time.perf_counter() - t) / k
    t = time.perf_counter()
    o = po.oracle_plyflatten(cloud, 0.0, 0.0, res, size, size, radius=radius)
    c = time.perf_counter() - t
    same = np.array_equal(o, r, equal_nan=True)
    print("%8d points -> %4d^2 x 4 bands, radius %d: HIP %.2f ms host to host (%.0f Mpoints/s), CPU port %.1f ms, identical %s"
          % (n, size, radius, g * 1e3, n / g / 1e6, c * 1e3, same), flush=True)
