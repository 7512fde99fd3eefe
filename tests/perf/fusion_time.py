"""tests/perf/fusion_time.py -- fusion.merge_n: the reference's np.apply_along_axis formulation (oracle, one host core)
against s2p_hip_merge_n_host on the same stack."""
import sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from oracle import pyoracle
from s2p_amd import _lib as L
rng = np.random.default_rng(0)
for (h, w, n) in ((350, 350, 2), (1024, 1024, 3)):
    base = rng.uniform(0, 100, (h, w))
    st = [(base + rng.normal(0, 1, (h, w))).astype(np.float32) for _ in range(n)]
    for a in st: a[rng.uniform(size=a.shape) < 0.2] = np.nan
    offs = [0.1 * i for i in range(n)]
    L.merge_n(st, offs, threshold=3)
    t = time.perf_counter(); k = 10
    for _ in range(k): g = L.merge_n(st, offs, threshold=3)
    tg = (time.perf_counter() - t) / k
    t = time.perf_counter(); c = pyoracle.oracle_merge_n(st, offs, "average_if_close", 3); tc = time.perf_counter() - t
    print("%dx%d x %d maps: numpy apply_along_axis %.2f s, HIP (host arrays in/out) %.2f ms, equal: %s" % (w, h, n, tc, tg * 1e3, np.array_equal(g, c, equal_nan=True)))
