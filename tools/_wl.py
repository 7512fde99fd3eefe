import sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from helpers import synth_pair
from s2p_amd import _lib as L
rng = np.random.default_rng(0)
big = rng.integers(0, 2000, (1200, 1200)).astype(np.uint16)
Hb = np.array([[0.999, -0.02, -40.0], [0.02, 0.999, -50.0], [0, 0, 1.0]])
a, b = synth_pair(1, 1024, 1024, lambda x, y: -20 + 0 * x)
a = a.astype(np.float32); b = b.astype(np.float32)
for i in range(30):
    L.census_sgm(a, b, -64, 63)
    L.warp(big, Hb, 1088, 1024)
