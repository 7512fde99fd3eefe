"""tools/mgm_diff.py -- where does the band-pipelined MGM kernel differ from the front-by-front one?  (GPU box)
Prints the SOURCES of the mismatch: failing pixels none of whose predecessors (in any of the 8 directions'
dependency graphs restricted to the failing set) fail, per assumed direction."""
import os, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from s2p_amd import _lib as L
from helpers import synth_pair

DIRS = [(1, 0), (-1, 0), (0, 1), (0, -1), (1, 1), (-1, 1), (-1, -1), (1, -1)]
SHAPES = [(320, 131, 257, -24, 40, False), (322, 131, 257, -32, 31, False), (323, 67, 129, -32, 31, False)]
for seed, H, W, dmin, dmax, nan in SHAPES:
    mid, amp = 0.5 * (dmin + dmax), 0.2 * (dmax - dmin)
    im1, im2 = synth_pair(seed, H, W, lambda x, y: mid + amp * np.sin(x / 23.) * np.cos(y / 19.), nan=nan)
    p = L.default_census_params(recursion=1)
    os.environ["S2P_MGM_IMPL"] = "steps"
    ref = L.census_sgm(im1, im2, dmin, dmax, params=p, dump="full")["S"]
    os.environ["S2P_MGM_IMPL"] = "bands"
    got = L.census_sgm(im1, im2, dmin, dmax, params=p, dump="full")["S"]
    badpx = np.any(ref != got, axis=2)
    print(seed, "%dx%d D=%d: %d failing pixels" % (H, W, ref.shape[2], badpx.sum()))
    if not badpx.any():
        continue
    ys, xs = np.nonzero(badpx)
    for r, (dx, dy) in enumerate(DIRS):
        ex, ey = -dy, dx
        src = []
        for y, x in zip(ys, xs):
            ok = True
            for (px, py) in ((x - dx, y - dy), (x - ex, y - ey)):
                if 0 <= px < W and 0 <= py < H and badpx[py, px]:
                    ok = False
            if ok:
                src.append((int(y), int(x)))
        print("   r=%d: %d sources" % (r, len(src)), src[:24])
