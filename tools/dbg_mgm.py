import sys, os
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from helpers import synth_pair, same
from s2p_amd import _lib as hip
from oracle import pyoracle as oracle
import test_gpu_mgm_bands as t
for (seed,H,W,dmin,dmax,nan,kw) in [c for c in t.SHAPES if not os.environ.get('ONLY') or str(c[0]) in os.environ['ONLY'].split(',')]:
    mid, amp = 0.5 * (dmin + dmax), 0.2 * (dmax - dmin)
    im1, im2 = synth_pair(seed, H, W, lambda x, y: mid + amp * np.sin(x / 23.) * np.cos(y / 19.), nan=nan)
    kw = dict(kw, recursion=1)
    o = oracle.oracle_census_sgm(im1, im2, dmin, dmax, params=oracle.census_params(**kw), dump="full")
    os.environ["S2P_MGM_IMPL"]="bands"
    try:
        r = hip.census_sgm(im1, im2, dmin, dmax, params=hip.default_census_params(**kw), dump="full")
    except Exception as e:
        print(seed, "EXC", e); continue
    bad = (o["S"] != r["S"])
    print(seed, H, W, dmin, dmax, "S diff:", int(bad.sum()), "of", bad.size, "disp same:", same(o["disp"], r["disp"]))
    if bad.any():
        ys,xs,ds = np.nonzero(bad)
        print("   rows", ys.min(), ys.max(), "cols", xs.min(), xs.max(), "first", ys[0], xs[0], ds[0], "d range", ds.min(), ds.max())
