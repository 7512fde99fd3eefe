"""tools/mgm_big.py [H W] -- MGM mode on a tile with more band workgroups than the chip can hold at once
(3000 x 3000 x 128: 12 lattices x 94 bands of 32 rows, two bands per CU): the ticketed band order must make progress without
all workgroups being resident.  Compares the one-launch kernel with the front-by-front one (bit-exact) and times both."""
import os, sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from s2p_amd import _lib as L
from helpers import synth_pair, same

H = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
im1, im2 = synth_pair(11, H, W, lambda x, y: 40 * np.sin(2 * np.pi * x / 700.) * np.cos(2 * np.pi * y / 900.))
p = L.default_census_params(recursion=1)
out = {}
for impl in ("steps", "bands"):
    os.environ["S2P_MGM_IMPL"] = impl
    L.census_sgm(im1, im2, -64, 63, params=p, want_conf=False)          # workspace allocation
    t = time.perf_counter()
    out[impl] = L.census_sgm(im1, im2, -64, 63, params=p, want_conf=False)
    print("%s: %dx%d host call %.1f ms" % (impl, H, W, (time.perf_counter() - t) * 1e3), flush=True)
ok = same(out["steps"]["disp"], out["bands"]["disp"]) and same(out["steps"]["mask"], out["bands"]["mask"])
print("bands == steps:", ok, "valid fraction %.3f" % np.isfinite(out["bands"]["disp"]).mean())
sys.exit(0 if ok else 1)
