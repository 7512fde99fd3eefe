"""tools/shim_time.py -- end to end through the file-level drop-in (s2p_amd.block_matching.compute_disparity_map:
TIFF decode -> one library call -> TIFF/PNG encode), the way the untouched orchestrator would call it."""
import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from helpers import synth_pair
from s2p_amd import block_matching, io as rio
im1, im2 = synth_pair(1000, 1024, 1024, lambda x, y: 40 * np.sin(2 * np.pi * x / 512.) * np.cos(2 * np.pi * y / 512.))
d = tempfile.mkdtemp()
p1, p2 = os.path.join(d, "rectified_ref.tif"), os.path.join(d, "rectified_sec.tif")
rio.write_image(p1, im1); rio.write_image(p2, im2)
for algo in ("mgm", "mgm_multi", "sgbm"):
    disp, mask = os.path.join(d, "disp_%s.tif" % algo), os.path.join(d, "mask_%s.png" % algo)
    so = sys.stdout; sys.stdout = open(os.devnull, "w")
    try:
        for _ in range(2): block_matching.compute_disparity_map(p1, p2, disp, mask, algo, -64, 63)
        t = time.perf_counter(); n = 10
        for _ in range(n): block_matching.compute_disparity_map(p1, p2, disp, mask, algo, -64, 63)
        dt = (time.perf_counter() - t) / n
    finally:
        sys.stdout = so
    print("compute_disparity_map('%s') on 1024x1024 float32 TIFFs, 128 disparities: %.1f ms per call (files in, files out; rasterio: %s)" % (algo, dt * 1e3, rio.HAVE_RASTERIO))
