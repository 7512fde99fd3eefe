"""tools/shim_breakdown.py -- where the milliseconds of one file-level compute_disparity_map('mgm') call go."""
import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from helpers import synth_pair
from s2p_amd import block_matching, io as rio, _lib
im1, im2 = synth_pair(1000, 1024, 1024, lambda x, y: 40 * np.sin(2 * np.pi * x / 512.) * np.cos(2 * np.pi * y / 512.))
d = tempfile.mkdtemp()
p1, p2 = os.path.join(d, "a.tif"), os.path.join(d, "b.tif")
rio.write_image(p1, im1); rio.write_image(p2, im2)
kind, p = block_matching.matcher_params("mgm")


def t(f, n=20):
    f(); f()
    t0 = time.perf_counter()
    for _ in range(n): r = f()
    return (time.perf_counter() - t0) / n * 1e3, r


ms, _ = t(lambda: rio.image_size(p1)); print("image_size %.2f ms" % ms)
ms, ab = t(lambda: rio.read_images([p1, p2])); print("read_images %.2f ms" % ms)
ms, _ = t(lambda: [rio.read_image(p1), rio.read_image(p2)]); print("read serial %.2f ms" % ms)
a, b = ab
ms, r = t(lambda: _lib.census_sgm(a, b, -64, 63, params=p)); print("census_sgm host pageable %.2f ms" % ms)
ms, r2 = t(lambda: _lib.census_sgm(a, b, -64, 63, params=p, want_conf=False)); print("  without conf %.2f ms" % ms)
try:
    import torch
    pa = torch.empty(a.shape, dtype=torch.float32, pin_memory=True).numpy(); pa[:] = a
    pb = torch.empty(a.shape, dtype=torch.float32, pin_memory=True).numpy(); pb[:] = b
    ms, _ = t(lambda: _lib.census_sgm(pa, pb, -64, 63, params=p)); print("census_sgm pinned inputs (outputs pageable) %.2f ms" % ms)
except Exception as e:
    print("no torch pin", e)
outs = [(os.path.join(d, "d.tif"), r["disp"]), (os.path.join(d, "c.tif"), r["conf"]), (os.path.join(d, "m.png"), r["mask"])]
ms, _ = t(lambda: rio.write_images(outs)); print("write_images %.2f ms" % ms)
for pth, arr in outs:
    ms, _ = t(lambda: rio.write_image(pth, arr)); print("  write %s %.2f ms" % (os.path.basename(pth), ms))
ms, _ = t(lambda: np.empty((1024, 1024), np.float32)); print("np.empty 4MB %.3f ms" % ms)
ms, _ = t(lambda: np.zeros((1024, 1024), np.float32) + 1); print("touch 4MB %.3f ms" % ms)
