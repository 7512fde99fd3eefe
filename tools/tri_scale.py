"""tools/tri_scale.py -- triangulation kernel time vs tile size: the reference tile upsampled k x k (disparities and
rectifying homographies scaled accordingly, so every pixel still falls inside the image domain)."""
import ctypes, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from helpers import load_golden
from s2p_amd import _lib as L, triangulation as tri
g, m = load_golden("tri_tile"), load_golden("mgm_tile")
r1, r2 = tri.rpc_from_geotiff_tag(g["rpc1"]), tri.rpc_from_geotiff_tag(g["rpc2"])
x, y, w, h = (int(v) for v in g["tile"])
ctx = L.context(0)
for k in (1, 2, 3):
    S = np.diag([k, k, 1.0])
    disp = np.kron(m["disp"], np.ones((k, k), np.float32)) * k
    mask = np.kron(g["mask_rect"], np.ones((k, k), np.uint8))
    args = (r1, r2, S @ g["H_ref"], S @ g["H_sec"], disp, mask, (x, x + w, y, y + h), g["mask_orig"])
    for _ in range(2): out = tri.disp_to_lonlatalt(*args, A=g["A"])
    L.check(L.lib().s2p_hip_timing_enable(ctx, 1)); L.check(L.lib().s2p_hip_timing_reset(ctx))
    for _ in range(5): out = tri.disp_to_lonlatalt(*args, A=g["A"])
    ms, n = ctypes.c_double(), ctypes.c_int()
    L.check(L.lib().s2p_hip_timing_get(ctx, b"triangulate", ctypes.byref(ms), ctypes.byref(n)))
    L.check(L.lib().s2p_hip_timing_enable(ctx, 0))
    npx = int(np.isfinite(out[1]).sum())
    print("x%d: %dx%d, %d triangulated pixels, kernel %.3f ms -> %.0f Mpx/s" % (k, disp.shape[1], disp.shape[0], npx, ms.value / n.value, npx / (ms.value / n.value) / 1e3))
