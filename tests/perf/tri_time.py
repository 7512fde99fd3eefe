"""tests/perf/tri_time.py -- time the triangulation kernel on the reference tile (HIP events) and the CPU reference beside it."""
import ctypes, sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from helpers import load_golden
from oracle import pyoracle as po
from s2p_amd import _lib as L, triangulation as tri
g, m = load_golden("tri_tile"), load_golden("mgm_tile")
r1, r2 = po.rpc_from_geotiff_tag(g["rpc1"]), po.rpc_from_geotiff_tag(g["rpc2"])
x, y, w, h = (int(v) for v in g["tile"])
args = (r1, r2, g["H_ref"], g["H_sec"], m["disp"], g["mask_rect"], (x, x + w, y, y + h), g["mask_orig"])
ctx = L.context(0)
for _ in range(3): tri.disp_to_lonlatalt(*args, A=g["A"])
L.check(L.lib().s2p_hip_timing_enable(ctx, 1)); L.check(L.lib().s2p_hip_timing_reset(ctx))
t = time.perf_counter(); n = 20
for _ in range(n): tri.disp_to_lonlatalt(*args, A=g["A"])
wall = (time.perf_counter() - t) / n
ms, k = ctypes.c_double(), ctypes.c_int()
L.check(L.lib().s2p_hip_timing_get(ctx, b"triangulate", ctypes.byref(ms), ctypes.byref(k)))
npx = int(np.count_nonzero(np.isfinite(tri.disp_to_lonlatalt(*args, A=g["A"])[1])))
print("GPU kernel %.3f ms (%d launches), host call %.3f ms, %d triangulated pixels -> %.1f Mpx/s kernel" % (ms.value / k.value, k.value, wall * 1e3, npx, npx / (ms.value / k.value) / 1e3))
fn = po.ref_disp_to_lonlatalt if po.have_ref_tri() else po.oracle_disp_to_lonlatalt
t = time.perf_counter(); fn(r1, r2, g["H_ref"], g["H_sec"] @ np.linalg.inv(g["A"]), m["disp"], g["mask_rect"], (x, x + w, y, y + h), g["mask_orig"]); c = time.perf_counter() - t
print("CPU %s: %.3f s -> %.3f Mpx/s (1 core)" % ("reference" if po.have_ref_tri() else "port", c, npx / c / 1e6))
# triangulation.height_map (s2p/triangulation.py:346-389): padded triangulation + the scipy resampling, host to host
hargs = (x, y, w, h, r1, r2, g["H_ref"], g["H_sec"], m["disp"], g["mask_rect"], g["mask_orig"])
for _ in range(3): tri.height_map(*hargs, A=g["A"])
t = time.perf_counter()
for _ in range(n): hm = tri.height_map(*hargs, A=g["A"])
hw = (time.perf_counter() - t) / n
lla = tri.disp_to_lonlatalt(r1, r2, g["H_ref"], g["H_sec"], m["disp"], g["mask_rect"], (x - 1, x + w + 2, y - 1, y + h + 2),
                            np.pad(g["mask_orig"], 1, constant_values=1), A=g["A"])[0]
T = np.array([[1.0, 0, x], [0, 1.0, y], [0, 0, 1.0]])
t = time.perf_counter()
for _ in range(5): ref = po.oracle_height_transfer(lla[:, :, 2], np.dot(g["H_ref"], T), w, h)
sc = (time.perf_counter() - t) / 5
print("height_map: HIP %.3f ms host to host (triangulation + resampling); the resampling alone with scipy on one core %.2f ms; identical %s"
      % (hw * 1e3, sc * 1e3, np.array_equal(ref, hm, equal_nan=True)))
