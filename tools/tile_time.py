"""tools/tile_time.py -- one tile through steps 3-5 (rectify, match, mask/erode, triangulate): the one-call
pipeline (s2p_hip_tile_host) against one entry point per step with host arrays in between, on the reference's
own tile (tests/golden) and with several tiles in flight."""
import sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from helpers import load_golden
from s2p_amd import _lib as L, tiles, triangulation

g1, g2, g3 = load_golden("warp_tile"), load_golden("mgm_tile"), load_golden("tri_tile")
w, h = (int(v) for v in g1["size"])
r1, r2 = triangulation.rpc_from_geotiff_tag(g3["rpc1"]), triangulation.rpc_from_geotiff_tag(g3["rpc2"])
ra, rb = r1, r2
x, y, tw, th = (int(v) for v in g3["tile"])
tri = dict(rpca=ra, rpcb=rb, ha=g3["H_ref"], hb=g3["H_sec"] @ np.linalg.inv(g3["A"]), msk_orig=g3["mask_orig"], bbox=(x, x + tw, y, y + th))
d_ref = g2["disp"]
dmin, dmax = int(np.floor(np.nanmin(d_ref))) - 4, int(np.ceil(np.nanmax(d_ref))) + 4


def fused():
    return L.tile(g1["src"], g1["H"], g2["src"], g2["H"], w, h, dmin, dmax, algo="census", erosion=2, tri=tri, want_rect=False)


def steps():
    a = L.warp(g1["src"], g1["H"], w, h); b = L.warp(g2["src"], g2["H"], w, h)
    m = L.census_sgm(a, b, dmin, dmax, want_conf=False, params=L.default_census_params(recursion=0))
    mask = L.erode_mask(m["mask"], 2)
    return triangulation.disp_to_lonlatalt(ra, rb, g3["H_ref"], g3["H_sec"], m["disp"], mask, tri["bbox"], g3["mask_orig"], A=g3["A"])


for name, f in (("one call (s2p_hip_tile_host)", fused), ("one entry point per step", steps)):
    for _ in range(3): f()
    t = time.perf_counter(); n = 30
    for _ in range(n): f()
    print("%-32s %.2f ms / tile (%dx%d, %d disparities)" % (name, (time.perf_counter() - t) / n * 1e3, w, h, dmax - dmin + 1))
for k in (1, 2, 4):
    jobs = [tiles.TileJob(i, g1["src"], g1["H"], g2["src"], g2["H"], w, h, dmin, dmax, erosion=2, tri=tri) for i in range(48)]
    tiles.process_tiles(jobs[:2 * k], in_flight=k)
    t = time.perf_counter()
    tiles.process_tiles(jobs, in_flight=k)
    dt = time.perf_counter() - t
    print("process_tiles, %d in flight: %.2f ms / tile, %.0f tiles/s" % (k, dt / len(jobs) * 1e3, len(jobs) / dt))
acc = [0.0]
def sink(job, res):
    acc[0] += float(np.nansum(res["lonlatalt"][::64, ::64, 2]))      # a consumer that looks at the result
for k in (1, 2, 4):
    jobs = [tiles.TileJob(i, g1["src"], g1["H"], g2["src"], g2["H"], w, h, dmin, dmax, erosion=2, tri=tri) for i in range(96)]
    tiles.process_tiles(jobs[:2 * k], in_flight=k, sink=sink)
    t = time.perf_counter()
    tiles.process_tiles(jobs, in_flight=k, sink=sink)
    dt = time.perf_counter() - t
    print("process_tiles with a sink (recycled buffers), %d in flight: %.2f ms / tile, %.0f tiles/s" % (k, dt / len(jobs) * 1e3, len(jobs) / dt))
