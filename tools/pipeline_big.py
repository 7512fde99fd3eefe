"""tools/pipeline_big.py -- steps 3-5 end to end (image windows -> disparity, mask, lon/lat/alt) on a tile of about the
BASELINE size: the reference's own tile upsampled 2 x 2 (windows by pixel replication, homographies conjugated with
the scaling, disparity range doubled), so the geometry stays real.  Host arrays in, host arrays out."""
import sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from helpers import load_golden
from s2p_amd import _lib as L, tiles, triangulation

g1, g2, g3 = load_golden("warp_tile"), load_golden("mgm_tile"), load_golden("tri_tile")
k = 2
S = np.diag([k, k, 1.0])
Si = np.linalg.inv(S)
w, h = (int(v) * k for v in g1["size"])
src1 = np.kron(g1["src"], np.ones((k, k), g1["src"].dtype))
src2 = np.kron(g2["src"], np.ones((k, k), g2["src"].dtype))
H1, H2 = S @ g1["H"] @ Si, S @ g2["H"] @ Si
d_ref = g2["disp"]
dmin, dmax = k * (int(np.floor(np.nanmin(d_ref))) - 4), k * (int(np.ceil(np.nanmax(d_ref))) + 4)
x, y, tw, th = (int(v) for v in g3["tile"])
tri = dict(rpca=triangulation.rpc_from_geotiff_tag(g3["rpc1"]), rpcb=triangulation.rpc_from_geotiff_tag(g3["rpc2"]),
           ha=S @ g3["H_ref"], hb=S @ g3["H_sec"] @ np.linalg.inv(g3["A"]), msk_orig=g3["mask_orig"], bbox=(x, x + tw, y, y + th))
out = L.tile(src1, H1, src2, H2, w, h, dmin, dmax, algo="census", erosion=2, tri=tri, want_rect=False)
print("tile %dx%d, %d disparities, windows %s / %s: %.1f %% of the pixels triangulated, altitude %.0f..%.0f m" % (
    w, h, dmax - dmin + 1, src1.shape, src2.shape, 100 * np.isfinite(out["err"]).mean(), np.nanmin(out["lonlatalt"][..., 2]), np.nanmax(out["lonlatalt"][..., 2])))
for _ in range(2): L.tile(src1, H1, src2, H2, w, h, dmin, dmax, algo="census", erosion=2, tri=tri, want_rect=False, out=out)
t = time.perf_counter(); n = 20
for _ in range(n): L.tile(src1, H1, src2, H2, w, h, dmin, dmax, algo="census", erosion=2, tri=tri, want_rect=False, out=out)
print("one call per tile, one at a time: %.2f ms / tile" % ((time.perf_counter() - t) / n * 1e3))
acc = [0.0]
def sink(job, res): acc[0] += float(res["lonlatalt"][0, 0, 2] == 0)
for fl in (1, 2, 4):
    jobs = [tiles.TileJob(i, src1, H1, src2, H2, w, h, dmin, dmax, erosion=2, tri=tri) for i in range(48)]
    tiles.process_tiles(jobs[:2 * fl], in_flight=fl, sink=sink)
    t = time.perf_counter(); tiles.process_tiles(jobs, in_flight=fl, sink=sink); dt = time.perf_counter() - t
    print("process_tiles, %d in flight: %.2f ms / tile, %.0f tiles/s, %.0f Mpx/s" % (fl, dt / len(jobs) * 1e3, len(jobs) / dt, len(jobs) * w * h / dt / 1e6))
