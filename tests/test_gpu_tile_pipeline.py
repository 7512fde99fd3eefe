"""GPU tests of the one-call tile pipeline (s2p_hip_tile_host: rectify -> match -> mask/erode -> triangulate,
SURVEY.md 8(f) rank 3) on the reference's own tile (tests/golden/warp_tile, mgm_tile, tri_tile = data files of
the reference's tests/data/input_pair and input_triangulation/pair_1):
every intermediate equals, byte for byte, what the separate entry points return when the tile is handed
from step to step through host memory -- the way the reference hands it through files
(s2p/__init__.py:147-159,178-190,213-233) -- and the resulting heights agree with the reference's stored
point cloud."""
import numpy as np
import pytest

from helpers import load_golden, same

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from s2p_amd import _lib
    assert _lib.device_count() > 0, "no MI355X visible: the HIP path has no fallback"
    return _lib


def reference_tile(oracle):
    from s2p_amd.triangulation import RPCStruct
    g1, g2, g3 = load_golden("warp_tile"), load_golden("mgm_tile"), load_golden("tri_tile")
    w, h = (int(v) for v in g1["size"])
    r1, r2 = oracle.rpc_from_geotiff_tag(g3["rpc1"]), oracle.rpc_from_geotiff_tag(g3["rpc2"])
    ra, rb = RPCStruct(), RPCStruct()
    import ctypes
    ctypes.memmove(ctypes.addressof(ra), ctypes.addressof(r1), ctypes.sizeof(ra))
    ctypes.memmove(ctypes.addressof(rb), ctypes.addressof(r2), ctypes.sizeof(rb))
    x, y, tw, th = (int(v) for v in g3["tile"])
    tri = dict(rpca=ra, rpcb=rb, ha=g3["H_ref"], hb=g3["H_sec"] @ np.linalg.inv(g3["A"]),
               msk_orig=g3["mask_orig"], bbox=(x, x + tw, y, y + th))
    d_ref = g2["disp"]
    dmin, dmax = int(np.floor(np.nanmin(d_ref))) - 4, int(np.ceil(np.nanmax(d_ref))) + 4
    return g1, g2, g3, w, h, dmin, dmax, tri


@pytest.mark.parametrize("algo", ["census", "sgbm"])
def test_one_call_equals_step_by_step(hip, oracle, algo):
    from s2p_amd import triangulation
    g1, g2, g3, w, h, dmin, dmax, tri = reference_tile(oracle)
    out = hip.tile(g1["src"], g1["H"], g2["src"], g2["H"], w, h, dmin, dmax, algo=algo, erosion=2, tri=tri)
    # the same tile, one entry point per step, host arrays in between
    rect1 = hip.warp(g1["src"], g1["H"], w, h)
    rect2 = hip.warp(g2["src"], g2["H"], w, h)
    m = (hip.census_sgm if algo == "census" else hip.sgbm)(rect1, rect2, dmin, dmax)
    mask = hip.erode_mask(m["mask"], 2)
    lla, err = triangulation.disp_to_lonlatalt(tri["rpca"], tri["rpcb"], g3["H_ref"], g3["H_sec"], m["disp"], mask,
                                               tri["bbox"], g3["mask_orig"], A=g3["A"])
    assert same(out["rect1"], rect1) and same(out["rect2"], rect2)
    assert same(out["disp"], m["disp"]) and np.array_equal(out["mask"], mask)
    assert same(out["lonlatalt"], lla) and same(out["err"], err)
    assert np.isfinite(out["lonlatalt"]).any()


def test_heights_agree_with_the_reference_point_cloud(hip, oracle):
    """End to end from the two image windows: altitudes against the reference's stored triangulation of ITS mgm
    disparity map (every 4th pixel).  Different matcher (census/SGM stand-in), same geometry: the medians agree
    to centimetres, 95 % of the common pixels to a metre."""
    g1, g2, g3, w, h, dmin, dmax, tri = reference_tile(oracle)
    out = hip.tile(g1["src"], g1["H"], g2["src"], g2["H"], w, h, dmin, dmax, algo="census", erosion=0, tri=tri, want_rect=False)
    alt, ref = out["lonlatalt"][::4, ::4, 2], g3["lonlatalt_4"][:, :, 2]
    both = np.isfinite(alt) & np.isfinite(ref)
    assert both.mean() > 0.5
    e = np.abs(alt[both] - ref[both])
    print("altitude |err| vs reference cloud: median %.3f m, p95 %.3f m, common %.3f" % (np.median(e), np.percentile(e, 95), both.mean()))
    assert np.median(e) < 0.25 and np.percentile(e, 95) < 1.5
    lonlat = np.abs(out["lonlatalt"][::4, ::4, :2][both] - g3["lonlatalt_4"][:, :, :2][both])
    assert np.percentile(lonlat, 95) < 2e-5                      # ~2 m in degrees


def test_without_triangulation_and_bad_arguments(hip, oracle):
    g1, g2, g3, w, h, dmin, dmax, tri = reference_tile(oracle)
    out = hip.tile(g1["src"], g1["H"], g2["src"], g2["H"], w, h, dmin, dmax, algo="census")
    assert "lonlatalt" not in out and np.isfinite(out["disp"]).mean() > 0.5 and set(np.unique(out["mask"])) <= {0, 1}
    with pytest.raises(hip.HipError) as e:
        hip.tile(g1["src"], np.zeros((3, 3)), g2["src"], g2["H"], w, h, dmin, dmax)      # singular homography
    assert e.value.code == hip.BAD_ARGUMENT
    with pytest.raises(hip.HipError) as e:
        hip.tile(g1["src"], g1["H"], g2["src"], g2["H"], w, h, 5, 4)                      # empty range
    assert e.value.code == hip.EMPTY_RANGE
    assert hip.tile(g1["src"], g1["H"], g2["src"], g2["H"], w, h, dmin, dmax)["disp"].shape == (h, w)   # context still usable


def test_process_tiles_scheduler(hip, oracle):
    """tiles.process_tiles: several TileJobs in flight on separate streams give the same bytes as one at a time."""
    from s2p_amd import tiles
    g1, g2, g3, w, h, dmin, dmax, tri = reference_tile(oracle)
    jobs = [tiles.TileJob(i, g1["src"], g1["H"], g2["src"], g2["H"], w, h, dmin, dmax, erosion=i % 3, tri=tri if i % 2 else None)
            for i in range(6)]
    serial = tiles.process_tiles(jobs, algo="mgm", in_flight=1)
    par = tiles.process_tiles(jobs, algo="mgm", in_flight=3)
    assert sorted(par) == list(range(6))
    for i in range(6):
        assert sorted(par[i]) == sorted(serial[i])
        for k in serial[i]:
            assert same(np.asarray(par[i][k], np.float64), np.asarray(serial[i][k], np.float64)), (i, k)
    assert "lonlatalt" in par[1] and "lonlatalt" not in par[0]


def test_process_tiles_sink_recycles_buffers(hip, oracle):
    """Streaming mode: results are handed to a sink and their buffers reused; what the sink sees equals the collected results."""
    from s2p_amd import tiles
    g1, g2, g3, w, h, dmin, dmax, tri = reference_tile(oracle)
    jobs = [tiles.TileJob(i, g1["src"], g1["H"], g2["src"], g2["H"], w, h, dmin + (i % 2), dmax, erosion=i % 3, tri=tri) for i in range(8)]
    want = tiles.process_tiles(jobs, in_flight=1)
    seen = {}

    def sink(job, res):
        seen[job.index] = {k: v.copy() for k, v in res.items()}
    ret = tiles.process_tiles(jobs, in_flight=2, sink=sink)
    assert all(v is None for v in ret.values()) and sorted(seen) == list(range(8))
    for i in range(8):
        for k in want[i]:
            assert same(np.asarray(seen[i][k], np.float64), np.asarray(want[i][k], np.float64)), (i, k)
