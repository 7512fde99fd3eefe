"""GPU parity tests of the triangulation kernel through the C ABI: lon/lat/alt bit-exact against the
reference's own disp_to_lonlatalt (golden subsample; live library when it travelled) and the oracle; the
residual `err` goes through hypot() (libm on the CPU, OCML on the GPU): compared to 1 float32 ulp."""
import ctypes

import numpy as np
import pytest

from helpers import load_golden, same

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from s2p_amd import _lib
    assert _lib.device_count() > 0
    return _lib


def tile_inputs(oracle):
    g = load_golden("tri_tile")
    m = load_golden("mgm_tile")
    r1, r2 = oracle.rpc_from_geotiff_tag(g["rpc1"]), oracle.rpc_from_geotiff_tag(g["rpc2"])
    x, y, w, h = (int(v) for v in g["tile"])
    return g, m, r1, r2, (x, x + w, y, y + h)


def err_close(a, b):
    return np.array_equal(np.isnan(a), np.isnan(b)) and np.all(np.abs(a - b)[np.isfinite(a)] <= 2e-7 * np.maximum(1e-3, np.abs(a[np.isfinite(a)])) + 1e-9)


def test_reference_tile(hip, oracle):
    from s2p_amd import triangulation as tri
    g, m, r1, r2, bbx = tile_inputs(oracle)
    lla, err = tri.disp_to_lonlatalt(r1, r2, g["H_ref"], g["H_sec"], m["disp"], g["mask_rect"], bbx, g["mask_orig"], A=g["A"])
    assert same(g["lonlatalt_4"], lla[::4, ::4])                      # the reference's own output
    assert err_close(g["err_4"], err[::4, ::4])
    o = oracle.oracle_disp_to_lonlatalt(r1, r2, g["H_ref"], g["H_sec"] @ np.linalg.inv(g["A"]), m["disp"], g["mask_rect"], bbx, g["mask_orig"])
    assert same(o[0], lla) and err_close(o[1], err)
    if oracle.have_ref_tri():
        a = oracle.ref_disp_to_lonlatalt(r1, r2, g["H_ref"], g["H_sec"] @ np.linalg.inv(g["A"]), m["disp"], g["mask_rect"], bbx, g["mask_orig"])
        assert same(a[0], lla) and err_close(a[1], err)


def test_reference_symbol_is_a_drop_in(hip, oracle):
    """`disp_to_lonlatalt`, the symbol s2p/triangulation.py:117-145 calls in lib/disp_to_h.so, exported with the
    reference's argument list: call it exactly the way the reference does."""
    from numpy.ctypeslib import ndpointer
    g, m, r1, r2, bbx = tile_inputs(oracle)
    lib = hip.lib()
    disp = m["disp"]
    h, w = disp.shape
    mo = g["mask_orig"].astype(np.float32)
    hh, ww = mo.shape
    lib.disp_to_lonlatalt.restype = None
    lib.disp_to_lonlatalt.argtypes = (ndpointer(dtype=ctypes.c_double, shape=(h, w, 3)), ndpointer(dtype=ctypes.c_float, shape=(h, w)),
                                      ndpointer(dtype=ctypes.c_float, shape=(h, w)), ndpointer(dtype=ctypes.c_float, shape=(h, w)),
                                      ndpointer(dtype=ctypes.c_float, shape=(h, w)), ctypes.c_int, ctypes.c_int,
                                      ndpointer(dtype=ctypes.c_float, shape=(hh, ww)), ctypes.c_int, ctypes.c_int,
                                      ndpointer(dtype=ctypes.c_double, shape=(9,)), ndpointer(dtype=ctypes.c_double, shape=(9,)),
                                      ctypes.c_void_p, ctypes.c_void_p, ndpointer(dtype=ctypes.c_float, shape=(4,)))
    lonlatalt = np.zeros((h, w, 3), dtype='float64')
    err = np.zeros((h, w), dtype='float32')
    H2 = np.dot(g["H_sec"], np.linalg.inv(g["A"]))
    lib.disp_to_lonlatalt(lonlatalt, err, disp.astype('float32'), np.zeros((h, w), dtype='float32'),
                          g["mask_rect"].astype('float32'), w, h, mo, ww, hh, g["H_ref"].flatten(), H2.flatten(),
                          ctypes.byref(r1), ctypes.byref(r2), np.asarray(bbx, dtype='float32'))
    assert same(g["lonlatalt_4"], lonlatalt[::4, ::4])


def test_random_masks_and_ranges(hip, oracle):
    from s2p_amd import triangulation as tri
    g, m, r1, r2, bbx = tile_inputs(oracle)
    rng = np.random.default_rng(4)
    disp = (m["disp"] + rng.uniform(-3, 3, m["disp"].shape)).astype(np.float32)[40:140, 60:260]
    disp[rng.uniform(size=disp.shape) < 0.1] = np.nan
    msk = (rng.uniform(size=disp.shape) > 0.2).astype(np.uint8) * np.isfinite(disp)
    mo = (rng.uniform(size=(350, 350)) > 0.1).astype(np.uint8)
    T = np.array([[1, 0, -60], [0, 1, -40], [0, 0, 1.0]])                # the window's own rectifying homographies
    H1, H2 = T @ g["H_ref"], T @ g["H_sec"]
    lla, err = tri.disp_to_lonlatalt(r1, r2, H1, H2, np.nan_to_num(disp), msk, bbx, mo)
    o = oracle.oracle_disp_to_lonlatalt(r1, r2, H1, H2, np.nan_to_num(disp), msk, bbx, mo)
    assert same(o[0], lla) and err_close(o[1], err)
    assert 0 < np.isfinite(err).mean() < 1


# ---- the rest of lib/disp_to_h.so: stereo_corresp_to_lonlatalt, count_3d_neighbors, remove_isolated_3d_points ----
def test_filter3d_reference_fixture(hip):
    from s2p_amd import triangulation as tri
    g = load_golden("filter3d")
    r, p, n, q = float(g["params"][0]), int(g["params"][1]), int(g["params"][2]), int(g["params"][3])
    assert np.array_equal(tri.count_3d_neighbors(g["xyz"], r, p), g["count"])          # the reference's own output
    xyz = g["xyz"].copy()
    tri.remove_isolated_3d_points(xyz, r, p, n, q)
    assert np.array_equal(np.isnan(xyz[:, :, 0]), g["removed"])
    assert same(xyz[~g["removed"]], g["xyz"][~g["removed"]]) and np.isnan(xyz[g["removed"]]).all()


@pytest.mark.parametrize("seed,shape,r,p,n,q", [(11, (40, 50), 1.0, 2, 9, 1), (12, (33, 70), 0.8, 3, 6, 2), (13, (301, 257), 2.0, 4, 30, 1),
                                                (14, (20, 20), 1.0, 0, 1, 0), (15, (30, 41), 1.2, 2, 100, 1), (16, (1, 90), 1.0, 2, 3, 1),
                                                (17, (90, 1), 1.0, 2, 3, 1)])
def test_filter3d_against_oracle(hip, oracle, seed, shape, r, p, n, q):
    from helpers import synth_cloud
    from s2p_amd import triangulation as tri
    cloud = synth_cloud(seed, *shape)
    assert np.array_equal(tri.count_3d_neighbors(cloud, r, p), oracle.oracle_count_3d_neighbors(cloud, r, p))
    xyz = cloud.copy()
    tri.remove_isolated_3d_points(xyz, r, p, n, q)
    assert same(xyz, oracle.oracle_remove_isolated_3d_points(cloud, r, p, n, q))
    if oracle.have_ref_tri() and cloud.size < 50000:
        assert same(xyz, oracle.ref_remove_isolated_3d_points(cloud, r, p, n, q))


def test_long_rescue_chain(hip, oracle):
    """A single row of points each 0.9 r from the next, only the first one accepted: the whole row is saved one
    link at a time (hundreds of dependent rescues; the reference's raster sweep does it in one pass left to right)."""
    from s2p_amd import triangulation as tri
    w = 700
    xyz = np.zeros((3, w, 3))
    xyz[:, :, 0] = 0.9 * np.arange(w)[None, :]
    xyz[0, :, 1] = 50.0; xyz[2, :, 1] = -50.0                      # rows 0 and 2 far away: they never help
    xyz[0, :, 2] = 1000.0 * np.arange(w); xyz[2, :, 2] = -1000.0 * np.arange(w) - 7.0
    xyz[:, :3, 1:] = 0.0                                           # a dense 3x3 blob at the left end: accepted
    xyz[1, :, 1] = 0.0
    cnt = tri.count_3d_neighbors(xyz, 1.0, 1)
    assert cnt[1, 1] == 9 and cnt[1, 2] == 7 and cnt[1, 300] == 3
    out = xyz.copy()
    tri.remove_isolated_3d_points(out, 1.0, 1, 9, 1)
    ref = oracle.oracle_remove_isolated_3d_points(xyz, 1.0, 1, 9, 1)
    assert same(out, ref)
    assert np.isfinite(out[1]).all() and np.isnan(out[0, 5:]).all()


def test_corresp_and_dropin_symbols(hip, oracle):
    """stereo_corresp_to_lonlatalt, count_3d_neighbors, remove_isolated_3d_points: the Python mirrors and the
    reference-named symbols bound exactly as s2p/triangulation.py:244-258,292-299,324-328 binds lib/disp_to_h.so."""
    from numpy.ctypeslib import ndpointer
    from s2p_amd import triangulation as tri
    g, t = load_golden("filter3d"), load_golden("tri_tile")
    r1, r2 = oracle.rpc_from_geotiff_tag(t["rpc1"]), oracle.rpc_from_geotiff_tag(t["rpc2"])
    lla, err = tri.stereo_corresp_to_lonlatalt(r1, r2, g["pts1"], g["pts2"])
    assert same(lla, g["corresp_lonlatalt"]) and err_close(g["corresp_err"], err)
    lib = hip.lib()
    n = len(g["pts1"])
    lib.stereo_corresp_to_lonlatalt.restype = None
    lib.stereo_corresp_to_lonlatalt.argtypes = (ndpointer(dtype=ctypes.c_double, shape=(n, 3)), ndpointer(dtype=ctypes.c_float, shape=(n,)),
                                                ndpointer(dtype=ctypes.c_float, shape=(n, 2)), ndpointer(dtype=ctypes.c_float, shape=(n, 2)),
                                                ctypes.c_int, ctypes.POINTER(type(r1)), ctypes.POINTER(type(r2)))
    lla2, err2 = np.zeros((n, 3)), np.zeros(n, np.float32)
    lib.stereo_corresp_to_lonlatalt(lla2, err2, g["pts1"], g["pts2"], n, ctypes.byref(r1), ctypes.byref(r2))
    assert same(lla2, lla) and same(err2, err)
    h, w, _ = g["xyz"].shape
    lib.count_3d_neighbors.restype = None
    lib.count_3d_neighbors.argtypes = (ndpointer(dtype=ctypes.c_int, shape=(h, w)), ndpointer(dtype=ctypes.c_double, shape=(h, w, 3)),
                                       ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int)
    out = np.zeros((h, w), dtype="int32")
    lib.count_3d_neighbors(out, np.ascontiguousarray(g["xyz"]), w, h, 1.0, 2)
    assert np.array_equal(out, g["count"])
    lib.remove_isolated_3d_points.restype = None
    lib.remove_isolated_3d_points.argtypes = (ndpointer(dtype=ctypes.c_double, shape=(h, w, 3)), ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                              ctypes.c_int, ctypes.c_int, ctypes.c_int)
    xyz = np.ascontiguousarray(g["xyz"].copy())
    lib.remove_isolated_3d_points(xyz, w, h, 1.0, 2, 9, 1)
    assert np.array_equal(np.isnan(xyz[:, :, 0]), g["removed"])
    xyz2 = g["xyz"].copy()
    tri.filter_xyz(xyz2, 1.0, 9, 0.5)                               # p = ceil(r / gsd) = 2
    assert same(xyz2, xyz)


def test_xyz_mirrors_with_the_reference_signatures(hip, oracle):
    """disp_to_xyz / stereo_corresp_to_xyz (s2p/triangulation.py:85-164, 220-262): the functions the orchestrator calls, with
    its argument lists; lon / lat / alt when out_crs is None, the UTM zone's easting / northing otherwise (the conversion is
    s2p_amd.geographiclib's, checked against pyproj-made values in tests/test_geographiclib.py)."""
    from s2p_amd import triangulation as tri, geographiclib
    g, m, r1, r2, bbx = tile_inputs(oracle)
    lla, err = tri.disp_to_xyz(r1, r2, g["H_ref"], g["H_sec"], m["disp"], g["mask_rect"], bbx, g["mask_orig"], A=g["A"])
    assert same(g["lonlatalt_4"], lla[::4, ::4]) and err_close(g["err_4"], err[::4, ::4])
    ok = np.isfinite(lla[:, :, 0])
    code = geographiclib.epsg_code_from_utm_zone(geographiclib.compute_utm_zone(np.nanmean(lla[:, :, 0]), np.nanmean(lla[:, :, 1])))
    xyz, err2 = tri.disp_to_xyz(r1, r2, g["H_ref"], g["H_sec"], m["disp"], g["mask_rect"], bbx, g["mask_orig"], A=g["A"], out_crs="epsg:%d" % code)
    e, n = geographiclib.lonlat_to_utm(lla[:, :, 0][ok], lla[:, :, 1][ok], *geographiclib.utm_zone_from_epsg(code))
    assert same(err, err2) and np.array_equal(np.isfinite(xyz[:, :, 0]), ok)
    assert same(xyz[:, :, 0][ok], e) and same(xyz[:, :, 1][ok], n) and same(xyz[:, :, 2], lla[:, :, 2])
    f = load_golden("filter3d")
    p, perr = tri.stereo_corresp_to_xyz(r1, r2, f["pts1"], f["pts2"])
    assert same(p, f["corresp_lonlatalt"]) and perr.shape == (len(p), 1) and err_close(f["corresp_err"], perr[:, 0])


def test_rpc_from_geotiff_tag_matches_the_oracle_parser(hip, oracle):
    """The product's own RPCCoefficientTag reader (s2p_amd.triangulation.rpc_from_geotiff_tag) fills the struct
    byte for byte like the one the fixtures were generated with."""
    from s2p_amd import triangulation as tri
    t = load_golden("tri_tile")
    for k in ("rpc1", "rpc2"):
        a, b = tri.rpc_from_geotiff_tag(t[k]), oracle.rpc_from_geotiff_tag(t[k])
        assert ctypes.sizeof(a) == ctypes.sizeof(b)
        assert np.array_equal(np.frombuffer(bytes(a), np.uint8), np.frombuffer(bytes(b), np.uint8))
    with pytest.raises(ValueError):
        tri.rpc_from_geotiff_tag(t["rpc1"][:50])


TRANSFER_CASES = [
    # seed, (hr, wr) rectified, (h, w) original, H (affine), nan fraction
    (1, (120, 150), (110, 140), [[0.98, 0.05, 3.2], [-0.04, 1.01, -2.1], [0, 0, 1]], 0.05),
    (2, (90, 64), (128, 70), [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0, 0, 1]], 0.0),          # identity, larger output: zeros outside (cval), no NaN
    (3, (200, 333), (150, 300), [[1.1, -0.2, 12.5], [0.15, 0.9, -7.25], [0, 0, 1]], 0.2),
    (4, (64, 64), (64, 64), [[1.0, 0.0, 0.5], [0.0, 1.0, -0.5], [0, 0, 1]], 0.01),         # half-pixel shift: ties of the order-0 rounding
    (5, (50, 80), (40, 60), [[0.5, 0.0, 100.0], [0.0, 0.5, 100.0], [0, 0, 1]], 0.1),       # entirely outside the input
    (6, (1, 1), (3, 3), [[1.0, 0.0, -1.0], [0.0, 1.0, -1.0], [0, 0, 1]], 0.0),
]


@pytest.mark.parametrize("seed,rs,os_,H,nanf", TRANSFER_CASES)
def test_height_transfer_matches_scipy_bit_for_bit(hip, oracle, seed, rs, os_, H, nanf):
    """The resampling half of triangulation.height_map (s2p/triangulation.py:376-389) against scipy itself."""
    rng = np.random.default_rng(seed)
    hm = rng.normal(50, 10, rs)
    hm[rng.uniform(size=rs) < nanf] = np.nan
    if nanf > 0.04:
        hm[rs[0] // 3: rs[0] // 3 + 9, rs[1] // 2: rs[1] // 2 + 17] = np.nan
    H = np.array(H, np.float64)
    want = oracle.oracle_height_transfer(hm, H, os_[1], os_[0])
    got = hip.height_transfer(hm, H, os_[1], os_[0])
    assert got.shape == want.shape == os_ and got.dtype == np.float64
    assert same(want, got)


def test_height_transfer_refuses_a_projective_matrix(hip):
    with pytest.raises(hip.HipError) as e:
        hip.height_transfer(np.zeros((8, 8)), [[1, 0, 0], [0, 1, 0], [1e-6, 0, 1]], 8, 8)
    assert e.value.code == hip.BAD_ARGUMENT


def test_height_map_on_the_reference_tile(hip, oracle):
    """triangulation.height_map end to end (s2p/triangulation.py:346-389) on the reference's tile: the padded
    disp_to_xyz call + the scipy resampling, against the oracle triangulation followed by scipy itself.  The
    reference's H_ref is affine (pushbroom rectification), as scipy requires."""
    from s2p_amd import triangulation as tri
    g, m, r1, r2, bbx = tile_inputs(oracle)
    x, y, w, h = (int(v) for v in g["tile"])
    assert np.array_equal(np.asarray(g["H_ref"])[2], [0, 0, 1])
    got = tri.height_map(x, y, w, h, r1, r2, g["H_ref"], g["H_sec"], m["disp"], g["mask_rect"], g["mask_orig"], A=g["A"])
    p = 1
    lla, _ = oracle.oracle_disp_to_lonlatalt(r1, r2, g["H_ref"], g["H_sec"] @ np.linalg.inv(g["A"]), m["disp"], g["mask_rect"],
                                             (x - p, x + w + 2 * p, y - p, y + h + 2 * p), np.pad(g["mask_orig"], p, constant_values=1))
    T = np.array([[1.0, 0, x], [0, 1.0, y], [0, 0, 1.0]])
    want = oracle.oracle_height_transfer(lla[:, :, 2], np.dot(g["H_ref"], T), w, h)
    assert got.shape == (h, w) and same(want, got)
    assert np.isfinite(got).mean() > 0.5
