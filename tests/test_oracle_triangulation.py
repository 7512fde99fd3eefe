"""CPU tests of the triangulation restatement (oracle/triangulation_oracle.c) against the reference's own
disp_to_lonlatalt: the golden subsample in tests/golden/tri_tile.npz and, where
oracle/_ref/libdisp_to_h_ref.so exists, the full tile live.  Bar: bit-exact (float64, same order)."""
import numpy as np
import pytest

from helpers import load_golden, same


def tile_inputs(oracle):
    g = load_golden("tri_tile")
    m = load_golden("mgm_tile")
    r1, r2 = oracle.rpc_from_geotiff_tag(g["rpc1"]), oracle.rpc_from_geotiff_tag(g["rpc2"])
    x, y, w, h = (int(v) for v in g["tile"])
    H2 = g["H_sec"] @ np.linalg.inv(g["A"])
    return g, (r1, r2, g["H_ref"], H2, m["disp"], g["mask_rect"], (x, x + w, y, y + h), g["mask_orig"])


def test_restatement_matches_reference_golden(oracle):
    g, args = tile_inputs(oracle)
    lla, err = oracle.oracle_disp_to_lonlatalt(*args)
    assert same(g["lonlatalt_4"], lla[::4, ::4]) and same(g["err_4"], err[::4, ::4])
    v = np.isfinite(err)
    assert 0.5 < v.mean() < 0.6 and np.nanmax(err) < 0.2                 # 56.6 % of the tile, sub-pixel residuals
    assert 2200 < np.nanmin(lla[..., 2]) and np.nanmax(lla[..., 2]) < 2400


@pytest.mark.ref
def test_restatement_matches_live_reference(oracle):
    if not oracle.have_ref_tri():
        pytest.skip("oracle/_ref/libdisp_to_h_ref.so not built (needs /root/reference)")
    g, args = tile_inputs(oracle)
    a = oracle.ref_disp_to_lonlatalt(*args)
    b = oracle.oracle_disp_to_lonlatalt(*args)
    assert same(a[0], b[0]) and same(a[1], b[1])


# ---- the rest of lib/disp_to_h.so: stereo_corresp_to_lonlatalt, count_3d_neighbors, remove_isolated_3d_points ----
def test_filter3d_restatement_matches_reference_golden(oracle):
    from helpers import synth_cloud
    g = load_golden("filter3d")
    r, p, n, q = float(g["params"][0]), int(g["params"][1]), int(g["params"][2]), int(g["params"][3])
    assert same(synth_cloud(31, 72, 96), g["xyz"])                         # the generator the GPU tests use too
    assert np.array_equal(oracle.oracle_count_3d_neighbors(g["xyz"], r, p), g["count"])
    out = oracle.oracle_remove_isolated_3d_points(g["xyz"], r, p, n, q)
    assert np.array_equal(np.isnan(out[:, :, 0]), g["removed"])
    keep = ~g["removed"]
    assert same(out[keep], g["xyz"][keep]) and np.isnan(out[g["removed"]]).all()
    # the mercy step matters in this fixture: some points below the count threshold survive
    assert ((g["count"] < n) & keep).sum() > 20


def test_corresp_restatement_matches_reference_golden(oracle):
    g, t = load_golden("filter3d"), load_golden("tri_tile")
    r1, r2 = oracle.rpc_from_geotiff_tag(t["rpc1"]), oracle.rpc_from_geotiff_tag(t["rpc2"])
    lla, err = oracle.oracle_stereo_corresp_to_lonlatalt(r1, r2, g["pts1"], g["pts2"])
    assert same(lla, g["corresp_lonlatalt"]) and same(err, g["corresp_err"])


@pytest.mark.ref
@pytest.mark.parametrize("seed,shape,r,p,n,q", [(1, (40, 50), 1.0, 2, 9, 1), (2, (33, 70), 0.8, 3, 6, 2), (3, (64, 64), 2.0, 4, 30, 1),
                                                (4, (20, 20), 1.0, 0, 1, 0), (5, (30, 41), 1.2, 2, 100, 1)])
def test_filter3d_restatement_matches_live_reference(oracle, seed, shape, r, p, n, q):
    """The flood formulation of the oracle against the reference's raster-order loop, live."""
    if not oracle.have_ref_tri():
        pytest.skip("oracle/_ref/libdisp_to_h_ref.so not built (needs /root/reference)")
    from helpers import synth_cloud
    xyz = synth_cloud(seed, *shape)
    assert np.array_equal(oracle.ref_count_3d_neighbors(xyz, r, p), oracle.oracle_count_3d_neighbors(xyz, r, p))
    assert same(oracle.ref_remove_isolated_3d_points(xyz, r, p, n, q), oracle.oracle_remove_isolated_3d_points(xyz, r, p, n, q))
