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
