"""The two tail entries added for the tri-stereo end-to-end path: common.cargarse_basura and the localisation of
triangulation.height_map_to_xyz, HIP vs the oracle (bit-exact: same float32 / float64 operations in the same order)."""
import numpy as np
import pytest

from helpers import load_golden, same

pytestmark = pytest.mark.gpu


def _height_map(seed, h, w):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    hm = (100 + 8 * np.sin(xx / 17.0) * np.cos(yy / 13.0) + rng.normal(0, 0.3, (h, w))).astype(np.float32)
    for _ in range(max(2, h * w // 3000)):                       # buildings: sharp steps of 6 .. 40 m
        y0, x0 = rng.integers(0, h - 4), rng.integers(0, w - 4)
        hm[y0:y0 + rng.integers(3, 40), x0:x0 + rng.integers(3, 40)] += rng.uniform(6, 40)
    hm[rng.uniform(size=hm.shape) < 0.04] = np.nan
    hm[rng.uniform(size=hm.shape) < 0.01] += 30
    return hm


@pytest.mark.parametrize("seed,h,w", [(0, 350, 350), (1, 97, 203), (2, 3, 3), (3, 64, 1), (4, 700, 513), (5, 5, 1200)])
def test_cargarse_basura_matches_the_oracle(oracle, seed, h, w):
    from s2p_amd import _lib
    hm = _height_map(seed, max(h, 5), max(w, 5))[:h, :w].copy()
    if min(h, w) < 3:
        with pytest.raises(_lib.HipError):
            _lib.cargarse_basura(hm)
        return
    got, want = _lib.cargarse_basura(hm), oracle.oracle_cargarse_basura(hm)
    assert same(got, want)
    assert np.isnan(want).sum() > np.isnan(hm).sum() or h * w < 50


def test_cargarse_basura_all_nan_and_flat(oracle):
    from s2p_amd import _lib
    a = np.full((40, 50), np.nan, np.float32)
    assert np.isnan(_lib.cargarse_basura(a)).all()
    b = np.full((40, 50), 12.5, np.float32)
    assert same(_lib.cargarse_basura(b), b)                      # one component of 2000 pixels, range 0
    c = np.full((10, 10), 12.5, np.float32)
    assert np.isnan(_lib.cargarse_basura(c)).all()               # 100 pixels < 200: removed, as in the reference's rule


def test_file_level_cargarse_basura(tmp_path, oracle):
    from s2p_amd import common, io as rio
    hm = _height_map(9, 120, 160)
    p = str(tmp_path / "height_map.tif")
    rio.write_image(p, hm)
    common.cargarse_basura(p, p)                                  # in place, as heights_fusion calls it
    assert same(rio.read_image(p), oracle.oracle_cargarse_basura(hm))


def test_height_map_localisation_matches_the_oracle(oracle):
    from s2p_amd import _lib, triangulation, geographiclib
    g = load_golden("tri_tile")
    x, y, w, h = (int(v) for v in g["tile"])
    rng = np.random.default_rng(4)
    hm = (2300 + 40 * rng.standard_normal((h // 2, w // 2))).astype(np.float32)
    hm[rng.uniform(size=hm.shape) < 0.1] = np.nan
    got = _lib.height_map_to_lonlatalt(triangulation.rpc_from_geotiff_tag(g["rpc1"]), hm, x, y)
    want = oracle.oracle_height_map_to_lonlatalt(oracle.rpc_from_geotiff_tag(g["rpc1"]), hm, x, y)
    assert same(got, want)
    assert np.isnan(got[np.isnan(hm)]).all() and np.isfinite(got[np.isfinite(hm)]).all()
    # the mirror with the reference's signature: lon / lat / alt, then UTM through geographiclib
    xyz = triangulation.height_map_to_xyz(hm, triangulation.rpc_from_geotiff_tag(g["rpc1"]), x, y, out_crs="epsg:32740")
    e, n = geographiclib.lonlat_to_utm(want[..., 0], want[..., 1], 40, True)
    ok = np.isfinite(hm)
    assert np.array_equal(xyz[..., 0][ok], e[ok]) and np.array_equal(xyz[..., 1][ok], n[ok]) and np.array_equal(xyz[..., 2][ok], hm[ok].astype(np.float64))
    with pytest.raises(NotImplementedError):
        triangulation.height_map_to_xyz(hm, triangulation.rpc_from_geotiff_tag(g["rpc1"]), x, y, out_crs="epsg:2154")
