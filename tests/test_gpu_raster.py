"""GPU parity of the DSM rasterisation (s2p_hip_plyflatten_host / s2p_amd.rasterization) through the C ABI: bit-exact
against oracle/rasterize_oracle.c for unweighted means (any radius) and on the reference's golden window; within a
float32 ulp-scale tolerance for Gaussian weights (exp comes from two different math libraries)."""
import ctypes

import numpy as np
import pytest

from helpers import same
from test_oracle_raster import golden_cloud, write_ply

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from s2p_amd import _lib
    assert _lib.device_count() > 0, "no MI355X visible: the HIP path has no fallback"
    return _lib


def test_reference_golden_window(hip, oracle):
    g, cloud, (xoff, yoff, xsize, ysize), (r0, c0, hh, ww) = golden_cloud()
    out = hip.plyflatten(cloud, xoff, yoff, float(g["resolution"]), xsize, ysize)
    assert same(out[r0:r0 + hh, c0:c0 + ww, 0], g["expected"])        # the stored dsm_40cm.tiff, bit for bit
    assert same(out, oracle.oracle_plyflatten(cloud, xoff, yoff, float(g["resolution"]), xsize, ysize))


CASES = [
    # seed, n, nb, xsize, ysize, res, radius
    (1, 5000, 1, 64, 48, 0.5, 0),
    (2, 20000, 4, 100, 130, 0.4, 0),
    (3, 3000, 5, 33, 77, 1.0, 1),
    (4, 8000, 2, 50, 50, 0.25, 3),
    (5, 1, 1, 1, 1, 2.0, 0),
    (6, 0, 3, 7, 5, 1.0, 2),                       # no points at all: a NaN raster
    (7, 200000, 1, 40, 40, 1.0, 0),                # ~125 points per cell: long lists, the order matters most
    (8, 50000, 16, 1500, 1100, 0.1, 0),            # sparse: most cells empty, 16 bands (the maximum)
    (9, 300000, 2, 3, 2, 50.0, 0),                 # 50 000 points per cell: the heap-sort path of crowded cells
    (10, 60000, 1, 6, 6, 10.0, 2),                 # crowded cells with a radius: ~20 000 contributions each
]


@pytest.mark.parametrize("seed,n,nb,xsize,ysize,res,radius", CASES)
def test_unweighted_is_bit_exact(hip, oracle, seed, n, nb, xsize, ysize, res, radius):
    rng = np.random.default_rng(seed)
    xoff, yoff = 360000.0 + seed, 7650000.0 - seed
    x = rng.uniform(xoff - 2 * res, xoff + (xsize + 2) * res, n)           # some points fall outside the raster
    y = rng.uniform(yoff - (ysize + 2) * res, yoff + 2 * res, n)
    vals = rng.normal(2300, 30, (n, nb))
    cloud = np.column_stack([x, y, vals]) if n else np.zeros((0, 2 + nb))
    if n > 10:
        cloud[3, 0] = np.nan; cloud[7, 1] = np.inf; cloud[9, 2] = np.nan   # a NaN value poisons its cell, as on the CPU
    o = oracle.oracle_plyflatten(cloud, xoff, yoff, res, xsize, ysize, radius=radius)
    r = hip.plyflatten(cloud, xoff, yoff, res, xsize, ysize, radius=radius)
    assert same(o, r)
    assert same(r, hip.plyflatten(cloud, xoff, yoff, res, xsize, ysize, radius=radius))   # and run to run


def test_gaussian_weights_within_tolerance(hip, oracle):
    rng = np.random.default_rng(11)
    n, xsize, ysize, res = 30000, 90, 70, 0.4
    cloud = np.column_stack([rng.uniform(0, xsize * res, n), rng.uniform(-ysize * res, 0, n), rng.normal(100, 5, (n, 2))])
    for radius, sigma in ((1, 0.3), (2, 0.5), (3, 5.0)):
        o = oracle.oracle_plyflatten(cloud, 0.0, 0.0, res, xsize, ysize, radius=radius, sigma=sigma)
        r = hip.plyflatten(cloud, 0.0, 0.0, res, xsize, ysize, radius=radius, sigma=sigma)
        assert same(np.isnan(o), np.isnan(r))
        # tolerance: the weights go through exp() of two math libraries (glibc / ROCm device libs), which may differ in
        # the last bit of the double before it is rounded to float32: all but a handful of cells are bit-identical
        assert np.allclose(o, r, rtol=2e-6, atol=0, equal_nan=True)
        assert np.mean(o == r) + np.mean(np.isnan(o)) > 0.999


def test_files_to_dsm_and_the_drop_in_symbol(hip, oracle, tmp_path):
    """plyflatten_from_plyfiles_list on PLY files as s2p's plys_to_dsm calls it (s2p/__init__.py:462-466), and the
    `rasterize_cloud` symbol with plyflatten's own ctypes signature."""
    from s2p_amd import rasterization as R
    g, cloud, (xoff, yoff, xsize, ysize), (r0, c0, hh, ww) = golden_cloud()
    half = len(cloud) // 2
    p1, p2 = str(tmp_path / "a.ply"), str(tmp_path / "b.ply")
    write_ply(p1, g["xyz"][:half], g["rgb"][:half], ["created by S2P", str(g["comments"])])
    write_ply(p2, g["xyz"][half:], g["rgb"][half:], ["created by S2P", str(g["comments"])])
    raster, profile = R.plyflatten_from_plyfiles_list([p1, p2], float(g["resolution"]), roi=(xoff, yoff, xsize, ysize))
    assert raster.shape == (ysize, xsize, 4) and profile["crs"] == "epsg:32740"
    assert same(raster[r0:r0 + hh, c0:c0 + ww, 0], g["expected"])
    assert profile["transform"] == (0.4, 0.0, xoff, 0.0, -0.4, yoff)
    auto, prof2 = R.plyflatten_from_plyfiles_list([p1, p2], float(g["resolution"]), radius=1, sigma=None)
    ax, ay = prof2["transform"][2], prof2["transform"][5]
    assert same(auto, oracle.oracle_plyflatten(cloud, ax, ay, 0.4, auto.shape[1], auto.shape[0], radius=1))
    # the C symbol of plyflatten's shared library
    L = hip.lib()
    fn = L.rasterize_cloud
    fn.restype = None
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                   ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float]
    out = np.empty((ysize, xsize, 4), np.float32)
    c = np.ascontiguousarray(cloud)
    fn(c.ctypes.data, out.ctypes.data, len(c), 4, xoff, yoff, 0.4, xsize, ysize, 0, float("inf"))
    assert same(out, raster)


def test_bad_arguments(hip):
    c = np.zeros((4, 3))
    with pytest.raises(hip.HipError) as e:
        hip.plyflatten(c, 0, 0, 0.0, 4, 4)
    assert e.value.code == hip.BAD_ARGUMENT
    with pytest.raises(hip.HipError) as e:
        hip.plyflatten(np.zeros((4, 20)), 0, 0, 1.0, 4, 4)
    assert e.value.code == hip.BAD_ARGUMENT
    with pytest.raises(ValueError):
        hip.plyflatten(np.zeros((4, 2)), 0, 0, 1.0, 4, 4)
