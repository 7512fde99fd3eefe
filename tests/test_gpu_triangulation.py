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
