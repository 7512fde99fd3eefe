"""s2p_hip_census_sgm_dev_batch: n equal-shape tiles under one aggregation launch (one ready queue, staggered tiles) give the
results of n single calls, bit for bit -- for both MGM modes, several shapes (multi-band lattices), every stagger setting, and
the parameter sets that fall back to tile-by-tile processing."""
import ctypes
import os

import numpy as np
import pytest

from helpers import DevMem, same, synth_pair

pytestmark = pytest.mark.gpu


def _batch(hip, tiles, dmin, dmax, params):
    lib, dm = hip.lib(), DevMem()
    n = len(tiles)
    h, w = tiles[0][0].shape
    try:
        a = [dm.upload(np.ascontiguousarray(t[0], np.float32)) for t in tiles]
        b = [dm.upload(np.ascontiguousarray(t[1], np.float32)) for t in tiles]
        d = [dm.upload(np.zeros((h, w), np.float32)) for _ in tiles]
        c = [dm.upload(np.zeros((h, w), np.float32)) for _ in tiles]
        m = [dm.upload(np.zeros((h, w), np.uint8)) for _ in tiles]
        arr = lambda ps: (ctypes.c_void_p * n)(*[p.value for p in ps])
        ctx = ctypes.c_void_p()
        hip.check(lib.s2p_hip_ctx_create(0, None, ctypes.byref(ctx)))
        A, B, Dd, Cc, Mm = arr(a), arr(b), arr(d), arr(c), arr(m)
        hip.check(lib.s2p_hip_census_sgm_dev_batch(ctx, n, A, B, w, h, dmin, dmax, ctypes.byref(params), Dd, Cc, Mm))
        hip.check(lib.s2p_hip_ctx_sync(ctx))
        out = [(dm.download(d[t], (h, w), np.float32), dm.download(c[t], (h, w), np.float32), dm.download(m[t], (h, w), np.uint8)) for t in range(n)]
        lib.s2p_hip_ctx_destroy(ctx)
        return out
    finally:
        dm.free()


def _tiles(n, h, w, amp, seed0):
    return [synth_pair(seed0 + t, h, w, lambda x, y, t=t: amp * np.sin(x / (19. + 3 * t)) * np.cos(y / (23. - 2 * t)), nan=(t == 1)) for t in range(n)]


@pytest.mark.parametrize("h,w,dmin,dmax,n,kw", [
    (96, 160, -12, 19, 3, {"recursion": 2}),
    (96, 160, -12, 19, 4, {"recursion": 1}),
    (300, 420, -31, 32, 3, {"recursion": 2}),                   # 10 bands per axis lattice
    (300, 260, -120, 135, 2, {"recursion": 2, "median": 0}),    # D = 256
    (70, 90, -3, 4, 5, {"recursion": 1, "nb_dir": 4}),
    (128, 128, -8, 8, 3, {"recursion": 0}),                     # 8-path: tile by tile
    (300, 256, -24, 40, 2, {"recursion": 2, "scales": 6}),      # multi-scale: tile by tile
])
def test_batch_equals_single_calls(h, w, dmin, dmax, n, kw):
    from s2p_amd import _lib as hip
    tiles = _tiles(n, h, w, 0.3 * (dmax - dmin), 500)
    p = hip.default_census_params(**{"recursion": 0, **kw})
    got = _batch(hip, tiles, dmin, dmax, p)
    for t, (im1, im2) in enumerate(tiles):
        r = hip.census_sgm(im1, im2, dmin, dmax, params=p)
        assert same(got[t][0], r["disp"]) and same(got[t][1], r["conf"]) and np.array_equal(got[t][2], r["mask"]), "tile %d" % t


@pytest.mark.parametrize("stagger", ["-1", "0", "64", "255"])
def test_batch_under_every_stagger(stagger):
    from s2p_amd import _lib as hip
    tiles = _tiles(4, 200, 310, 9.0, 700)
    p = hip.default_census_params(recursion=2)
    os.environ["S2P_MGM_STAGGER"] = stagger
    try:
        got = _batch(hip, tiles, -16, 15, p)
    finally:
        del os.environ["S2P_MGM_STAGGER"]
    for t, (im1, im2) in enumerate(tiles):
        r = hip.census_sgm(im1, im2, -16, 15, params=p)
        assert same(got[t][0], r["disp"]) and np.array_equal(got[t][2], r["mask"]), "tile %d, stagger %s" % (t, stagger)


def test_batch_full_size_against_the_oracle(oracle):
    from s2p_amd import _lib as hip
    amp = 40.0
    tiles = [synth_pair(1000 + t, 1024, 1024, lambda x, y: amp * np.sin(2 * np.pi * x / 512.) * np.cos(2 * np.pi * y / 512.)) for t in range(3)]
    p = hip.default_census_params(recursion=2)
    got = _batch(hip, tiles, -64, 63, p)
    o = oracle.oracle_census_sgm(tiles[2][0], tiles[2][1], -64, 63, params=oracle.census_params(recursion=2))
    assert same(got[2][0], o["disp"]) and np.array_equal(got[2][2], o["mask"])
    for t in (0, 1):
        r = hip.census_sgm(tiles[t][0], tiles[t][1], -64, 63, params=p)
        assert same(got[t][0], r["disp"])
