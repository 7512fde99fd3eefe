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


def _ms_tiles(n, h, w, dmin, dmax, seed0):
    """Tiles whose disparity fields sit in DIFFERENT parts of the configured range (and one with NaN areas), so that the levels'
    unions differ from tile to tile and the batch's hull is wider than every tile's own range."""
    span = dmax - dmin
    tiles = []
    for t in range(n):
        mid = dmin + span * (0.25 + 0.5 * t / max(1, n - 1))
        amp = span * (0.04 + 0.03 * t)
        tiles.append(synth_pair(seed0 + t, h, w, lambda x, y, mid=mid, amp=amp, t=t: mid + amp * np.sin(x / (29. + 5 * t)) * np.cos(y / (31. - 3 * t)), nan=(t == 1)))
    return tiles


@pytest.mark.parametrize("h,w,dmin,dmax,n,kw", [
    (300, 256, -24, 40, 3, {"recursion": 2, "scales": 6}),                                     # 2 levels
    (300, 256, -24, 40, 3, {"recursion": 1, "scales": 6, "median": 0, "remove_small_cc": 25}), # the 'mgm_multi' call's shape of parameters
    (520, 540, -60, 70, 4, {"recursion": 1, "scales": 6, "median": 0, "remove_small_cc": 25, "P1": 10, "P2": 42, "lr_check": 2}),   # 3 levels, m = 1.3
    (260, 300, -20, 27, 2, {"recursion": 1, "scales": 6, "subpix": 2, "median": 0}),           # half-pixel candidates
    (300, 256, -24, 40, 2, {"recursion": 2, "scales": 6, "nb_dir": 4}),
    (300, 256, -24, 40, 2, {"recursion": 1, "scales": 6, "P1": 30, "P2": 120}),                # P2 > 115: tile by tile (see census_batches)
    (512, 512, -100, 120, 3, {"recursion": 1, "scales": 3, "cost": 1, "median": 0}),           # ZNCC cost, scales capped at 3
])
def test_multi_scale_batch_equals_single_calls(h, w, dmin, dmax, n, kw):
    """mgm_multi tiles in one call (round 4): level by level for all tiles, one volume shape per level (the hull of the tiles' ranges),
    one aggregation launch per level, every tile keeping its own range through the [lo, hi] planes and the WTA window -- the same
    bytes as n single calls, whose volumes cover each tile's own range only."""
    from s2p_amd import _lib as hip
    tiles = _ms_tiles(n, h, w, dmin, dmax, 900)
    p = hip.default_census_params(**kw)
    got = _batch(hip, tiles, dmin, dmax, p)
    for t, (im1, im2) in enumerate(tiles):
        r = hip.census_sgm(im1, im2, dmin, dmax, params=p)
        assert same(got[t][0], r["disp"]), "tile %d: disparity" % t
        assert same(got[t][1], r["conf"]) and np.array_equal(got[t][2], r["mask"]), "tile %d" % t
        assert np.isfinite(r["disp"]).mean() > 0.3


def test_multi_scale_batch_fuzz():
    from s2p_amd import _lib as hip
    rng = np.random.default_rng(2024)
    for it in range(10):
        h, w = int(rng.integers(256, 420)), int(rng.integers(256, 420))
        dmin = -int(rng.integers(5, 90))
        dmax = int(rng.integers(5, 90))
        n = int(rng.integers(2, 5))
        kw = {"recursion": int(rng.integers(1, 3)), "scales": 6, "median": int(rng.integers(0, 2)), "remove_small_cc": int(rng.choice([0, 25])),
              "lr_check": int(rng.integers(0, 3)), "nb_dir": int(rng.choice([4, 8])), "census_win": int(rng.choice([3, 5]))}
        m = float(rng.choice([1.0, 1.3, 2.0]))
        kw["P1"], kw["P2"] = int(np.floor(8 * m + 0.5)), int(np.floor(32 * m + 0.5))
        tiles = _ms_tiles(n, h, w, dmin, dmax, 3000 + 10 * it)
        p = hip.default_census_params(**kw)
        got = _batch(hip, tiles, dmin, dmax, p)
        for t, (im1, im2) in enumerate(tiles):
            r = hip.census_sgm(im1, im2, dmin, dmax, params=p)
            assert same(got[t][0], r["disp"]) and same(got[t][1], r["conf"]) and np.array_equal(got[t][2], r["mask"]), (it, t, h, w, dmin, dmax, kw)
