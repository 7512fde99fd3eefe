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
    (300, 260, -70, 72, 3, {"recursion": 2}),                   # D = 144: batches run 16 candidates per lane (padded), a tile alone 8 on 32 lanes
    (200, 330, -90, 100, 2, {"recursion": 2, "median": 0}),     # D = 192 (what BASELINE configs[2] has)
    (160, 200, -115, 120, 3, {"recursion": 1}),                 # D = 240
    (300, 260, -70, 72, 2, {"recursion": 2, "nb_dir": 16}),     # D = 144, 52 lattices
    (200, 310, -47, 48, 3, {"recursion": 2}),                   # D = 96: batches run 12 candidates per lane on 8 lanes, a tile alone 8 on 16 (padded)
    (150, 200, -20, 27, 4, {"recursion": 1, "median": 0}),      # D = 48: 12 per lane on 4 lanes
    (70, 90, -3, 4, 5, {"recursion": 1, "nb_dir": 4}),
    (300, 420, -31, 32, 3, {"recursion": 2, "nb_dir": 16}),     # 52 lattices per tile, 16 e-volumes
    (96, 160, -60, 67, 4, {"recursion": 1, "nb_dir": 16, "median": 0}),
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
    (300, 256, -24, 40, 3, {"recursion": 1, "scales": 6, "nb_dir": 16, "median": 0, "remove_small_cc": 25}),
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


def _hetero(hip, tiles, ranges, p):
    n = len(tiles)
    outs = [(np.full(t[0].shape, 7, np.float32), np.full(t[0].shape, 7, np.float32), np.full(t[0].shape, 7, np.uint8)) for t in tiles]
    a = [np.ascontiguousarray(t[0], np.float32) for t in tiles]
    b = [np.ascontiguousarray(t[1], np.float32) for t in tiles]
    hip.census_sgm_host_batch_v(hip.context(), [x.ctypes.data for x in a], [x.ctypes.data for x in b], [t[0].shape[1] for t in tiles],
                                [t[0].shape[0] for t in tiles], [r[0] for r in ranges], [r[1] for r in ranges], p,
                                [o[0].ctypes.data for o in outs], [o[1].ctypes.data for o in outs], [o[2].ctypes.data for o in outs])
    return outs


@pytest.mark.parametrize("shapes,kw", [
    ([(96, 160, -12, 19), (100, 150, -10, 17), (90, 170, -14, 20)], {"recursion": 2}),                       # one depth (32), three sizes
    ([(200, 310, -40, 50), (230, 280, -30, 33), (180, 330, -47, 48), (200, 310, -35, 30)], {"recursion": 1}),  # depths 96 / 64 / 96 / 80 -> 96
    ([(300, 420, -60, 67), (280, 400, -50, 45)], {"recursion": 2, "median": 0}),                               # depth 128: G = 16, 4-wave batch bands
    ([(130, 128, -120, 135), (128, 140, -100, 110)], {"recursion": 2}),                                        # depth 256
    ([(200, 310, -90, 100), (230, 280, -70, 72), (180, 330, -80, 60)], {"recursion": 2}),                      # depths 192 / 144 / 144 -> 192: the padded 16-per-lane layout
    ([(150, 200, -100, 110), (140, 210, -112, 115)], {"recursion": 1, "median": 0}),                           # depths 224 / 240 -> 240
    ([(70, 90, -3, 4), (64, 100, -2, 6), (80, 80, -4, 3), (70, 90, -3, 4), (75, 85, -1, 7)], {"recursion": 1, "nb_dir": 4}),
    ([(200, 310, -40, 50), (230, 280, -30, 33), (180, 330, -47, 48)], {"recursion": 2, "nb_dir": 16}),       # 16 directions: 52 lattices per tile
    ([(300, 256, -24, 40), (280, 260, -20, 37)], {"recursion": 1, "scales": 6, "nb_dir": 16, "median": 0}),
    ([(96, 160, -12, 19), (100, 150, -10, 17)], {"recursion": 0}),                                             # 8-path: tile by tile
    ([(300, 256, -24, 40), (280, 260, -20, 37), (290, 270, -30, 33)], {"recursion": 1, "scales": 6, "median": 0, "remove_small_cc": 25}),   # multi-scale (2 levels each)
    ([(520, 540, -60, 70), (530, 512, -50, 66)], {"recursion": 2, "scales": 6, "lr_check": 2}),                # 3 levels each
    ([(300, 256, -24, 40), (250, 260, -20, 37)], {"recursion": 1, "scales": 6}),                               # 2 levels and 1 level: tile by tile
    ([(96, 160, -12, 19), (100, 150, -10, 17)], {"recursion": 1, "P1": 30, "P2": 120}),                        # P2 > 115: tile by tile
    ([(96, 160, -12, 19), (100, 150, -10, 17), (90, 170, -14, 20)], {"recursion": 2, "mindiff": 6}),
    ([(1, 80, -3, 4), (90, 1, -2, 5), (120, 140, -4, 3), (2, 2, -3, 3)], {"recursion": 2}),                   # a row, a column, a tile, a speck
    ([(800, 780, -120, 135), (790, 800, -128, 120)], {"recursion": 2}),                                        # depth 256 from 768 px: 16 disparities per lane (K = 8)
    ([(300, 260, -250, 250), (280, 300, -230, 260)], {"recursion": 1, "P1": 4, "P2": 20}),                     # depth 512
])
def test_tiles_of_different_shapes_in_one_call_equal_single_calls(shapes, kw):
    """s2p_hip_census_sgm_host_batch_v (round 4): tiles of different sizes and disparity ranges under ONE aggregation launch -- volumes of
    the widest range's depth, per-tile geometry in the kernel -- give the bytes of single calls, whose volumes have each tile's own depth."""
    from s2p_amd import _lib as hip
    tiles = [synth_pair(1200 + t, h, w, lambda x, y, t=t, lo=lo, hi=hi: 0.5 * (lo + hi) + 0.3 * (hi - lo) * np.sin(x / (19. + 3 * t)) * np.cos(y / (23. - 2 * t)),
                        nan=(t == 1)) for t, (h, w, lo, hi) in enumerate(shapes)]
    p = hip.default_census_params(**kw)
    got = _hetero(hip, tiles, [(s[2], s[3]) for s in shapes], p)
    for t, ((im1, im2), (h, w, lo, hi)) in enumerate(zip(tiles, shapes)):
        r = hip.census_sgm(im1, im2, lo, hi, params=p)
        assert same(got[t][0], r["disp"]), "tile %d: disparity" % t
        assert same(got[t][1], r["conf"]) and np.array_equal(got[t][2], r["mask"]), "tile %d" % t


def test_different_shapes_fuzz():
    from s2p_amd import _lib as hip
    rng = np.random.default_rng(77)
    for it in range(12):
        n = int(rng.integers(2, 7))
        h0, w0 = int(rng.integers(60, 360)), int(rng.integers(60, 360))
        span = int(rng.integers(8, 200))
        shapes = []
        for t in range(n):
            lo = -int(rng.integers(0, span))
            shapes.append((h0 + int(rng.integers(-20, 21)), w0 + int(rng.integers(-20, 21)), lo, lo + int(span * rng.uniform(0.76, 1.0))))
        kw = {"recursion": int(rng.integers(1, 3)), "median": int(rng.integers(0, 2)), "nb_dir": int(rng.choice([4, 8])), "census_win": int(rng.choice([3, 5])),
              "lr_check": int(rng.integers(0, 2)), "remove_small_cc": int(rng.choice([0, 25]))}
        tiles = [synth_pair(5000 + 10 * it + t, h, w, lambda x, y, t=t, lo=lo, hi=hi: 0.5 * (lo + hi) + 0.25 * (hi - lo) * np.sin(x / (17. + 2 * t)) * np.cos(y / 21.),
                            nan=(t % 3 == 1)) for t, (h, w, lo, hi) in enumerate(shapes)]
        p = hip.default_census_params(**kw)
        got = _hetero(hip, tiles, [(s[2], s[3]) for s in shapes], p)
        for t, ((im1, im2), (h, w, lo, hi)) in enumerate(zip(tiles, shapes)):
            r = hip.census_sgm(im1, im2, lo, hi, params=p)
            assert same(got[t][0], r["disp"]) and same(got[t][1], r["conf"]) and np.array_equal(got[t][2], r["mask"]), (it, t, shapes, kw)


def test_different_shapes_multi_scale_fuzz():
    from s2p_amd import _lib as hip
    rng = np.random.default_rng(91)
    for it in range(6):
        n = int(rng.integers(2, 5))
        h0, w0 = int(rng.integers(270, 400)), int(rng.integers(270, 400))           # two levels for every tile
        span = int(rng.integers(20, 120))
        shapes = []
        for t in range(n):
            lo = -int(rng.integers(0, span))
            shapes.append((h0 + int(rng.integers(-12, 13)), w0 + int(rng.integers(-12, 13)), lo, lo + int(span * rng.uniform(0.8, 1.0))))
        m = float(rng.choice([1.0, 1.3]))
        kw = {"recursion": int(rng.integers(1, 3)), "scales": 6, "median": int(rng.integers(0, 2)), "remove_small_cc": int(rng.choice([0, 25])),
              "lr_check": int(rng.integers(0, 3)), "P1": int(np.floor(8 * m + 0.5)), "P2": int(np.floor(32 * m + 0.5))}
        tiles = [synth_pair(7000 + 10 * it + t, h, w, lambda x, y, t=t, lo=lo, hi=hi: 0.5 * (lo + hi) + 0.2 * (hi - lo) * np.sin(x / (27. + 2 * t)) * np.cos(y / 31.),
                            nan=(t % 3 == 1)) for t, (h, w, lo, hi) in enumerate(shapes)]
        p = hip.default_census_params(**kw)
        got = _hetero(hip, tiles, [(s[2], s[3]) for s in shapes], p)
        for t, ((im1, im2), (h, w, lo, hi)) in enumerate(zip(tiles, shapes)):
            r = hip.census_sgm(im1, im2, lo, hi, params=p)
            assert same(got[t][0], r["disp"]) and same(got[t][1], r["conf"]) and np.array_equal(got[t][2], r["mask"]), (it, t, shapes, kw)


def _host_batch(hip, tiles, dmin, dmax, p, layout):
    """The host batch entries on planes laid out as the caller says: "arena" = every tile's five planes back to back at ONE 256-byte-rounded
    stride in one block (what a broker arena looks like: two transfers per tile), "paged" = at a page-rounded stride (gaps too wide to be
    alignment padding: five transfers), "odd" = the conf plane moved away (no common stride: five)."""
    n = len(tiles)
    bufs, ad = [], {k: [] for k in ("im1", "im2", "disp", "conf", "mask")}
    for im1, im2 in tiles:
        h, w = im1.shape
        npx = h * w
        a4 = (npx * 4 + 255) // 256 * 256 if layout != "paged" else (npx * 4 + 4095) // 4096 * 4096
        buf = hip.pinned_empty((6 * a4 + 4096,), np.uint8)
        buf[:] = 0xEE                                           # (canary: a byte outside the planes and their < 256-byte gaps must survive the call)
        base = buf.ctypes.data
        offs = {"im1": 0, "im2": a4, "disp": 2 * a4, "conf": 3 * a4 if layout != "odd" else 5 * a4 + 256, "mask": 4 * a4}
        buf[offs["im1"]:offs["im1"] + npx * 4].view(np.float32)[:] = np.ascontiguousarray(im1, np.float32).ravel()
        buf[offs["im2"]:offs["im2"] + npx * 4].view(np.float32)[:] = np.ascontiguousarray(im2, np.float32).ravel()
        for k in ad:
            ad[k].append(base + offs[k])
        bufs.append((buf, offs, h, w))
    ctx = ctypes.c_void_p()
    hip.check(hip.lib().s2p_hip_ctx_create(0, None, ctypes.byref(ctx)))
    try:
        shapes = {(b[2], b[3]) for b in bufs}
        if len(shapes) == 1:
            h, w = bufs[0][2], bufs[0][3]
            hip.census_sgm_host_batch(ctx, ad["im1"], ad["im2"], w, h, dmin, dmax, p, ad["disp"], ad["conf"], ad["mask"], 60.0)
        else:
            hip.census_sgm_host_batch_v(ctx, ad["im1"], ad["im2"], [b[3] for b in bufs], [b[2] for b in bufs], [dmin] * n, [dmax] * n, p,
                                        ad["disp"], ad["conf"], ad["mask"], 60.0)
    finally:
        hip.lib().s2p_hip_ctx_destroy(ctx)
    out = []
    for buf, offs, h, w in bufs:
        npx = h * w
        end = max(offs["mask"] + npx, offs["conf"] + npx * 4)
        assert (buf[end + 256:] == 0xEE).all(), "the call wrote beyond the caller's planes"
        if layout == "arena":                                   # what travels in the alignment gaps behind disp and conf is zeros, not the workspace's past
            a4_ = offs["conf"] - offs["disp"]
            assert not buf[offs["disp"] + npx * 4:offs["conf"]].any() and not buf[offs["conf"] + npx * 4:offs["conf"] + a4_].any()
        else:                                                   # any other layout: nothing but the planes is touched
            assert (buf[offs["disp"] + npx * 4:min(offs["conf"], offs["mask"])] == 0xEE).all()
        out.append((buf[offs["disp"]:offs["disp"] + npx * 4].view(np.float32).reshape(h, w).copy(),
                    buf[offs["conf"]:offs["conf"] + npx * 4].view(np.float32).reshape(h, w).copy(),
                    buf[offs["mask"]:offs["mask"] + npx].reshape(h, w).copy()))
    return out


@pytest.mark.parametrize("ragged", [False, True])
def test_host_batch_with_one_stride_per_tile_moves_two_transfers_and_the_same_bytes(ragged):
    """Round 6 (csrc/api.hip: common_plane_stride): a caller that keeps a tile's five planes at one stride in one block -- the broker's
    arenas -- gets the inputs up in ONE copy and the three outputs down in ONE; any other layout keeps the five transfers.  Same results,
    byte for byte, as single calls, for equal tiles (s2p_hip_census_sgm_host_batch) and tiles of different sizes (..._batch_v)."""
    from s2p_amd import _lib as hip
    p = hip.default_census_params(recursion=2)
    tiles = _tiles(3, 97, 161, 9.0, 900)                         # (npx * 4 is no multiple of a page: the arena stride differs from the packed one)
    if ragged:
        tiles[1] = synth_pair(77, 120, 140, lambda x, y: 8.0 * np.sin(x / 21.) * np.cos(y / 17.))
    want = [hip.census_sgm(a, b, -12, 19, params=p) for a, b in tiles]
    for layout in ("arena", "paged", "odd"):
        got = _host_batch(hip, tiles, -12, 19, p, layout)
        for t in range(len(tiles)):
            assert same(got[t][0], want[t]["disp"]) and same(got[t][1], want[t]["conf"]) and np.array_equal(got[t][2], want[t]["mask"]), (layout, t)


def test_cu_masked_streams_change_no_byte():
    """S2P_HIP_CU_BAND / S2P_HIP_CU_ROWS put the band-pipelined MGM launches and the row kernels of a context on streams
    confined to disjoint CU masks (hipExtStreamCreateWithCUMask; measured: profiles/r06/cumask_sweep.txt, off by default).  A batched and
    a single call in a process of their own under 192 : 64 give the bytes of the unmasked library."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, hashlib; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import numpy as np\n"
            "from helpers import synth_pair\n"
            "from s2p_amd import _lib as hip\n"
            "p = hip.default_census_params(recursion=2)\n"
            "hsh = hashlib.blake2b(digest_size=16)\n"
            "for seed, (h, w) in enumerate(((300, 420), (96, 160))):\n"
            "    a, b = synth_pair(40 + seed, h, w, lambda x, y: 11.0 * np.sin(x / 23.) * np.cos(y / 19.))\n"
            "    r = hip.census_sgm(a, b, -31, 32, params=p)\n"
            "    for k in ('disp', 'conf', 'mask'): hsh.update(np.ascontiguousarray(r[k]).tobytes())\n"
            "print('DIGEST', hsh.hexdigest())\n") % (root, os.path.join(root, "tests"))
    outs = []
    for extra in ({}, {"S2P_HIP_CU_BAND": "192", "S2P_HIP_CU_ROWS": "64"}, {"S2P_HIP_CU_ROWS": "128"}):
        env = dict(os.environ, **extra)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stderr[-1500:]
        outs.append([l for l in r.stdout.splitlines() if l.startswith("DIGEST")][0])
    assert outs[0] == outs[1] == outs[2], outs
