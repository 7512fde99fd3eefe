"""GPU tests of the resampler and of the whole path on the reference's own tile
(tests/golden/warp_tile.npz, mgm_tile.npz), plus the file-level shims with the reference's
exception contracts (tests/block_matching_test.py, tests/common_test.py)."""
import os
import subprocess

import numpy as np
import pytest

from helpers import load_golden, same, synth_pair, DevMem

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from s2p_amd import _lib
    assert _lib.device_count() > 0, "no MI355X visible: the HIP path has no fallback"
    return _lib


def test_warp_matches_oracle_and_fixture(hip, oracle):
    g = load_golden("warp_tile")
    w, h = (int(v) for v in g["size"])
    out = hip.warp(g["src"], g["H"], w, h)
    ref = oracle.oracle_warp(g["src"], g["H"], w, h)
    assert same(out, ref)                                  # same float32 operation order as the oracle: bit-exact
    e = np.abs(out - g["expected"])[12:-12, 12:-12]
    assert e.mean() <= 0.02 and e.max() <= 0.15          # vs the reference binary's stored output


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.float32])
def test_warp_dtypes_projective_nan(hip, oracle, dtype):
    rng = np.random.default_rng(5)
    src = rng.uniform(0, 250, (97, 143))
    src = src.astype(dtype)
    if dtype == np.float32:
        src[40:43, 60:64] = np.nan
    H = np.array([[1.02, 0.05, -7.3], [-0.04, 0.97, 5.1], [2e-5, -1e-5, 1.0]])
    out = hip.warp(src, H, 160, 120)
    ref = oracle.oracle_warp(src, H, 160, 120)
    assert np.isnan(out).any() and np.isfinite(out).any()    # outside-domain and NaN-tap pixels
    assert same(out, ref)


@pytest.mark.parametrize("name", ["zoom-out 2", "zoom-out 1.5 + rotation + perspective", "zoom-in 2"])
def test_warp_off_the_fixtures_envelope_is_the_oracles(hip, oracle, name):
    """The zooms no stored artefact of the reference covers: the kernel equals the oracle bit for bit there too,
    and the oracle is checked against analytic answers on the CPU (tests/test_oracle_tile.py)."""
    sw, sh = 400, 360
    yy, xx = np.mgrid[0:sh, 0:sw].astype(np.float64)
    src = (500 + 180 * np.sin(2 * np.pi * xx / 37.0 + 0.3) * np.cos(2 * np.pi * yy / 53.0) + 90 * np.sin(2 * np.pi * (xx + 2 * yy) / 90.0)).astype(np.float32)
    c, s_ = np.cos(0.5), np.sin(0.5)
    H, w, h = {"zoom-out 2": (np.array([[0.5, 0, 3.25], [0, 0.5, -1.5], [0, 0, 1.0]]), 180, 160),
               "zoom-out 1.5 + rotation + perspective": (np.array([[c / 1.5, -s_ / 1.5, 90.0], [s_ / 1.5, c / 1.5, -40.0], [2e-5, -1e-5, 1.0]]), 220, 200),
               "zoom-in 2": (np.array([[2.0, 0, -300.5], [0, 2.0, -250.25], [0, 0, 1.0]]), 300, 260)}[name]
    assert same(hip.warp(src, H, w, h), oracle.oracle_warp(src, H, w, h))


def test_identity_warp_returns_the_image(hip):
    rng = np.random.default_rng(6)
    src = rng.uniform(100, 700, (50, 70)).astype(np.float32)
    out = hip.warp(src, np.eye(3), 70, 50)
    assert np.abs(out - src).max() < 1e-2                   # interpolating spline: exact at the knots up to fp32


def test_census_on_reference_tile_statistics(hip, oracle):
    g = load_golden("mgm_tile")
    w, h = (int(v) for v in g["size"])
    sec = hip.warp(g["src"], g["H"], w, h)
    d_ref = g["disp"]
    dmin, dmax = int(np.floor(np.nanmin(d_ref))) - 4, int(np.ceil(np.nanmax(d_ref))) + 4
    r = hip.census_sgm(g["ref"], sec, dmin, dmax, params=hip.default_census_params(recursion=0))
    d = r["disp"]
    both = np.isfinite(d) & np.isfinite(d_ref)
    e = np.abs(d[both] - d_ref[both])
    assert (e <= 0.5).mean() >= 0.985 and (e <= 1.0).mean() >= 0.995
    assert abs(np.isfinite(d).mean() - np.isfinite(d_ref).mean()) <= 0.01
    o = oracle.oracle_census_sgm(g["ref"], sec, dmin, dmax, params=oracle.census_params(recursion=0))
    assert same(o["disp"], d)                                # and bit-exact against the oracle on real data
    # MGM recursion: what the `mgm` binary does (north_star: >= 99 % of the valid pixels within 0.5 px)
    rm = hip.census_sgm(g["ref"], sec, dmin, dmax, params=hip.default_census_params(recursion=1))
    dm = rm["disp"]
    both = np.isfinite(dm) & np.isfinite(d_ref)
    e = np.abs(dm[both] - d_ref[both])
    print("mgm recursion vs stored mgm tile: %.4f within 0.5 px, %.4f within 1 px, median %.3f, valid %.3f vs %.3f" % (
        (e <= 0.5).mean(), (e <= 1.0).mean(), np.median(e), np.isfinite(dm).mean(), np.isfinite(d_ref).mean()))
    assert (e <= 0.5).mean() >= 0.99 and (e <= 1.0).mean() >= 0.997
    assert abs(np.isfinite(dm).mean() - np.isfinite(d_ref).mean()) <= 0.01
    assert same(oracle.oracle_census_sgm(g["ref"], sec, dmin, dmax, params=oracle.census_params(recursion=1))["disp"], dm)


def test_sgbm_on_reference_tile_exact(hip, oracle):
    g = load_golden("mgm_tile")
    w, h = (int(v) for v in g["size"])
    sec = hip.warp(g["src"], g["H"], w, h)
    r = hip.sgbm(g["ref"], sec, -45, 35)
    oracle.set_alias_oob(0)
    o = oracle.oracle_sgbm(g["ref"], sec, -45, 35)
    oracle.set_alias_oob(1)
    assert same(o["disp"], r["disp"])
    if oracle.have_ref():
        a = oracle.ref_sgbm(g["ref"], sec, -45, 35)
        assert same(a["disp"], r["disp"])                    # the real reference binary's arithmetic


def _files(tmp_path):
    from s2p_amd import io as rio
    im1, im2 = synth_pair(9, 96, 160, lambda x, y: 6 + 9 * np.sin(x / 37.) * np.cos(y / 29.))
    p1, p2 = str(tmp_path / "rectified_ref.tif"), str(tmp_path / "rectified_sec.tif")
    rio.write_image(p1, im1)
    rio.write_image(p2, im2)
    return im1, im2, p1, p2


@pytest.mark.parametrize("algo", ["sgbm", "mgm", "mgm_multi"])
def test_compute_disparity_map_files(hip, oracle, tmp_path, algo):
    from s2p_amd import block_matching as bm
    from s2p_amd import io as rio
    im1, im2, p1, p2 = _files(tmp_path)
    disp, mask = str(tmp_path / "rectified_disp.tif"), str(tmp_path / "rectified_mask.png")
    bm.compute_disparity_map(p1, p2, disp, mask, algo, -24.3, 39.6, timeout=600)
    d = rio.read_image(disp)
    m = rio.read_image(mask, np.uint8)
    assert d.dtype == np.float32 and d.shape == im1.shape and set(np.unique(m)) <= {0, 1}
    if algo == "sgbm":
        oracle.set_alias_oob(0)
        o = oracle.oracle_sgbm(im1, im2, -25, 40)
        oracle.set_alias_oob(1)
        assert same(o["disp"], d)
    else:
        # the call sites' parameters: 'mgm' = MEDIAN=1; 'mgm_multi' = REMOVESMALLCC=25 (one scale, whole-pixel candidates: -S 6 and SUBPIX=2 are opt-in)
        kw = dict(median=1, remove_small_cc=0) if algo == "mgm" else dict(median=0, remove_small_cc=25, scales=1, subpix=1)
        kw["recursion"] = 2                               # the `mgm` binaries' aggregation as modelled (TSGM=3: three predecessors), both call sites since round 5
        o = oracle.oracle_census_sgm(im1, im2, -25, 40, params=oracle.census_params(**kw))
        assert same(o["disp"], d)
        conf = rio.read_image(str(tmp_path / "rectified_disp_confidence.tif"))
        assert same(o["conf"], conf)
    assert same(oracle.oracle_rejection_mask(d, im1, im2), m)


def test_tile_scheduler_and_file_shim_agree(hip, tmp_path):
    """The tile scheduler builds its matcher parameters from `algo` + cfg exactly like the file-level shim
    (block_matching.matcher_params): the same tile gives the same disparities through either door, and a cfg change
    reaches both."""
    from s2p_amd import block_matching as bm
    from s2p_amd import io as rio
    from s2p_amd import tiles as T
    from s2p_amd.config import cfg
    im1, im2, p1, p2 = _files(tmp_path)
    disp, mask = str(tmp_path / "d.tif"), str(tmp_path / "m.png")
    old = {k: cfg.get(k) for k in ("stereo_regularity_multiplier", "mgm_leftright_threshold", "hip_mgm_recursion")}
    try:
        for algo, over in (("mgm", {}), ("mgm", {"mgm_leftright_threshold": 2.0, "hip_mgm_recursion": 0}),
                           ("mgm_multi", {"stereo_regularity_multiplier": 2.0}), ("sgbm", {})):
            cfg.update(over)
            bm.compute_disparity_map(p1, p2, disp, mask, algo, -24, 39)
            got = T.match_tiles([T.Tile(0, im1, im2, -24, 39)], algo=algo, in_flight=1)[0]
            assert same(rio.read_image(disp), got), (algo, over)
            for k, v in old.items():
                if v is None:
                    cfg.pop(k, None)
                else:
                    cfg[k] = v
        cfg["stereo_regularity_multiplier"] = 1.3            # 10.4 / 41.6: rounded to 10 / 42
        assert (bm.matcher_params("mgm_multi")[1].P1, bm.matcher_params("mgm_multi")[1].P2) == (10, 42)
        bm.compute_disparity_map(p1, p2, disp, mask, "mgm_multi", -24, 39)
        cfg["stereo_regularity_multiplier"] = 5.0            # P2 = 160: beyond the byte e-volumes
        with pytest.raises(NotImplementedError):
            bm.compute_disparity_map(p1, p2, disp, mask, "mgm_multi", -24, 39)
        cfg["stereo_regularity_multiplier"] = 1.0
        cfg["mgm_nb_directions"] = 16                        # s2p/config.py:149: the knight's moves on top (round 4), through both doors
        for algo in ("mgm", "mgm_multi"):
            assert bm.matcher_params(algo)[1].nb_dir == 16
            bm.compute_disparity_map(p1, p2, disp, mask, algo, -24, 39)
            got = T.match_tiles([T.Tile(0, im1, im2, -24, 39)], algo=algo, in_flight=1)[0]
            assert same(rio.read_image(disp), got), algo
        cfg["hip_mgm_recursion"] = 0                         # ... but not as 1-D paths
        with pytest.raises(NotImplementedError):
            T.match_tiles([T.Tile(0, im1, im2, -24, 39)], algo="mgm", in_flight=1)
        cfg.pop("hip_mgm_recursion", None)
        cfg["mgm_nb_directions"] = 12
        with pytest.raises(NotImplementedError):
            T.match_tiles([T.Tile(0, im1, im2, -24, 39)], algo="mgm", in_flight=1)
    finally:
        cfg["mgm_nb_directions"] = 8
        for k, v in old.items():
            if v is None:
                cfg.pop(k, None)
            else:
                cfg[k] = v


def test_timeout_and_exit_code_contracts(hip, tmp_path):
    """tests/block_matching_test.py:10-21 expects subprocess.TimeoutExpired from a 1 s budget on a
    matcher that takes > 1 s on the CPU; the GPU matcher finishes in ~1 ms, so the contract is
    exercised with a zero budget.  A failing binary (exit 1) maps to CalledProcessError
    (tests/common_test.py:16-21)."""
    from s2p_amd import block_matching as bm
    _, _, p1, p2 = _files(tmp_path)
    disp, mask = str(tmp_path / "d.tif"), str(tmp_path / "m.png")
    with pytest.raises(subprocess.TimeoutExpired):
        bm.compute_disparity_map(p1, p2, disp, mask, "mgm_multi", -100, 100, timeout=0)
    with pytest.raises(subprocess.CalledProcessError):
        bm.compute_disparity_map(p1, p2, disp, mask, "sgbm", 7, 7)


def test_image_apply_homography_files(hip, oracle, tmp_path):
    from s2p_amd import common
    from s2p_amd import io as rio
    g = load_golden("warp_tile")
    w, h = (int(v) for v in g["size"])
    src = str(tmp_path / "img.tif")
    rio.write_image(src, g["src"])                          # uint16, like the reference's GeoTIFFs
    out = str(tmp_path / "rectified.tif")
    common.image_apply_homography(out, src, g["H"], w + 0.54, h + 0.54)     # float sizes are truncated (:180)
    r = rio.read_image(out)
    assert r.shape == (h, w)
    e = np.abs(r - g["expected"])[12:-12, 12:-12]
    assert e.mean() <= 0.02 and e.max() <= 0.15


def test_tiles_in_flight_on_one_gpu(hip, oracle):
    """Several tiles in flight on one GPU (one context = one HIP stream per worker thread) give the
    same maps as one-at-a-time calls, for both matchers."""
    from s2p_amd import tiles as T
    jobs = []
    for i in range(6):
        im1, im2 = synth_pair(70 + i, 64 + 8 * i, 120, lambda x, y: 3 + 4 * np.sin(x / 21.) * np.cos(y / 17.))
        jobs.append(T.Tile(i, im1, im2, -12, 19))
    from s2p_amd.block_matching import matcher_params
    for algo in ("mgm", "mgm_multi", "sgbm"):
        par = T.match_tiles(jobs, algo=algo, device=0, in_flight=3)
        params = matcher_params(algo)[1]
        for t in jobs:
            one = (hip.sgbm if algo == "sgbm" else hip.census_sgm)(t.im1, t.im2, t.disp_min, t.disp_max, params=params)["disp"]
            assert same(one, par[t.index])


# ---- BASELINE.json configs as single-tile parity cases --------------------------------------------
def test_config0_real_256_tile_sgbm_64_disparities_through_files(hip, oracle, tmp_path):
    """configs[0]: input_pair geometry, one 256x256 tile, sgbm matcher, 64 disparities -- through the
    file-level shim, against the REAL reference matcher when it travelled to this box."""
    from s2p_amd import block_matching as bm
    from s2p_amd import io as rio
    g = load_golden("mgm_tile")
    w, h = (int(v) for v in g["size"])
    sec = hip.warp(g["src"], g["H"], w, h)
    a = np.ascontiguousarray(g["ref"][100:356, 120:376])
    b = np.ascontiguousarray(sec[100:356, 120:376])
    p1, p2 = str(tmp_path / "rectified_ref.tif"), str(tmp_path / "rectified_sec.tif")
    rio.write_image(p1, a)
    rio.write_image(p2, b)
    disp, mask = str(tmp_path / "rectified_disp.tif"), str(tmp_path / "rectified_mask.png")
    bm.compute_disparity_map(p1, p2, disp, mask, "sgbm", -32, 32)
    d = rio.read_image(disp)
    ref = oracle.ref_sgbm(a, b, -32, 32) if oracle.have_ref() else None
    oracle.set_alias_oob(0)
    o = oracle.oracle_sgbm(a, b, -32, 32)
    oracle.set_alias_oob(1)
    assert same(o["disp"], d)
    if ref is not None:
        assert same(ref["disp"], d)
    assert same(oracle.oracle_rejection_mask(d, a, b), rio.read_image(mask, np.uint8))


def test_config2_512_tile_192_disparities_census(hip, oracle):
    """configs[2] tile shape: 512x512, 192 disparities (lane groups of 32 with 8 padding lanes)."""
    im1, im2 = synth_pair(12, 512, 512, lambda x, y: 60 * np.sin(2 * np.pi * x / 400.) * np.cos(2 * np.pi * y / 300.))
    kw = dict(median=0, remove_small_cc=25)                       # the 'mgm_multi' call's options
    r = hip.census_sgm(im1, im2, -96, 95, params=hip.default_census_params(**{"recursion": 0, **kw}))
    o = oracle.oracle_census_sgm(im1, im2, -96, 95, params=oracle.census_params(**kw))
    assert same(o["disp"], r["disp"]) and same(o["mask"], r["mask"])


def test_config3_1024_tile_256_disparities_both_matchers(hip, oracle):
    """configs[3] tile shape: 1024x1024 (here 1000x1000 as adjust_tile_size produces), 256 disparities."""
    im1, im2 = synth_pair(13, 1000, 1000, lambda x, y: 90 * np.sin(2 * np.pi * x / 700.) * np.cos(2 * np.pi * y / 500.))
    r = hip.census_sgm(im1, im2, -128, 127, params=hip.default_census_params(recursion=0))
    o = oracle.oracle_census_sgm(im1, im2, -128, 127, params=oracle.census_params(recursion=0))
    assert same(o["disp"], r["disp"])
    s = hip.sgbm(im1, im2, -128, 128)
    oracle.set_alias_oob(0)
    so = oracle.oracle_sgbm(im1, im2, -128, 128)
    oracle.set_alias_oob(1)
    assert same(so["disp"], s["disp"])


@pytest.mark.parametrize("radius", [2, 3, 5])
def test_mask_erosion(hip, oracle, tmp_path, radius):
    from s2p_amd import io as rio, masking
    rng = np.random.default_rng(radius)
    m = (rng.uniform(size=(70, 110)) > 0.03).astype(np.uint8)
    assert same(oracle.oracle_erode(m, radius), hip.erode_mask(m, radius))
    p = str(tmp_path / "rectified_mask.png")
    rio.write_image(p, m)
    masking.erosion(p, p, radius)                    # in place, like s2p/__init__.py:190
    assert same(oracle.oracle_erode(m, radius), rio.read_image(p, np.uint8))
    masking.erosion(p, p, 1)                         # radius < 2: no-op (masking.py:96)


def test_dev_entry_points_and_graph_replay(hip):
    """Device-resident calls (what schedulers and bench.py use), eager and as a replayed hipGraph, give
    the same maps as the host-buffer calls; the graph is captured once per call signature."""
    import ctypes
    L = hip
    lib = L.lib()
    H, W = 200, 256
    im1, im2 = synth_pair(91, H, W, lambda x, y: 5 + 7 * np.sin(x / 33.) * np.cos(y / 27.))
    want_c = L.census_sgm(im1, im2, -20, 27, params=L.default_census_params(recursion=0))
    want_s = L.sgbm(im1, im2, -20, 28)
    mem = DevMem()
    ctx = ctypes.c_void_p()
    L.check(lib.s2p_hip_ctx_create(0, None, ctypes.byref(ctx)))
    try:
        d1, d2 = mem.upload(im1), mem.upload(im2)
        disp, aux = mem.upload(np.zeros((H, W), np.float32)), mem.upload(np.zeros((H, W), np.float32))
        mask = mem.upload(np.zeros((H, W), np.uint8))
        pc, ps = L.default_census_params(recursion=0), L.default_sgbm_params()
        for graphs in (0, 1):
            L.check(lib.s2p_hip_ctx_use_graphs(ctx, graphs))
            for rep in range(3):                              # with graphs: rep 0 captures, 1-2 replay
                mem.fill(disp, H * W * 4, 0); mem.fill(aux, H * W * 4, 0); mem.fill(mask, H * W, 7)
                L.check(lib.s2p_hip_census_sgm_dev(ctx, d1, d2, W, H, -20, 27, ctypes.byref(pc), disp, aux, mask))
                L.check(lib.s2p_hip_ctx_sync(ctx))
                assert same(want_c["disp"], mem.download(disp, (H, W), np.float32))
                assert same(want_c["conf"], mem.download(aux, (H, W), np.float32))
                assert same(want_c["mask"], mem.download(mask, (H, W), np.uint8))
                L.check(lib.s2p_hip_sgbm_dev(ctx, d1, d2, W, H, -20, 28, ctypes.byref(ps), disp, aux, mask))
                L.check(lib.s2p_hip_ctx_sync(ctx))
                assert same(want_s["disp"], mem.download(disp, (H, W), np.float32))
                assert same(want_s["mask"], mem.download(mask, (H, W), np.uint8))
    finally:
        lib.s2p_hip_ctx_destroy(ctx)
        mem.free()


def test_hot_path_end_to_end_through_files(hip, tmp_path):
    """The whole path the way s2p drives it for one tile (s2p/__init__.py:102-196), on the reference's own
    tile: rectify both images (rectify_tail -> 2 x image_apply_homography), match ('mgm'), erode the mask.
    Inputs: the crops of input_pair/img_0{1,2}.tif held by the fixtures, written as uint16 TIFFs."""
    from s2p_amd import block_matching as bm, common, io as rio, masking, rectification
    g1, g2 = load_golden("warp_tile"), load_golden("mgm_tile")
    w, h = (int(v) for v in g1["size"])
    p1, p2 = str(tmp_path / "img_01.tif"), str(tmp_path / "img_02.tif")
    rio.write_image(p1, g1["src"])
    rio.write_image(p2, g2["src"])
    Tm = np.linalg.inv(common.matrix_translation(44, 5))                    # undo the margins stored in H_ref/H_sec
    H1, H2 = Tm @ g1["H"], Tm @ g2["H"]
    x, y = 500 - int(g1["crop"][0]), 150 - int(g1["crop"][1])              # ROI [500,150,350,350] in crop coordinates
    out1, out2 = str(tmp_path / "rectified_ref.tif"), str(tmp_path / "rectified_sec.tif")
    H1m, H2m, dm, dM = rectification.rectify_tail(p1, p2, out1, out2, H1, H2, x, y, 350, 350, -43.2, 30.1,
                                                  hmargin=10, vmargin=5)
    assert np.allclose(H1m, g1["H"]) and np.allclose(H2m, g2["H"])
    ref = rio.read_image(out1)
    assert ref.shape == (h, w)
    assert np.abs(ref - g1["expected"])[12:-12, 12:-12].mean() <= 0.02     # == the reference's rectified_ref.tif
    disp, mask = str(tmp_path / "rectified_disp.tif"), str(tmp_path / "rectified_mask.png")
    bm.compute_disparity_map(out1, out2, disp, mask, "mgm", dm, dM, timeout=600)
    d = rio.read_image(disp)
    m0 = rio.read_image(mask, np.uint8)
    masking.erosion(mask, mask, 2)                                         # cfg['msk_erosion'] = 2
    m = rio.read_image(mask, np.uint8)
    assert m.sum() < m0.sum() and np.all(m <= m0)
    both = np.isfinite(d) & np.isfinite(g2["disp"])
    e = np.abs(d[both] - g2["disp"][both])
    # vs the reference's rectified_disp.tif (mgm): the north_star bar, end to end from the two image crops through files
    assert (e <= 0.5).mean() >= 0.99 and (e <= 1.0).mean() >= 0.997
    assert os.path.exists(str(tmp_path / "rectified_disp_confidence.tif"))


@pytest.mark.parametrize("shape", [(3, 5000), (5000, 3), (2, 41000), (41000, 2), (1, 300), (300, 1), (40, 1300), (1300, 33),
                                   (65, 4096), (4096, 65), (128, 129), (129, 128), (64, 200), (200, 64), (4097, 70), (700, 1000), (193, 257)])
def test_warp_prefilter_line_lengths(hip, oracle, shape):
    """Every path of the recursive prefilter, bit-exact against the oracle: register-resident lines (both sides in
    (64, 4096]: 2 to 64 chunks per line, tails of 1 to 64 samples), every LDS block size (32 lines down to 1) for
    shorter / longer lines, and the long-line global-memory path (> 40960 samples)."""
    rng = np.random.default_rng(shape[0] * 7 + shape[1])
    src = rng.uniform(0, 1000, shape).astype(np.float32)
    if min(shape) > 8:
        src[5, 7] = np.nan
    sh, sw = shape
    H = np.array([[1.0, 0.001, 0.25], [-0.001, 1.0, -0.5], [0.0, 0.0, 1.0]])
    w, h = min(sw, 700), min(sh, 700)
    out = hip.warp(src, H, w, h)
    ref = oracle.oracle_warp(src, H, w, h)
    assert same(out, ref)


def test_bench_two_ranks_control_flow(hip):
    """bench.py under torch.distributed.run with 2 ranks, exactly as the driver launches it, on this 1-GPU box:
    both ranks pinned to device 0 and gloo in place of RCCL (test hooks S2P_BENCH_DEVICE / S2P_BENCH_BACKEND).
    Checks the N > 1 control flow: barrier + max-over-ranks timing, rank-0-only JSON line, whole-job value."""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, S2P_BENCH_DEVICE="0", S2P_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29653", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
           "--size", "256", "--ndisp", "64", "--batch", "4", "--batch-launch", "2", "--job-tiles", "6", "--pool", "2"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                           # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["warmup"] == 2 and d["scaling"] == "weak"
    assert "cpu_baseline" not in d and "mosaic_gather_ms" in d       # CPU baseline only at N = 1
    per_tile = 256 * 256 * 64 / 1e6
    assert d["config"]["tiles_per_step"] == 4 and d["config"]["tiles_per_call"] == 2 and d["roofline"]["tiles_per_launch"] == 2
    assert abs(d["value"] - per_tile * 4 * 6 * 2 / (d["ms_per_step"] * 6e-3)) / d["value"] < 1e-3     # whole-job aggregate, batches of 4 tiles
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["frac"] > 0 and d["roofline"]["kernel"] == "k_mgm_bands"
    j = d["job"]                                                     # the fixed tile list, split over the two ranks by the shared queue
    assert j["scaling"] == "strong" and j["tiles"] == 6 and sum(j["tiles_per_rank"]) == 6 and j["n_gpus"] == 2 and j["mosaic_valid"] > 0.5


def test_bench_eight_ranks_control_flow(hip):
    """The driver's `--gpus 8` launch on this 1-GPU box: 8 ranks pinned to device 0, gloo in place of RCCL.
    No scaling figure is read from it -- only that the 8-rank control flow (process group, shared work queue over 8 ranks x 3
    workers, batched library calls, dynamic mosaic gather, one JSON line) holds together."""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # (8 ranks + this process on ONE device is more than the library's process fence admits by default -- on the 8-GPU node every rank
    #  has a device of its own; the hook that pins them all to device 0 lifts the fence with it)
    env = dict(os.environ, S2P_BENCH_DEVICE="0", S2P_BENCH_BACKEND="gloo", S2P_HIP_MAX_PROCS_PER_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", "29671", os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
           "--size", "192", "--ndisp", "32", "--batch", "4", "--batch-launch", "2", "--job-tiles", "40", "--job-batch", "2", "--pool", "2", "--distinct", "2"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["steps"] == 2 and "cpu_baseline" not in d and "pool" not in d
    j = d["job"]
    assert j["n_gpus"] == 8 and j["tiles"] == 40 and sum(j["tiles_per_rank"]) == 40 and len(j["tiles_per_rank"]) == 8
    assert j["mosaic_backend"] == "gloo" and j["mosaic_valid"] > 0.5


def test_bench_gpus_2_without_a_launcher_relaunches_itself(hip):
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment -- the way the driver calls `--gpus 1` --
    re-executes itself under torch.distributed.run instead of asserting; one JSON line with n_gpus 2 comes out."""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, S2P_BENCH_DEVICE="0", S2P_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--size", "256", "--ndisp", "64",
           "--batch", "4", "--batch-launch", "2", "--no-job"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "re-launching" in r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak"
