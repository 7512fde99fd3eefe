"""CPU tests of the Python mirror of the reference interface (no GPU needed): the contracts of
tests/block_matching_test.py:24-36 (range error BEFORE any work) and the host-side geometry of the
resampling shim."""
import numpy as np
import pytest


def _write_pair(tmp_path, w=64, h=32):
    from s2p_amd import io as rio
    rng = np.random.default_rng(0)
    a = rng.uniform(0, 1000, (h, w)).astype(np.float32)
    p1, p2 = str(tmp_path / "a.tif"), str(tmp_path / "b.tif")
    rio.write_image(p1, a)
    rio.write_image(p2, a)
    return p1, p2


def test_io_roundtrip(tmp_path):
    from s2p_amd import io as rio
    a = np.random.default_rng(1).uniform(-5, 5, (17, 23)).astype(np.float32)
    a[3, 4] = np.nan
    p = str(tmp_path / "x.tif")
    rio.write_image(p, a)
    assert rio.image_size(p) == (23, 17)
    b = rio.read_image(p)
    assert b.dtype == np.float32 and np.array_equal(a, b, equal_nan=True)
    m = (a > 0).astype(np.uint8)
    q = str(tmp_path / "m.png")
    rio.write_image(q, m)
    assert np.array_equal(rio.read_image(q, np.uint8), m)
    assert np.array_equal(rio.read_window(p, 2, 1, 9, 6), a[1:6, 2:9], equal_nan=True)


def _child_writes(args):
    from s2p_amd import io as rio
    d, m, a, k = args
    rio.write_images([(d + "/c%d.tif" % k, a), (d + "/e%d.tif" % k, a), (d + "/c%d.png" % k, m)])
    return bool(np.array_equal(rio.read_image(d + "/c%d.png" % k, np.uint8), m))


def test_pooled_encoders_large_mask_and_fork(tmp_path):
    """A mask large enough to be deflated in pieces reads back identically through an independent decoder (PIL), and a
    worker forked AFTER the parent used the encoder pool gets a pool of its own (the orchestrator forks: s2p/parallel.py)."""
    import multiprocessing
    from PIL import Image
    from s2p_amd import io as rio
    rng = np.random.default_rng(2)
    m = (rng.uniform(size=(700, 900)) < 0.9).astype(np.uint8)
    m[:, :50] = 0
    a = rng.uniform(0, 1, m.shape).astype(np.float32)
    d = str(tmp_path)
    rio.write_images([(d + "/d.tif", a), (d + "/e.tif", a), (d + "/m.png", m)])
    with Image.open(d + "/m.png") as im:
        assert im.mode == "L" and np.array_equal(np.array(im), m)
    got = rio.read_images([d + "/d.tif", d + "/m.png", d + "/e.tif"], np.float32)
    assert np.array_equal(got[0], a) and np.array_equal(got[1], m) and np.array_equal(got[2], a)
    with multiprocessing.get_context("fork").Pool(2) as pool:
        assert all(pool.map(_child_writes, [(d, m, a, k) for k in range(2)]))


@pytest.mark.parametrize("algo", ["sgbm", "mgm", "mgm_multi"])
def test_max_disp_range_error_before_any_work(tmp_path, algo):
    """tests/block_matching_test.py:24-36: max_disp_range=10 with a range of 200 raises, and it does
    so before the matcher (here: before the GPU library) is touched -- so this passes without a GPU."""
    from s2p_amd import block_matching as bm
    p1, p2 = _write_pair(tmp_path)
    with pytest.raises(bm.MaxDisparityRangeError):
        bm.compute_disparity_map(p1, p2, str(tmp_path / "d.tif"), str(tmp_path / "m.png"), algo,
                                 -100, 100, max_disp_range=10)


def test_range_is_clamped_to_image_width_then_rounded(tmp_path, monkeypatch):
    """s2p/block_matching.py:61-74: a range wider than the image is recentred to the width; bounds
    are floored / ceiled.  The matcher call is intercepted (no GPU)."""
    from s2p_amd import block_matching as bm
    p1, p2 = _write_pair(tmp_path, w=64)
    seen = {}

    def fake_sgbm(a, b, dmin, dmax, **kw):
        seen["range"] = (dmin, dmax)
        z = np.zeros_like(a)
        return dict(disp=z, cost=None, mask=z.astype(np.uint8))
    monkeypatch.setattr(bm._lib, "sgbm", fake_sgbm)
    bm.compute_disparity_map(p1, p2, str(tmp_path / "d.tif"), str(tmp_path / "m.png"), "sgbm", -100.5, 100.2)
    assert seen["range"] == (-32, 31)            # centre -0.15: int(-32.15) = -32, int(31.85) = 31 (int() truncates)
    bm.compute_disparity_map(p1, p2, str(tmp_path / "d.tif"), str(tmp_path / "m.png"), "sgbm", -3.5, 7.2)
    assert seen["range"] == (-4, 8)


def test_unhandled_algo_is_refused(tmp_path):
    from s2p_amd import block_matching as bm
    p1, p2 = _write_pair(tmp_path)
    with pytest.raises(NotImplementedError):
        bm.compute_disparity_map(p1, p2, str(tmp_path / "d.tif"), str(tmp_path / "m.png"), "tvl1", -5, 5)


def test_source_window_and_points():
    from s2p_amd import common
    H = np.array([[0.98, -0.17, 40.0], [0.17, 0.98, -12.0], [0, 0, 1.0]])
    pts = np.array([[0, 0], [100, 0], [0, 50.0]])
    q = common.points_apply_homography(H, pts)
    assert np.allclose(common.points_apply_homography(np.linalg.inv(H), q), pts)
    x0, y0, x1, y1 = common.source_window(H, 200, 100, 1000, 800)
    Hi = np.linalg.inv(H)
    c = common.points_apply_homography(Hi, [[0, 0], [200, 0], [0, 100], [200, 100]])
    assert x0 <= max(np.floor(c[:, 0].min()) - common.MARGIN, 0) and x1 >= min(np.ceil(c[:, 0].max()) + common.MARGIN, 1000)
    assert y0 <= max(np.floor(c[:, 1].min()) - common.MARGIN, 0) and y1 >= min(np.ceil(c[:, 1].max()) + common.MARGIN, 800)


def test_rectify_tail_geometry_matches_the_reference_tile(monkeypatch):
    """The stored tile of tests/data/input_triangulation/pair_1 (ROI [500,150,350,350]): H_ref.txt is
    the final H1 = T(hmargin, vmargin) . H, it maps the ROI bounding box to origin (44, 5), and the
    float size 503.54 x 425.54 is truncated by "%d" to the 503 x 425 of rectified_ref.tif
    (SURVEY.md App. D).  Undo the margins, run the tail, and recover exactly that."""
    from helpers import load_golden
    from s2p_amd import common, rectification
    g = load_golden("warp_tile")
    x0c, y0c = int(g["crop"][0]), int(g["crop"][1])
    Href = g["H"] @ np.linalg.inv(common.matrix_translation(x0c, y0c))      # H_ref.txt (crop offset removed)
    H = np.linalg.inv(common.matrix_translation(44, 5)) @ Href              # before the margins
    calls = []
    monkeypatch.setattr(common, "image_apply_homography", lambda out, im, Hm, w, h: calls.append((out, im, Hm, w, h)))
    H1, H2, dm, dM = rectification.rectify_tail("a.tif", "b.tif", "o1.tif", "o2.tif", H, H, 500, 150, 350, 350,
                                                -43.2, 30.1, hmargin=10, vmargin=5)
    assert np.allclose(H1, Href, atol=1e-9) and (dm, dM) == (-43.2, 30.1)
    assert len(calls) == 2 and calls[0][0] == "o1.tif" and calls[1][1] == "b.tif"
    w, h = calls[0][3], calls[0][4]
    assert (int(w), int(h)) == tuple(int(v) for v in g["size"]) == (503, 425)


def test_fusion_merge_n_file_contract(tmp_path, monkeypatch, oracle):
    """s2p_amd.fusion.merge_n (mirror of s2p/fusion.py:26-68): reads every input, hands arrays + offsets + operator +
    threshold to the library, writes a float32 map at `output`, and with debug=True the <input>_registered.tif files.
    The GPU call is replaced by the numpy oracle here (host logic only; the kernel's parity is a GPU test)."""
    from s2p_amd import fusion, io as rio, _lib
    rng = np.random.default_rng(3)
    maps = [rng.uniform(10, 20, (12, 17)).astype(np.float32) for _ in range(3)]
    maps[1][2, 3] = np.nan
    paths = []
    for i, m in enumerate(maps):
        p = str(tmp_path / ("h%d.tif" % i))
        rio.write_image(p, m)
        paths.append(p)
    seen = {}

    def fake(images, offsets, averaging="average_if_close", threshold=1, device=None):
        seen.update(n=len(images), averaging=averaging, threshold=threshold, offsets=list(offsets))
        return oracle.oracle_merge_n(images, offsets, averaging, threshold)
    monkeypatch.setattr(_lib, "merge_n", fake)
    out = str(tmp_path / "height_map.tif")
    fusion.merge_n(out, paths, [0.5, -0.25, 0.0], averaging="np.nanmedian", threshold=7, debug=True)
    assert seen == dict(n=3, averaging="np.nanmedian", threshold=7, offsets=[0.5, -0.25, 0.0])
    got = rio.read_image(out)
    assert got.dtype == np.float32 and np.array_equal(got, oracle.oracle_merge_n(maps, [0.5, -0.25, 0.0], "np.nanmedian", 7), equal_nan=True)
    reg = rio.read_image(str(tmp_path / "h1_registered.tif"))
    assert np.allclose(reg[0, 0], maps[1][0, 0] + 0.25 + np.mean([0.5, -0.25, 0.0]), atol=1e-5) and np.isnan(reg[2, 3])
    fusion.merge_n(str(tmp_path / "none.tif"), [], [])                      # nothing to merge: no file, no call
    assert not (tmp_path / "none.tif").exists()
    with pytest.raises(AssertionError):
        fusion.merge_n(out, paths, [0.0])


def test_half_pixel_grid_falls_back_on_very_wide_ranges():
    """'mgm_multi' asks for SUBPIX=2; beyond 511 px of range that would be more than the 1024 candidates the library takes:
    whole-pixel candidates are used instead of refusing the tile."""
    import ctypes
    from s2p_amd import block_matching as bm

    class P(ctypes.Structure):
        _fields_ = [("subpix", ctypes.c_int), ("scales", ctypes.c_int)]
    p = P(2, 6)
    assert bm.params_for_range("census", p, -200, 200).subpix == 2 and bm.params_for_range("census", p, -200, 200) is p
    q = bm.params_for_range("census", p, -300, 300)
    assert q.subpix == 1 and q.scales == 6 and p.subpix == 2                     # a copy: the shared parameters stay as they are
    assert bm.params_for_range("sgbm", p, -300, 300) is p


def test_triangulation_mirrors_keep_the_reference_signatures():
    """s2p/triangulation.py:85-86, 165, 220, 304, 331, 346: the functions a maintainer swaps by import take the reference's arguments."""
    import inspect
    from s2p_amd import triangulation as t
    names = lambda f: [p for p in inspect.signature(f).parameters if p != "device"]
    assert names(t.disp_to_xyz) == ["rpc1", "rpc2", "H1", "H2", "disp", "mask_rect", "img_bbx", "mask_orig", "A", "out_crs"]
    assert names(t.stereo_corresp_to_xyz) == ["rpc1", "rpc2", "pts1", "pts2", "out_crs"]
    assert names(t.height_map_to_xyz) == ["heights", "rpc", "off_x", "off_y", "out_crs"]
    assert names(t.height_map) == ["x", "y", "w", "h", "rpc1", "rpc2", "H1", "H2", "disp", "mask", "mask_orig", "A"]
    assert names(t.remove_isolated_3d_points) == ["xyz", "r", "p", "n", "q"] and names(t.filter_xyz) == ["xyz", "r", "n", "img_gsd"]
    a = np.array([[[2.35, 48.85, 100.0], [np.nan, np.nan, np.nan]]])
    assert t._to_crs(a, None) is a and t._to_crs(a, "EPSG:4979") is a
    u = t._to_crs(a, "epsg:32631")
    assert abs(u[0, 0, 0] - 452314.9) < 1.0 and abs(u[0, 0, 1] - 5410984.9) < 1.0 and u[0, 0, 2] == 100.0 and np.isnan(u[0, 1]).all()
    with pytest.raises(NotImplementedError):
        t._to_crs(a, "epsg:2154")


def _raise_hip_error(i):
    from s2p_amd import _lib
    if i == 3:
        raise _lib.HipError(_lib.UNSUPPORTED, "boom %d" % i)
    return i


def test_a_workers_hip_error_reaches_the_pools_parent():
    """s2p/parallel.py:100-105: a failing worker surfaces as an exception in r.get().  Exceptions cross the process boundary as pickles;
    until round 5 HipError (two-argument __init__) could not be unpickled, the parent's result-handler thread died on it and every
    later result of that Pool was lost -- r.get(timeout) answered TimeoutError: round 4's "worker that never came back"."""
    import multiprocessing as mp
    import pickle
    from s2p_amd import _lib, broker
    e = pickle.loads(pickle.dumps(_lib.HipError(_lib.TIMEOUT, "late")))
    assert isinstance(e, _lib.HipError) and e.code == _lib.TIMEOUT and "late" in str(e) and str(e).count("libs2p_hip") == 1
    assert isinstance(pickle.loads(pickle.dumps(broker.BrokerError("x"))), broker.BrokerError)
    with mp.get_context("fork").Pool(3) as pool:
        rs = [pool.apply_async(_raise_hip_error, (i,)) for i in range(8)]
        got = []
        for r in rs:
            try:
                got.append(r.get(20))
            except _lib.HipError as ex:
                got.append(("hip", ex.code))
    assert got == [0, 1, 2, ("hip", _lib.UNSUPPORTED), 4, 5, 6, 7]


def test_the_mgm_multi_deviation_is_announced_once_per_job(tmp_path, monkeypatch, capfd):
    """'mgm_multi' does not run what its call site passes (one scale, whole pixels where the reference gives -S 6, SUBPIX=2);
    the shim says so on stderr -- once per job (a marker named after the job's parent process), not once per Pool worker -- and names
    what the choice was fitted to: an artefact of plain `mgm`."""
    import multiprocessing as mp
    from s2p_amd import block_matching as bm
    monkeypatch.setenv("S2P_HIP_BROKER_DIR", str(tmp_path))
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        bm.matcher_params("mgm_multi")
        bm.matcher_params("mgm_multi")
        err = capfd.readouterr().err
        assert err.count("NOTICE s2p_amd: 'mgm_multi'") == 1 and "produced by plain `mgm`, NOT by mgm_multi" in err and "hip_mgm_multi_scales" in err
        ctx = mp.get_context("fork")
        with ctx.Pool(3) as pool:                                 # workers of a job whose parent has said it already: quiet
            pool.map(_notice_worker, range(6))
        assert capfd.readouterr().err.count("NOTICE s2p_amd") == 0
        other = tmp_path / "job2"
        other.mkdir(mode=0o700)
        monkeypatch.setenv("S2P_HIP_BROKER_DIR", str(other))       # (a job whose parent never calls the matcher itself: the orchestrator)
        with ctx.Pool(3) as pool:                                 # one notice between its workers, not one each
            pool.map(_notice_worker, range(6))
        assert capfd.readouterr().err.count("NOTICE s2p_amd") == 1
        bm.matcher_params("mgm_multi", dict(bm.cfg, hip_mgm_multi_scales=6, hip_mgm_multi_subpix=2))     # the call site's own semantics: nothing to announce
        assert "NOTICE" not in capfd.readouterr().err


def _notice_worker(i):
    import warnings
    from s2p_amd import block_matching as bm
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        bm.matcher_params("mgm_multi")
    return i


def test_the_cpu_budget_of_the_box_is_read_from_its_control_group(tmp_path):
    """Round 6: the GPU boxes show 256 hardware threads and grant 16 CPUs' worth of time (cgroup v2 cpu.max = "1600000 100000"); bench.py
    states the quota its many-process CPU baseline ran under, bench_pool.py the CPU a Pool used against it."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def load(name):
        spec = importlib.util.spec_from_file_location(name, os.path.join(root, name + ".py"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m
    bench, bp = load("bench"), load("bench_pool")
    (tmp_path / "cpu.max").write_text("1600000 100000\n")
    (tmp_path / "cpu.stat").write_text("usage_usec 14220261\nuser_usec 1\nsystem_usec 2\nnr_periods 391\nnr_throttled 4\nthrottled_usec 13948790\n")
    assert bench.cpu_quota(str(tmp_path)) == 16.0
    assert bp.cgroup_cpu(str(tmp_path)) == (16.0, 14220261, 13948790, 4)
    (tmp_path / "cpu.max").write_text("max 100000\n")
    assert bench.cpu_quota(str(tmp_path)) is None and bp.cgroup_cpu(str(tmp_path))[0] is None
    v1 = tmp_path / "v1"
    (v1 / "cpu").mkdir(parents=True)
    (v1 / "cpu" / "cpu.cfs_quota_us").write_text("400000\n")
    (v1 / "cpu" / "cpu.cfs_period_us").write_text("100000\n")
    assert bench.cpu_quota(str(v1)) == 4.0
    assert bench.cpu_quota(str(tmp_path / "nothing")) is None and bp.cgroup_cpu(str(tmp_path / "nothing")) == (None, None, None, None)
