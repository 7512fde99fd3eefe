"""GPU side of the broker path (s2p_amd/broker.py): the host-batch entry it calls and page-locking of foreign memory."""
import ctypes
import mmap
import os

import numpy as np
import pytest

from helpers import same, synth_pair
from s2p_amd import _lib

pytestmark = pytest.mark.gpu


def _pairs(n, h, w, amp=9.0):
    return [synth_pair(700 + k, h, w, lambda x, y: 5 + amp * np.sin(x / 41.) * np.cos(y / 31.)) for k in range(n)]


@pytest.mark.parametrize("rec,n,h,w,dmin,dmax", [(2, 3, 150, 200, -20, 27), (1, 5, 96, 160, -8, 39), (0, 2, 96, 160, -8, 39), (2, 8, 256, 256, -32, 31)])
def test_host_batch_equals_single_host_calls(rec, n, h, w, dmin, dmax):
    p = _lib.default_census_params(recursion=rec, median=1)
    pairs = _pairs(n, h, w)
    singles = [_lib.census_sgm(a, b, dmin, dmax, params=p) for a, b in pairs]
    disp = [np.full((h, w), 7, np.float32) for _ in range(n)]
    conf = [np.full((h, w), 7, np.float32) for _ in range(n)]
    mask = [np.full((h, w), 7, np.uint8) for _ in range(n)]
    conf_addr = [c.ctypes.data for c in conf]
    conf_addr[-1] = 0                                         # a slot that does not want its confidence image
    _lib.census_sgm_host_batch(_lib.context(), [a.ctypes.data for a, _ in pairs], [b.ctypes.data for _, b in pairs], w, h, dmin, dmax, p,
                               [d.ctypes.data for d in disp], conf_addr, [m.ctypes.data for m in mask])
    for k in range(n):
        assert same(singles[k]["disp"], disp[k]) and np.array_equal(singles[k]["mask"], mask[k]), k
        if k < n - 1:
            assert same(singles[k]["conf"], conf[k]), k
    assert np.all(conf[-1] == 7)


def test_multi_scale_parameters_run_through_the_batch_entry_one_by_one():
    p = _lib.default_census_params(recursion=1, median=0, scales=6, remove_small_cc=25)
    pairs = _pairs(2, 300, 400)
    singles = [_lib.census_sgm(a, b, -30, 33, params=p) for a, b in pairs]
    disp = [np.empty((300, 400), np.float32) for _ in range(2)]
    _lib.census_sgm_host_batch(_lib.context(), [a.ctypes.data for a, _ in pairs], [b.ctypes.data for _, b in pairs], 400, 300, -30, 33, p,
                               [d.ctypes.data for d in disp], [0, 0], [0, 0])
    for k in range(2):
        assert same(singles[k]["disp"], disp[k])


def test_a_memfd_mapping_can_be_page_locked_and_used_as_io_buffer():
    """What the broker does with a worker's arena: map the descriptor, hipHostRegister it, run a call on addresses inside it."""
    h, w = 128, 192
    size = 8 << 20
    fd = os.memfd_create("s2p_test_arena")
    os.ftruncate(fd, size)
    mm = mmap.mmap(fd, size)
    arr = np.frombuffer(mm, np.uint8)
    _lib.host_register(arr.ctypes.data, size)                 # raises HipError if the driver refuses this kind of memory
    try:
        (a, b), = _pairs(1, h, w)
        n4 = h * w * 4
        v = lambda o, dt, cnt: arr[o:o + cnt * np.dtype(dt).itemsize].view(dt).reshape(h, w)
        v(0, np.float32, h * w)[:] = a
        v(n4, np.float32, h * w)[:] = b
        p = _lib.default_census_params(recursion=2, median=1)
        base = arr.ctypes.data
        _lib.census_sgm_host_batch(_lib.context(), [base], [base + n4], w, h, -16, 15, p, [base + 2 * n4], [base + 3 * n4], [base + 4 * n4])
        ref = _lib.census_sgm(a, b, -16, 15, params=p)
        assert same(ref["disp"], v(2 * n4, np.float32, h * w)) and same(ref["conf"], v(3 * n4, np.float32, h * w))
        assert np.array_equal(ref["mask"], v(4 * n4, np.uint8, h * w))
    finally:
        _lib.host_unregister(arr.ctypes.data)
        del arr
        mm.close()
        os.close(fd)


def _mirror_chain(args):
    """One Pool worker's share of a tile's steps through the FILE-level mirrors: rectify (image_apply_homography), match, erode the
    mask (what s2p.stereo_matching does right after the matcher, s2p/__init__.py:189-190), triangulate, clean a height map, merge."""
    d, k = args
    import os
    import sys
    from helpers import load_golden
    from s2p_amd import _lib, block_matching as bm, common, fusion, masking, triangulation
    from s2p_amd import io as rio
    g1, g2 = load_golden("warp_tile"), load_golden("mgm_tile")
    w, h = (int(v) for v in g1["size"])
    p = lambda n: os.path.join(d, n.replace(".", "_%d." % k))
    rio.write_image(p("src1.tif"), g1["src"].astype(np.float32))
    rio.write_image(p("src2.tif"), g2["src"].astype(np.float32))
    so, sys.stdout = sys.stdout, open(os.devnull, "w")
    try:
        common.image_apply_homography(p("r1.tif"), p("src1.tif"), g1["H"], w, h)
        common.image_apply_homography(p("r2.tif"), p("src2.tif"), g2["H"], w, h)
        bm.compute_disparity_map(p("r1.tif"), p("r2.tif"), p("disp.tif"), p("mask.png"), "mgm", -40 - k, 30, timeout=600)
        masking.erosion(p("mask.png"), p("mask.png"), 2)
        common.cargarse_basura(p("disp.tif"), p("clean.tif"))
        fusion.merge_n(p("fused.tif"), [p("disp.tif"), p("clean.tif")], [0.0, 0.5], averaging="average_if_close", threshold=3)
    finally:
        sys.stdout.close()
        sys.stdout = so
    disp = rio.read_image(p("disp.tif"))
    n = triangulation.count_3d_neighbors(np.dstack([disp, disp, disp]).astype(np.float64), 3.0, 2)
    out = {n_: rio.read_image(p(n_ + ".tif")) for n_ in ("r1", "r2", "disp", "clean", "fused")}
    out["conf"] = rio.read_image(os.path.splitext(p("disp.tif"))[0] + "_confidence.tif")
    out["mask"] = rio.read_image(p("mask.png"), np.uint8)
    out["neighbors"] = n
    return k, out, len(_lib._ctx)                            # ... and how many HIP contexts this process created


def test_file_level_mirrors_in_pool_workers_go_through_the_broker(tmp_path):
    """Every array-level call of the mirrors travels through the broker when the caller is a Pool worker (@broker.remote): the
    worker never initialises HIP (no context of its own), and each file equals what the same chain writes in a plain process."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = '''
import sys, os, json, pickle
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import multiprocessing as mp
import numpy as np
import test_gpu_broker as t
from s2p_amd import broker
d = %r
mode = sys.argv[1]
if mode == "pool":
    os.environ["S2P_HIP_BROKER_DIR"] = os.path.join(d, "broker")
    with mp.get_context("fork").Pool(3) as pool:
        res = pool.map(t._mirror_chain, [(d, k) for k in range(3)])
    st = broker.stats(0)
    broker._clients[(os.getpid(), 0)].close(); broker.shutdown(0)
    assert all(nctx == 0 for _, _, nctx in res), "a Pool worker created its own HIP context"
    assert st["fn_calls"] >= 3 * 5 and st["requests"] == 3 and st["errors"] == 0, st
else:
    res = [t._mirror_chain((d, k)) for k in range(3)]
    assert all(nctx >= 1 for _, _, nctx in res)
pickle.dump({k: o for k, o, _ in res}, open(os.path.join(d, mode + ".pkl"), "wb"))
print("ok", mode)
''' % (root, root, str(tmp_path))
    import pickle
    for mode in ("pool", "plain"):
        r = subprocess.run([sys.executable, "-c", code, mode], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "ok " + mode in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
    a = pickle.load(open(os.path.join(str(tmp_path), "pool.pkl"), "rb"))
    b = pickle.load(open(os.path.join(str(tmp_path), "plain.pkl"), "rb"))
    for k in range(3):
        for name in b[k]:
            assert np.array_equal(a[k][name], b[k][name], equal_nan=True), (k, name)


def test_a_worker_survives_a_broker_that_left(tmp_path, monkeypatch):
    """Between two steps of a job the broker may have left (idle) or been killed: the next call of a worker that still holds the old
    connection starts a new broker, hands it the arena again and gets the same bytes."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = '''
import sys, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
os.environ["S2P_HIP_BROKER"] = "1"; os.environ["S2P_HIP_BROKER_DIR"] = %r
import numpy as np
from helpers import synth_pair
from s2p_amd import broker, block_matching as bm, io as rio
d = %r
a, b = synth_pair(77, 200, 260, lambda x, y: 5 + 6 * np.sin(x / 31.) * np.cos(y / 27.))
rio.write_image(os.path.join(d, "a.tif"), a); rio.write_image(os.path.join(d, "b.tif"), b)
sys.stdout = open(os.devnull, "w")
def run(tag):
    bm.compute_disparity_map(os.path.join(d, "a.tif"), os.path.join(d, "b.tif"), os.path.join(d, tag + ".tif"), os.path.join(d, tag + ".png"), "mgm", -16, 15)
    return rio.read_image(os.path.join(d, tag + ".tif")), rio.read_image(os.path.join(d, tag + ".png"), np.uint8)
d1, m1 = run("one")
pid1 = broker.client(0).hello["pid"]
os.kill(pid1, 9)                                              # the broker dies; this process still holds its connection
import time; time.sleep(0.3)
d2, m2 = run("two")
pid2 = broker.client(0).hello["pid"]
assert pid2 != pid1
assert np.array_equal(d1, d2, equal_nan=True) and np.array_equal(m1, m2)
broker.client(0).close(); broker.shutdown(0)
sys.__stdout__.write("ok\\n")
''' % (root, root, str(tmp_path / "broker"), str(tmp_path))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-1000:] + r.stderr[-3000:]
