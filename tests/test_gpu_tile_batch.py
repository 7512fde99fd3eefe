"""s2p_hip_tile_host_batch (several tiles of one shape through ONE library call, one batched matcher launch) against
s2p_hip_tile_host tile by tile, and the scheduler's batch mode against its one-tile-per-call mode: byte-identical."""
import numpy as np
import pytest

from helpers import same, tile_views

pytestmark = pytest.mark.gpu

PAD = 12


def _H(dx, dy):
    return np.array([[1.0, 0.0, -PAD + dx], [0.0, 1.0, -PAD + dy], [0.0, 0.0, 1.0]])


def _jobs(T, n, size, nd, seed0=7, sizes=None):
    dmin, dmax = -nd // 2, nd // 2 - 1
    jobs = []
    for k in range(n):
        s = sizes[k] if sizes else size
        v = tile_views(seed0 + 13 * k, s + 2 * PAD, nd, 2)
        jobs.append(T.TileJob(k, v[0], _H(0.25 + 0.125 * k, 0.5), v[1], _H(0.25, 0.5 - 0.0625 * k), s, s, dmin, dmax, erosion=k % 3))
    return jobs


@pytest.mark.parametrize("algo,size,nd,n", [("mgm", 320, 64, 5), ("mgm", 200, 256, 3), ("mgm_multi", 256, 64, 2)])
def test_batch_call_equals_single_calls(algo, size, nd, n):
    from s2p_amd import _lib
    from s2p_amd.block_matching import matcher_params, params_for_range
    from s2p_amd import tiles as T
    kind, params = matcher_params(algo, None)
    jobs = _jobs(T, n, size, nd)
    p = params_for_range(kind, params, jobs[0].disp_min, jobs[0].disp_max)
    kw = [dict(src1=j.src1, H1=j.H1, src2=j.src2, H2=j.H2, w=j.w, h=j.h, dmin=j.disp_min, dmax=j.disp_max, params=p, erosion=j.erosion) for j in jobs]
    single = [_lib.tile(algo=kind, **k) for k in kw]
    got = _lib.tile_batch(kw)
    assert len(got) == n
    for a, b in zip(got, single):
        assert sorted(a) == sorted(b)
        for key in a:
            assert same(a[key], b[key]) if a[key].dtype != np.uint8 else np.array_equal(a[key], b[key]), key
        assert np.isfinite(a["disp"]).mean() > (0.5 if 2 * nd <= size else 0.1)      # (256 disparities on a 200-px tile: most candidates leave the image)
    # recycled result buffers give the same bytes again
    again = _lib.tile_batch([dict(k, out=o) for k, o in zip(kw, got)])
    for a, b in zip(again, single):
        assert same(a["disp"], b["disp"]) and np.array_equal(a["mask"], b["mask"])


def test_batch_takes_tiles_of_different_shapes_where_the_matcher_can():
    """Since round 4 tiles of different sizes / ranges share a call in the single-scale MGM modes (per-tile geometry in the aggregation
    launch): byte-identical to single calls.  Parameter sets without that form (8-path, P2 > 115, more than one pyramid level) still refuse mixed shapes."""
    from s2p_amd import _lib, tiles as T
    jobs = _jobs(T, 3, 128, 32, sizes=[128, 144, 136])
    kw = [dict(src1=j.src1, H1=j.H1, src2=j.src2, H2=j.H2, w=j.w, h=j.h, dmin=j.disp_min, dmax=j.disp_max + k, erosion=j.erosion) for k, j in enumerate(jobs)]
    got = _lib.tile_batch(kw)                                  # default parameters: the 'mgm' call (recursion 2, single scale)
    for a, k in zip(got, kw):
        b = _lib.tile(**k)
        assert same(a["disp"], b["disp"]) and np.array_equal(a["mask"], b["mask"]) and same(a["rect2"], b["rect2"])
    for p in (_lib.default_census_params(recursion=0), _lib.default_census_params(recursion=1, P1=30, P2=120)):
        with pytest.raises(_lib.HipError, match="different sizes"):
            _lib.tile_batch([dict(k, params=p) for k in kw])
    with pytest.raises(_lib.HipError, match="differs from tile 0"):
        _lib.tile_batch([kw[0], dict(kw[1], params=_lib.default_census_params(P2=40))])
    # sgbm tiles (algo 0) are not batched: refused at the C boundary
    import ctypes
    descs = [_lib._tile_desc(**dict(kw[0], algo="sgbm")) for _ in range(2)]
    Tn, On = (_lib.TileDesc * 2)(), (_lib.TileOut * 2)()
    for i, d in enumerate(descs):
        Tn[i], On[i] = d[0], d[1]
    c = _lib.context(None)
    assert _lib.lib().s2p_hip_tile_host_batch(c, 2, Tn, On, ctypes.c_double(-1.0)) == _lib.BAD_ARGUMENT
    assert _lib.lib().s2p_hip_tile_host_batch(c, 0, Tn, On, ctypes.c_double(-1.0)) == _lib.BAD_ARGUMENT


def test_batch_with_triangulation_equals_single_calls(oracle):
    """The reference's own tile (tests/golden: input_pair windows, RPCs) three times in one batch -- as it is, with the second
    window shifted by half a pixel, and with another erosion radius: the triangulation inputs and outputs are per tile."""
    from s2p_amd import _lib
    from test_gpu_tile_pipeline import reference_tile
    g1, g2, g3, w, h, dmin, dmax, tri = reference_tile(oracle)
    shift = np.array([[1.0, 0.0, 0.5], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]])
    kw = [dict(src1=g1["src"], H1=g1["H"], src2=g2["src"], H2=g2["H"], w=w, h=h, dmin=dmin, dmax=dmax, erosion=2, tri=tri, want_rect=False),
          dict(src1=g1["src"], H1=g1["H"], src2=g2["src"], H2=shift @ g2["H"], w=w, h=h, dmin=dmin, dmax=dmax, erosion=0,
               tri=dict(tri, hb=shift @ tri["hb"]), want_rect=True),
          dict(src1=g1["src"], H1=g1["H"], src2=g2["src"], H2=g2["H"], w=w, h=h, dmin=dmin, dmax=dmax, erosion=5, want_rect=False)]
    single = [_lib.tile(**k) for k in kw]
    got = _lib.tile_batch(kw)
    for a, b in zip(got, single):
        assert sorted(a) == sorted(b)
        for key in a:
            assert np.array_equal(a[key], b[key]) if a[key].dtype == np.uint8 else same(a[key], b[key]), key
    assert np.isfinite(got[0]["lonlatalt"]).any() and not same(got[0]["disp"], got[1]["disp"])


@pytest.mark.parametrize("sink", [False, True])
def test_queue_in_batches_equals_queue_one_by_one(sink):
    """11 tiles, the 4th and the last two of another size (border tiles), through process_queue with batch = 4 on two streams:
    the same bytes as one tile per call, with and without a streaming sink (recycled result buffers)."""
    from s2p_amd import tiles as T
    sizes = [256] * 11
    sizes[3] = sizes[9] = sizes[10] = 192
    jobs = _jobs(T, 11, 256, 64, sizes=sizes)
    ref = T.process_queue(jobs, T.WorkQueue(11), algo="mgm", in_flight=2)
    if not sink:
        got = T.process_queue(jobs, T.WorkQueue(11, chunk=4), algo="mgm", in_flight=2, batch=4)
        got = {i: (r["disp"], r["mask"]) for i, r in got.items()}
    else:
        got = {}

        def take(job, r):
            got[job.index] = (r["disp"].copy(), r["mask"].copy())
        out = T.process_queue(jobs, T.WorkQueue(11, chunk=4), algo="mgm", in_flight=2, batch=4, sink=take)
        assert all(v is None for v in out.values()) and sorted(out) == list(range(11))
    assert sorted(got) == list(range(11))
    for i in range(11):
        assert same(got[i][0], ref[i]["disp"]) and np.array_equal(got[i][1], ref[i]["mask"]), i
