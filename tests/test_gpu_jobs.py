"""Job-level tests of BASELINE configs[3] / configs[4]: seeded tiles of the job shapes through the
tile scheduler (shared work queue, tiles in flight on separate HIP streams, dynamic mosaic gather) in the drop-in's default
mode (MGM recursion), checked against the CPU oracle and against a serial run; and the MGM mode against the oracle at the
full tile shapes of configs[1] and configs[3]."""
import numpy as np
import pytest

from helpers import SHIM_RECURSION, same, synth_pair, tile_views

pytestmark = pytest.mark.gpu

PAD = 12
HS = np.array([[1.0, 0.0, -PAD + 0.25], [0.0, 1.0, -PAD + 0.5], [0.0, 0.0, 1.0]])     # the jobs' rectifying map (bench.py run_job)


def _oracle_tile(oracle, v0, v1, size, dmin, dmax, params):
    r1, r2 = oracle.oracle_warp(v0, HS, size, size), oracle.oracle_warp(v1, HS, size, size)
    return oracle.oracle_census_sgm(r1, r2, dmin, dmax, params=params)


def test_config4_shard_through_the_queue(oracle):
    """8 tiles of the configs[3] job (1000 x 1000, 256 disparities, seed = 1000 ty + tx): host windows -> rectify -> match
    (MGM recursion) -> mask -> host through process_queue with 3 tiles in flight, then the dynamic mosaic gather.  Two tiles
    equal the oracle bit for bit, every tile equals a one-at-a-time run, the mosaic equals its serial assembly."""
    from s2p_amd import tiles as T
    size, nd, n = 1000, 256, 8
    dmin, dmax = -nd // 2, nd // 2 - 1
    views = [tile_views(1000 * (k // 4) + (k % 4) * 5, size + 2 * PAD, nd, 2) for k in range(n)]
    jobs = [T.TileJob(i, v[0], HS, v[1], HS, size, size, dmin, dmax) for i, v in enumerate(views)]
    res = T.process_queue(jobs, T.WorkQueue(n), algo="mgm", in_flight=3)
    assert sorted(res) == list(range(n))
    serial = T.process_tiles(jobs, algo="mgm", in_flight=1)
    for i in range(n):
        assert same(res[i]["disp"], serial[i]["disp"]) and np.array_equal(res[i]["mask"], serial[i]["mask"])
        assert np.isfinite(res[i]["disp"]).mean() > 0.9
    pm = oracle.census_params(recursion=SHIM_RECURSION)
    for i in (1, 6):
        o = _oracle_tile(oracle, views[i][0], views[i][1], size, dmin, dmax, pm)
        assert same(res[i]["disp"], o["disp"]) and np.array_equal(res[i]["mask"], o["mask"])
    dec = 4
    ts = size // dec
    layout = [((i // 4) * ts, (i % 4) * ts, ts, ts) for i in range(n)]
    local = {i: res[i]["disp"][::dec, ::dec].copy() for i in range(n)}
    mosaic = T.gather_mosaic(local, layout, (2 * ts, 4 * ts), dynamic=True)
    want = np.full((2 * ts, 4 * ts), np.nan, np.float32)
    for i, (y0, x0, h, w) in enumerate(layout):
        want[y0:y0 + h, x0:x0 + w] = serial[i]["disp"][::dec, ::dec]
    assert same(mosaic, want)


def test_config5_shard_two_pairs_and_fusion(oracle):
    """A shard of the configs[4] job: 4 tiles x 2 pairs (640 x 640, 128 disparities) through the queue, per-pair MGM matching,
    then fusion.merge_n per tile; one tile checked end to end against the oracle chain (resampler, matcher, merge_n)."""
    from s2p_amd import _lib, tiles as T
    size, nd, n = 640, 128, 4
    dmin, dmax = -nd // 2, nd // 2 - 1
    views = [tile_views(1000 * k + 3, size + 2 * PAD, nd, 3) for k in range(n)]
    jobs = [T.TileJob(2 * i + p, v[0], HS, v[1 + p], HS, size, size, dmin, dmax) for i, v in enumerate(views) for p in range(2)]
    res = T.process_queue(jobs, T.WorkQueue(len(jobs)), algo="mgm", in_flight=3)
    fused = []
    for i in range(n):
        hs = [res[2 * i + p]["disp"] * np.float32(1.0 / (1 + p)) for p in range(2)]     # "heights": view p sees (1 + p) x the parallax
        fused.append(_lib.merge_n(hs, [0.0, 0.0], "average_if_close", threshold=3.0))
        assert np.isfinite(fused[-1]).mean() > 0.8
    pm = oracle.census_params(recursion=SHIM_RECURSION)
    i = 2
    oh = [_oracle_tile(oracle, views[i][0], views[i][1 + p], size, dmin, dmax, pm)["disp"] * np.float32(1.0 / (1 + p)) for p in range(2)]
    for p in range(2):
        assert same(res[2 * i + p]["disp"] * np.float32(1.0 / (1 + p)), oh[p])
    assert same(fused[i], oracle.oracle_merge_n(oh, [0.0, 0.0], "average_if_close", 3.0))


@pytest.mark.parametrize("size,nd,rec,dirs", [(1024, 128, 2, 8), (1000, 256, 2, 8), (1024, 128, 1, 8), (1024, 128, 2, 16), (1000, 256, 1, 16)])
def test_mgm_mode_equals_the_oracle_at_the_full_tile_shapes(oracle, size, nd, rec, dirs):
    """The drop-in's default aggregation (recursion = 1, the band-pipelined launch) against the CPU oracle at the tile shapes
    of configs[1] and configs[3] themselves (so far HIP vs HIP at this size)."""
    from s2p_amd import _lib
    amp = 0.3125 * nd
    im1, im2 = synth_pair(1000, size, size, lambda x, y: amp * np.sin(2 * np.pi * x / (size / 2.)) * np.cos(2 * np.pi * y / (size / 2.)))
    dmin, dmax = -nd // 2, nd // 2 - 1
    r = _lib.census_sgm(im1, im2, dmin, dmax, params=_lib.default_census_params(recursion=rec, nb_dir=dirs))     # (16: the knight's moves on top, round 4)
    o = oracle.oracle_census_sgm(im1, im2, dmin, dmax, params=oracle.census_params(recursion=rec, nb_dir=dirs))
    assert same(r["disp"], o["disp"]) and np.array_equal(r["mask"], o["mask"]) and same(r["conf"], o["conf"])
    assert np.isfinite(r["disp"]).mean() > 0.9


@pytest.mark.parametrize("size,nd,n", [(1000, 256, 4), (1024, 128, 8)])
def test_job_batch_shape_equals_the_oracle(oracle, size, nd, n):
    """The call shapes the bench times: 4 tiles of 1000 x 1000 x 256 per library call -- the `job`'s
    s2p_hip_tile_host_batch shape, where tiles from 768 px run the 16-disparities-per-lane (K = 8) lattice layout under a
    batched queue -- and 8 tiles of 1024 x 1024 x 128 per call, the headline's own call shape.  One tile of the batch against
    the CPU oracle bit for bit (resampler + matcher + mask), every tile against its own single call."""
    from s2p_amd import _lib
    from s2p_amd.block_matching import matcher_params
    dmin, dmax = -nd // 2, nd // 2 - 1
    kind, params = matcher_params("mgm")
    assert kind == "census" and params.recursion == SHIM_RECURSION
    views = [tile_views(1000 * (k // 4) + (k % 4) * 7 + 2, size + 2 * PAD, nd, 2) for k in range(n)]
    kw = [dict(src1=v[0], H1=HS, src2=v[1], H2=HS, w=size, h=size, dmin=dmin, dmax=dmax, params=params, want_rect=False) for v in views]
    got = _lib.tile_batch(kw)
    assert len(got) == n
    for k in range(n):
        one = _lib.tile(algo="census", **kw[k])
        assert same(got[k]["disp"], one["disp"]) and np.array_equal(got[k]["mask"], one["mask"]), "tile %d of the batch != its single call" % k
        assert np.isfinite(got[k]["disp"]).mean() > 0.9
    k = n - 2
    o = _oracle_tile(oracle, views[k][0], views[k][1], size, dmin, dmax, oracle.census_params(recursion=SHIM_RECURSION))
    assert same(got[k]["disp"], o["disp"]) and np.array_equal(got[k]["mask"], o["mask"])
