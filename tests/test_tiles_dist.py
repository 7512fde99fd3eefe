"""Multi-process CPU tests (gloo, world_size 2) of the tile-parallel path: ownership, in-flight
scheduling and the one collective (mosaic gather).  The matcher is injected -- the HIP path has no
CPU fallback -- so this covers the host logic that runs identically under RCCL on GPUs."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def fake_matcher(tile):
    """Deterministic stand-in: a function of the tile's content and range (NaN where im1 < 0)."""
    d = (tile.im1 * 0.5 + tile.im2 * 0.25 + tile.disp_min).astype(np.float32)
    d[tile.im1 < 0] = np.nan
    return d


def make_tiles():
    from s2p_amd.tiles import Tile
    rng = np.random.default_rng(0)
    tiles, layout = [], []
    sizes = [(32, 48), (32, 40), (24, 48), (24, 40), (16, 20), (30, 30), (8, 64)]
    pos = [(0, 0), (0, 40), (28, 0), (28, 40), (50, 10), (50, 40), (80, 0)]       # overlapping margins
    for i, ((h, w), (y0, x0)) in enumerate(zip(sizes, pos)):
        a = rng.uniform(-1, 10, (h, w)).astype(np.float32)
        b = rng.uniform(0, 10, (h, w)).astype(np.float32)
        tiles.append(Tile(i, a, b, -3 - i, 5 + i, y0, x0))
        layout.append((y0, x0, h, w))
    return tiles, layout, (96, 96)


def serial_mosaic():
    tiles, layout, shape = make_tiles()
    m = np.full(shape, np.nan, np.float32)
    for t, (y0, x0, h, w) in zip(tiles, layout):
        m[y0:y0 + h, x0:x0 + w] = fake_matcher(t)
    return m


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from s2p_amd import tiles as T
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tiles, layout, shape = make_tiles()
        mine = T.shard(tiles, rank, world)
        assert [t.index for t in mine] == list(range(rank, len(tiles), world))
        res = T.match_tiles(mine, in_flight=2, matcher=fake_matcher)
        assert sorted(res) == [t.index for t in mine]
        mosaic = T.gather_mosaic(res, layout, shape, dst=0)
        # the same list through the shared work queue: whoever is free takes the next tile; every tile exactly once
        import time

        class Job:
            def __init__(self, t):
                self.index, self.t = t.index, t
                self.h, self.w, self.disp_min, self.disp_max = t.im1.shape[0], t.im1.shape[1], 0, 7

        def slow_runner(job):                          # rank 0 is 5x slower per tile: the queue gives it fewer
            time.sleep(0.01 if rank == 0 else 0.002)
            return fake_matcher(job.t)
        wq = T.WorkQueue(len(tiles))
        dyn = T.process_queue([Job(t) for t in tiles], wq, in_flight=2, runner=slow_runner)
        counts = [None] * world
        dist.all_gather_object(counts, sorted(dyn))
        assert sorted(i for c in counts for i in c) == list(range(len(tiles))), counts
        mosaic_dyn = T.gather_mosaic(dyn, layout, shape, dst=0, dynamic=True)
        # ... and in batches: chunks of 2 from the shared counter, handed to runner.many; every tile exactly once over the ranks
        groups = []

        def many(group):
            groups.append([j.index for j in group])
            return [slow_runner(j) for j in group]
        slow_runner.many = many
        bat = T.process_queue([Job(t) for t in tiles], T.WorkQueue(len(tiles), chunk=2), in_flight=2, runner=slow_runner, batch=2)
        counts_b = [None] * world
        dist.all_gather_object(counts_b, sorted(bat))
        assert sorted(i for c in counts_b for i in c) == list(range(len(tiles))), counts_b
        assert sorted(i for g in groups for i in g) == sorted(bat) and all(len(g) <= 2 for g in groups)
        for i in bat:
            assert np.array_equal(bat[i], fake_matcher(tiles[i]), equal_nan=True)
        # a tile held by two ranks must be refused, also when the two owners sum to a valid third (ranks 0 and 1 -> "rank 2")
        if world >= 2:
            twice = dict(dyn)
            dup = next(i for c in counts[1:2] for i in c)          # a tile rank 1 owns
            if rank == 0:
                twice[dup] = fake_matcher(tiles[dup])
            refused = False
            try:
                T.gather_mosaic(twice, layout, shape, dst=0, dynamic=True)
            except RuntimeError as e:
                refused = "several" in str(e)
            assert refused
        if rank == 0:
            assert np.array_equal(mosaic, mosaic_dyn, equal_nan=True)
            q.put(mosaic)
        else:
            assert mosaic is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_tiles_and_mosaic_gather_gloo(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    mosaic = q.get(timeout=900)          # a cold container pages torch in for a minute or two
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    assert np.array_equal(mosaic, serial_mosaic(), equal_nan=True)


def test_single_process_paths():
    from s2p_amd import tiles as T
    tiles, layout, shape = make_tiles()
    assert T.shard(list(range(10)), 1, 4) == [1, 5, 9]
    res = T.match_tiles(tiles, in_flight=1, matcher=fake_matcher)
    m = T.gather_mosaic(res, layout, shape)
    assert np.array_equal(m, serial_mosaic(), equal_nan=True)


def test_process_tiles_scheduling_logic():
    """process_tiles with an injected runner (no GPU): ownership by index, order-independent result map,
    in-flight threading returns the same thing as the serial loop."""
    from s2p_amd import tiles as T
    jobs = [T.TileJob(i, None, None, None, None, 8, 4, -3 - i, 5 + i, erosion=i % 2) for i in range(7)]

    def runner(job):
        return {"disp": np.full((job.h, job.w), job.index + 0.5 * job.erosion, np.float32), "range": (job.disp_min, job.disp_max)}
    mine = T.shard(jobs, 1, 3)
    assert [j.index for j in mine] == [1, 4]
    a = T.process_tiles(jobs, in_flight=1, runner=runner)
    b = T.process_tiles(jobs, in_flight=4, runner=runner)
    assert sorted(a) == sorted(b) == list(range(7))
    for i in a:
        assert np.array_equal(a[i]["disp"], b[i]["disp"]) and a[i]["range"] == b[i]["range"] == (-3 - i, 5 + i)


class PlainQueue:
    """A queue without guided(): hands out fixed chunks (what process_queue needs of a queue is next())."""

    def __init__(self, n, chunk):
        import threading
        self.n, self.chunk, self.at, self.lock = n, chunk, 0, threading.Lock()

    def next(self):
        with self.lock:
            a = self.at
            self.at = min(self.n, a + self.chunk)
            return list(range(a, self.at))

    def __iter__(self):
        while True:
            got = self.next()
            if not got:
                return
            yield from got


def test_process_queue_batches_group_consecutive_tiles_of_one_shape():
    """process_queue(batch = 3) with an injected runner: a worker hands what it pulled from the queue (chunks of 3) to
    runner.many in runs of consecutive same-shape tiles; other shapes form their own groups; every tile exactly once; a
    runner without `many` is called tile by tile; batch = 1 never calls `many`."""
    from s2p_amd import tiles as T
    shapes = [(8, 4), (8, 4), (8, 4), (8, 4), (6, 4), (8, 4), (8, 4), (8, 4), (8, 4), (8, 4), (6, 4)]
    jobs = [T.TileJob(i, None, None, None, None, w, h, -3, 5 + (i == 8)) for i, (w, h) in enumerate(shapes)]      # tile 8: another range
    assert [len(g) for g in T._groups(jobs, 3)] == [3, 1, 1, 3, 1, 1, 1] and [len(g) for g in T._groups(jobs, 8)] == [4, 1, 3, 1, 1, 1]
    calls = []

    def runner(job):
        calls.append([job.index])
        return job.index * 10

    def many(group):
        assert all(T.same_shape(group[0], j) for j in group) and len(group) <= 3
        calls.append([j.index for j in group])
        return [j.index * 10 for j in group]
    runner.many = many
    for in_flight in (1, 3):
        calls.clear()
        out = T.process_queue(jobs, PlainQueue(len(jobs), 3), in_flight=in_flight, runner=runner, batch=3)
        assert out == {i: 10 * i for i in range(len(jobs))}
        assert sorted(i for c in calls for i in c) == list(range(len(jobs)))
        assert sorted(calls) == [[0, 1, 2], [3], [4], [5], [6, 7], [8], [9], [10]]          # chunks [0-2] [3-5] [6-8] [9-10], split by shape
    calls.clear()
    del runner.many
    out = T.process_queue(jobs, PlainQueue(len(jobs), 3), in_flight=2, runner=runner, batch=3)
    assert out == {i: 10 * i for i in range(len(jobs))} and sorted(calls) == [[i] for i in range(len(jobs))]
    runner.many = many
    calls.clear()
    out = T.process_queue(jobs, PlainQueue(len(jobs), 1), in_flight=1, runner=runner, batch=3)       # chunk 1: the worker asks 3 times
    assert out == {i: 10 * i for i in range(len(jobs))} and sorted(calls) == [[0, 1, 2], [3], [4], [5], [6, 7], [8], [9], [10]]
    calls.clear()
    out = T.process_queue(jobs, T.WorkQueue(len(jobs)), in_flight=2, runner=runner)
    assert out == {i: 10 * i for i in range(len(jobs))} and sorted(calls) == [[i] for i in range(len(jobs))]


def test_work_queue_requests_shrink_towards_the_end():
    """Guided self-scheduling of the batched requests: full batches while the list is long, single tiles at the end, so that no
    worker ends a whole batch after the others; every tile exactly once whatever the request sizes."""
    from s2p_amd import tiles as T
    n = 100
    jobs = [T.TileJob(i, None, None, None, None, 8, 4, -3, 5) for i in range(n)]
    q = T.WorkQueue(n)
    sizes = []
    while True:
        want = q.guided(4, 3)
        got = q.next(want)
        if not got:
            break
        sizes.append(len(got))
    assert sum(sizes) == n and sizes[0] == 4 and sizes[-1] == 1 and sizes == sorted(sizes, reverse=True)
    assert sizes.count(4) >= 15 and max(sizes[-6:]) == 1          # the last 6 tiles (2 x 3 workers) go one at a time
    calls = []

    def runner(job):
        calls.append([job.index])
        return job.index

    def many(group):
        calls.append([j.index for j in group])
        return [j.index for j in group]
    runner.many = many
    out = T.process_queue(jobs, T.WorkQueue(n), in_flight=3, runner=runner, batch=4)
    assert out == {i: i for i in range(n)} and sorted(i for c in calls for i in c) == list(range(n))
    assert max(len(c) for c in calls) == 4 and sum(len(c) == 4 for c in calls) >= 15


def _worker8(rank, world, port, q):
    """The driver's 8-GPU job without 8 GPUs: 400 mock tiles through process_queue(batch = 4) on a gloo
    group of 8 ranks x 3 workers, one shared counter; per-tile cost jitters 3x around its mean and rank 5 is 2x slower."""
    sys.path.insert(0, ROOT)
    import time
    import torch.distributed as dist
    from s2p_amd import tiles as T
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, ts, cols, batch, in_flight = 400, 16, 20, 4, 3
        cost = 0.004 * (0.5 + np.random.default_rng(7).random(n) * 1.5)          # seconds per tile, the same table on every rank

        class Job:
            def __init__(self, i):
                self.index, self.w, self.h, self.disp_min, self.disp_max = i, ts, ts, 0, 7

        def runner(job):
            time.sleep(cost[job.index] * (2.0 if rank == 5 else 1.0))
            return np.full((ts, ts), job.index, np.float32)

        def many(group):                                   # one "library call" for the group: its tiles' costs add up
            return [runner(j) for j in group]
        runner.many = many
        jobs = [Job(i) for i in range(n)]
        dist.barrier()
        t0 = time.monotonic()
        mine = T.process_queue(jobs, T.WorkQueue(n, chunk=batch), in_flight=in_flight, runner=runner, batch=batch)
        t_done = time.monotonic() - t0
        owned = [None] * world
        dist.all_gather_object(owned, sorted(mine))
        times = [None] * world
        dist.all_gather_object(times, t_done)
        assert sorted(i for c in owned for i in c) == list(range(n)), "a tile was processed twice or not at all"
        layout = [((i // cols) * ts, (i % cols) * ts, ts, ts) for i in range(n)]
        mosaic = T.gather_mosaic(mine, layout, (n // cols * ts, cols * ts), dst=0, dynamic=True)
        if rank == 0:
            want = np.repeat(np.repeat(np.arange(n, dtype=np.float32).reshape(n // cols, cols), ts, 0), ts, 1)
            assert np.array_equal(mosaic, want)
            q.put((times, [len(c) for c in owned]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_eight_ranks_share_one_queue_and_end_together():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = 8
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    times, counts = q.get(timeout=900)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    assert sum(counts) == 400 and min(counts) > 0
    assert counts[5] < max(counts)                                 # the slow rank took fewer tiles
    # the guided chunks shrink towards the end of the list: the last rank ends within ONE batch of the first
    # (a batch of 4 tiles at the mean cost of 5 ms, twice that on the slow rank) -- not a whole extra round later
    one_batch = 4 * 0.005 * 2.0
    assert max(times) - min(times) <= one_batch + 0.02, (times, counts)


def test_disjointness_of_a_tiling_is_decided_by_a_sweep():
    """gather_mosaic places disjoint tiles in any order (threads) and overlapping ones in list order.  The decision is an O(N log N)
    sweep since round 5 (an N x N matrix before: GBs at 10^4 tiles): same answers as the matrix on random layouts, and a
    10^4-tile grid in milliseconds."""
    import time
    from s2p_amd.tiles import _tiles_disjoint
    rng = np.random.default_rng(0)

    def brute(layout):
        ys, xs, hs, ws = (np.array([t[k] for t in layout]) for k in range(4))
        ov = (np.minimum((ys + hs)[:, None], (ys + hs)[None, :]) > np.maximum(ys[:, None], ys[None, :])) & \
             (np.minimum((xs + ws)[:, None], (xs + ws)[None, :]) > np.maximum(xs[:, None], xs[None, :]))
        return int(ov.sum()) == len(layout)
    seen = set()
    for _ in range(2000):
        lay = [(int(rng.integers(0, 20)), int(rng.integers(0, 20)), int(rng.integers(1, 8)), int(rng.integers(1, 8))) for _ in range(rng.integers(1, 12))]
        assert _tiles_disjoint(lay) == brute(lay), lay
        seen.add(brute(lay))
    assert seen == {True, False}
    grid = [(y * 10, x * 10, 10, 10) for y in range(100) for x in range(100)]
    t = time.time()
    assert _tiles_disjoint(grid) and time.time() - t < 1.0
    grid[5000] = (grid[5000][0] - 1,) + grid[5000][1:]
    assert not _tiles_disjoint(grid)
    assert _tiles_disjoint([]) and _tiles_disjoint([(0, 0, 5, 5), (0, 5, 5, 5), (5, 0, 5, 10)])
