"""s2p_amd/tiles.py -- tile-level data parallelism of the hot path over the GPUs of one node.

The reference's only parallelism is tiles x pairs handed to a multiprocessing.Pool
(s2p/__init__.py:561-562,578-591; s2p/parallel.py:58-110); tiles share nothing while they are
rectified and matched.  Here: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI
on GPUs, "gloo" in the CPU tests), a static round-robin shard of the tile list per rank or a shared work
queue (WorkQueue: tile cost varies with the disparity range and the valid area), several tiles
in flight per GPU on separate HIP streams (one libs2p_hip context per worker thread: the C calls release
the GIL), and NO collective on the data path.  The only exchange is the final gather of the per-rank
result tiles into a mosaic on one rank -- the counterpart of the reference's file-based merge
(s2p/__init__.py:509-525, utils/s2p_mosaic.py).
"""
import queue
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np


def shard(items, rank, world_size):
    """Static round-robin ownership: item i belongs to rank i % world_size."""
    return list(items[rank::world_size])


def owner(index, world_size):
    return index % world_size


class WorkQueue:
    """Dynamic ownership: a shared counter hands out tile indices `chunk` at a time to whoever asks next -- ranks (one
    process per GPU) and the worker threads inside a rank alike -- so a rank whose tiles turn out cheap (small disparity
    range, NaN-heavy border tiles) takes more of them.  Across processes the counter lives in torch.distributed's
    key-value store (`add` is atomic; the reference's counterpart is multiprocessing.Pool's shared task queue,
    s2p/parallel.py:76-98); without an initialised process group it is a local counter.  Every rank must create its
    queues in the same order (the key is a sequence number)."""
    _seq = 0

    def __init__(self, n_items, chunk=1, store=None):
        import torch.distributed as dist
        self.n, self.chunk = int(n_items), max(1, int(chunk))
        WorkQueue._seq += 1
        self.key = "s2p_amd_workqueue_%d" % WorkQueue._seq
        self.store = store
        if store is None and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            from torch.distributed import distributed_c10d
            self.store = distributed_c10d._get_default_store()
        self._local = 0
        self._lock = threading.Lock()
        self.seen = 0                                         # highest counter value this process has seen (a lower bound of what is taken)
        self.parties = 1
        if dist.is_available() and dist.is_initialized():
            self.parties = dist.get_world_size()

    def next(self, k=None):
        """The next indices to work on ([] when the list is exhausted); k: how many (default: the queue's chunk)."""
        k = self.chunk if k is None else max(1, int(k))
        if self.store is not None:
            end = int(self.store.add(self.key, k))
        else:
            with self._lock:
                self._local += k
                end = self._local
        self.seen = max(self.seen, min(end, self.n))
        return list(range(min(end - k, self.n), min(end, self.n)))

    def guided(self, most, workers):
        """How many indices a worker that would like `most` should take now so that the job ends evenly: at most half of an equal
        share of what is (as far as this process knows) still unclaimed, split over every rank's `workers` (guided self-scheduling).
        With big requests all the way a rank could end one whole batch after the others: 4 tiles of a 50-tile share."""
        left = self.n - self.seen
        return max(1, min(int(most), left // (2 * max(1, self.parties * workers))))

    def __iter__(self):
        while True:
            got = self.next()
            if not got:
                return
            yield from got


class Tile:
    """One unit of work: a rectified pair (arrays) + its disparity range + where its result goes in
    the mosaic (y0, x0).  `index` is its position in the global tile list (defines ownership)."""

    def __init__(self, index, im1, im2, disp_min, disp_max, y0=0, x0=0):
        self.index, self.im1, self.im2 = index, im1, im2
        self.disp_min, self.disp_max, self.y0, self.x0 = disp_min, disp_max, y0, x0


_pools = {}
_pools_lock = threading.Lock()


def _context_pool(device, n):
    """n libs2p_hip contexts (= HIP streams + workspaces) on `device`, created once per process and reused by
    every later call: a context owns a workspace of the size of the largest tile it has seen."""
    import ctypes
    from s2p_amd import _lib
    with _pools_lock:
        have = _pools.setdefault(device, [])
        while len(have) < n:
            p = ctypes.c_void_p()
            _lib.check(_lib.lib().s2p_hip_ctx_create(device, None, ctypes.byref(p)))
            have.append(p)
        q = queue.Queue()
        for c in have[:n]:
            q.put(c)
    return q


def _hip_matcher(algo, device, in_flight, config=None):
    """Matcher running each tile on a context borrowed from the per-device pool (one HIP stream per in-flight tile),
    with the parameters `algo` + cfg give the file-level compute_disparity_map (block_matching.matcher_params)."""
    from s2p_amd import _lib
    from s2p_amd.block_matching import matcher_params, params_for_range
    kind, params = matcher_params(algo, config)
    pool = _context_pool(device, max(in_flight, 1))

    def run(tile):
        ctx = pool.get()
        try:
            if kind == "sgbm":
                return _lib.sgbm(tile.im1, tile.im2, tile.disp_min, tile.disp_max, params=params, want_cost=False, device=device, ctx=ctx)["disp"]
            return _lib.census_sgm(tile.im1, tile.im2, tile.disp_min, tile.disp_max, params=params_for_range(kind, params, tile.disp_min, tile.disp_max),
                                   want_conf=False, device=device, ctx=ctx)["disp"]
        finally:
            pool.put(ctx)
    return run


class TileJob:
    """One tile for the whole path of steps 3-5 (rectify -> match -> mask -> triangulate): the two source
    windows with their window-to-rectified homographies, the rectified size, the disparity range and, when the
    3-D points are wanted, the triangulation inputs (dict as s2p_amd._lib.tile: rpca, rpcb, ha, hb, msk_orig,
    bbox).  `index` = position in the global tile list (defines ownership)."""

    def __init__(self, index, src1, H1, src2, H2, w, h, disp_min, disp_max, erosion=0, tri=None):
        self.index, self.src1, self.H1, self.src2, self.H2 = index, src1, H1, src2, H2
        self.w, self.h, self.disp_min, self.disp_max, self.erosion, self.tri = w, h, disp_min, disp_max, erosion, tri


def _hip_pipeline(algo, device, in_flight, want_rect=False, sink=None, config=None, pinned=True):
    """TileJob -> result dict through ONE library call per tile (s2p_hip_tile_host) on a context (= HIP stream)
    borrowed from the per-device pool: nothing of a tile touches the host between rectification and triangulation.
    The matcher runs with the parameters `algo` + cfg give the file-level compute_disparity_map.  `pinned`: the result arrays
    are page-locked (s2p_amd._lib.pinned_empty: their blocks are recycled as results are dropped), so the download of one tile
    overlaps the kernels of the next; callers that want the same for the uploads hand over source windows made with
    _lib.pinned_copy / read with io.read_image(alloc=_lib.pinned_empty)."""
    from s2p_amd import _lib
    from s2p_amd.block_matching import matcher_params, params_for_range
    kind, params = matcher_params(algo, config)
    pool = _context_pool(device, max(in_flight, 1))

    recycle = {}                                              # context -> result buffers of its previous tile(s)

    def run_many(group):
        """Several TileJobs of one shape and range (same_shape) through ONE library call (s2p_hip_tile_host_batch): one batched
        matcher launch for the group.  Returns the list of result dicts (of Nones with a sink)."""
        if kind == "sgbm" or len(group) == 1:
            return [run(j) for j in group]
        ctx = pool.get()
        try:
            old = recycle.get(ctx.value) if sink else None
            old = old if isinstance(old, list) else [old]
            j0 = group[0]
            p = params_for_range(kind, params, j0.disp_min, j0.disp_max)
            res = _lib.tile_batch([dict(src1=j.src1, H1=j.H1, src2=j.src2, H2=j.H2, w=j.w, h=j.h, dmin=j.disp_min, dmax=j.disp_max, params=p,
                                        erosion=j.erosion, tri=j.tri, want_rect=want_rect, out=old[k] if k < len(old) else None, pinned=pinned)
                                   for k, j in enumerate(group)], device=device, ctx=ctx)
            if sink is None:
                return res
            for j, r in zip(group, res):
                sink(j, r)
            recycle[ctx.value] = res
            return [None] * len(group)
        finally:
            pool.put(ctx)

    def run(job):
        ctx = pool.get()
        try:
            res = _lib.tile(job.src1, job.H1, job.src2, job.H2, job.w, job.h, job.disp_min, job.disp_max,
                            algo=kind, params=params_for_range(kind, params, job.disp_min, job.disp_max), erosion=job.erosion, tri=job.tri,
                            want_rect=want_rect, device=device, ctx=ctx, out=_first(recycle.get(ctx.value)) if sink else None, pinned=pinned)
            if sink is None:
                return res
            sink(job, res)                                    # the consumer is done with the arrays when it returns
            recycle[ctx.value] = res
            return None
        finally:
            pool.put(ctx)
    run.many = run_many
    # tiles of different sizes / ranges share a call (s2p_hip_tile_host_batch -> census_batch_hetero_enqueue) in the single-scale MGM modes
    # with P2 <= 115 when the common depth wastes at most a quarter on either of them; otherwise only equal shapes do
    hetero = kind == "census" and params.recursion >= 1 and params.scales <= 1 and params.P2 <= 115 and params.subpix != 2

    from .broker import census_depth

    def compatible(a, b):
        if same_shape(a, b):
            return True
        if not hetero:
            return False
        da, db = census_depth(a.disp_min, a.disp_max), census_depth(b.disp_min, b.disp_max)
        return min(da, db) * 4 >= max(da, db) * 3
    run.compatible = compatible
    return run


def _first(r):
    return r[0] if isinstance(r, list) else r


def same_shape(a, b):
    """Two TileJobs that one batched call can take together: same rectified size and disparity range."""
    return (a.w, a.h, a.disp_min, a.disp_max) == (b.w, b.h, b.disp_min, b.disp_max)


def _groups(jobs, batch, compatible=None):
    """Runs of consecutive jobs one library call can take together (same_shape, or the runner's `compatible`), at most `batch` long."""
    compatible = compatible or same_shape
    out = []
    for j in jobs:
        if out and len(out[-1]) < batch and all(compatible(g, j) for g in out[-1]):
            out[-1].append(j)
        else:
            out.append([j])
    return out


def process_tiles(jobs, algo="mgm", device=None, in_flight=2, runner=None, want_rect=False, sink=None, config=None):
    """Steps 3-5 of the reference for this rank's TileJobs, `in_flight` tiles at a time on separate HIP
    streams: the GPU-side replacement of the three Pool passes over the tile list
    (s2p/__init__.py:578-591 through s2p/parallel.py:58-110).  Returns {job.index: dict(disp, mask[,
    lonlatalt, err, rect1, rect2])}.  `runner` can be injected for CPU tests of the scheduling logic.
    `sink(job, result)`: stream the results to a consumer instead of collecting them -- it is called from the
    worker thread, the result arrays are recycled for the next tile of that worker once it returns, and the
    returned dict maps every index to None.  `config`: a cfg-like dict (default: s2p_amd.config.cfg) -- the same keys
    the file-level shim reads, so a tile gives the same disparities through either door."""
    if runner is None:
        from s2p_amd import _lib
        if device is None:
            device = _lib.default_device()
        runner = _hip_pipeline(algo, device, in_flight, want_rect, sink, config)
    if in_flight <= 1:
        return {j.index: runner(j) for j in jobs}
    with ThreadPoolExecutor(max_workers=in_flight) as ex:
        return dict(zip([j.index for j in jobs], ex.map(runner, jobs)))


def process_queue(jobs, queue, algo="mgm", device=None, in_flight=2, runner=None, want_rect=False, sink=None, config=None, batch=1):
    """process_tiles with dynamic ownership: `jobs` is the GLOBAL list (same on every rank), `queue` a WorkQueue over
    it; each of this rank's `in_flight` workers pulls the next index when it is free.  Returns {index: result} for the
    tiles this rank ended up processing.
    `batch` > 1: a worker pulls `batch` indices (one WorkQueue chunk of that size, or several smaller ones) and hands the consecutive same-shape tiles among them
    to ONE library call, up to `batch` at a time (s2p_hip_tile_host_batch: one batched matcher launch; byte-identical
    results) -- the way to fill the chip with the MGM matcher, whose single-tile launch follows the tile's dependency
    chain.  Tiles of other shapes (border tiles) simply form their own, smaller groups."""
    if runner is None:
        from s2p_amd import _lib
        if device is None:
            device = _lib.default_device()
        runner = _hip_pipeline(algo, device, in_flight, want_rect, sink, config)
    out, lock = {}, threading.Lock()

    many = getattr(runner, "many", None) or (lambda group: [runner(j) for j in group])

    def worker():
        if batch <= 1:
            for i in queue:
                r = runner(jobs[i])
                with lock:
                    out[jobs[i].index] = r
            return
        guided = getattr(queue, "guided", None)
        while True:
            want = guided(batch, in_flight) if guided else batch          # smaller requests towards the end of the list
            got = queue.next(want) if guided else queue.next()
            while got and len(got) < want:                    # (a queue that hands out fewer at a time: ask again)
                more = queue.next()
                if not more:
                    break
                got += more
            if not got:
                return
            for group in _groups([jobs[i] for i in got], min(batch, 16) if getattr(runner, "compatible", None) else batch, getattr(runner, "compatible", None)):
                rs = many(group)
                with lock:
                    for j, r in zip(group, rs):
                        out[j.index] = r
    if in_flight <= 1:
        worker()
        return out
    with ThreadPoolExecutor(max_workers=in_flight) as ex:
        for f in [ex.submit(worker) for _ in range(in_flight)]:
            f.result()
    return out


def match_tiles(tiles, algo="mgm", device=None, in_flight=2, matcher=None, config=None):
    """Run the matcher on this rank's tiles, `in_flight` at a time.  Returns {tile.index: disparity}.
    `matcher` (tile -> array) can be injected (CPU tests of the scheduling logic); by default the HIP
    path is used -- there is no CPU fallback."""
    if matcher is None:
        from s2p_amd import _lib
        if device is None:
            device = _lib.default_device()
        matcher = _hip_matcher(algo, device, in_flight, config)
    if in_flight <= 1:
        return {t.index: matcher(t) for t in tiles}
    with ThreadPoolExecutor(max_workers=in_flight) as ex:
        return dict(zip([t.index for t in tiles], ex.map(matcher, tiles)))


def _tiles_disjoint(layout):
    """True when no two rectangles (y0, x0, h, w) of `layout` overlap.  A sweep over y: rectangles sorted by their top edge, those
    whose bottom edge lies above the current top retired, the x-intervals of the live ones kept sorted -- O(N log N) for the grids a
    tiling produces, and it stops at the first overlap, which is where a reference tiling (tiles carry margins) ends at once.
    (Until round 4 an N x N matrix of int64 temporaries: several GB on the destination rank at 10^4 tiles)"""
    import bisect
    import heapq
    order = sorted(range(len(layout)), key=lambda i: (layout[i][0], layout[i][1]))
    live_x = []                                                 # sorted (x0, x1) of the rectangles the sweep line crosses
    ends = []                                                   # heap of (y1, x0, x1)
    for i in order:
        y0, x0, h, w = (int(v) for v in layout[i])
        if h <= 0 or w <= 0:
            continue
        while ends and ends[0][0] <= y0:
            _, a, b = heapq.heappop(ends)
            live_x.pop(bisect.bisect_left(live_x, (a, b)))
        k = bisect.bisect_left(live_x, (x0, x0 + w))
        if (k < len(live_x) and live_x[k][0] < x0 + w) or (k > 0 and live_x[k - 1][1] > x0):
            return False
        live_x.insert(k, (x0, x0 + w))
        heapq.heappush(ends, (y0 + h, x0, x0 + w))
    return True


def gather_mosaic(local_results, layout, shape, dst=0, group=None, device="cpu", dynamic=False, out=None, collectives=None):
    """Gather the per-rank tiles into one float32 mosaic on rank `dst` (None elsewhere).
    dynamic=True: ownership is whatever `local_results` holds on each rank (WorkQueue scheduling) -- one extra tiny
    all-reduce tells every rank who has what; otherwise the static round-robin of `shard`.

    local_results: {index: 2-D float32 array} for the tiles this rank owns
    layout: list over ALL tiles of (y0, x0, h, w), index = position in the list (same on every rank)
    shape: (H, W) of the mosaic.  Pixels no tile covers are NaN; later tiles overwrite earlier ones
    where they overlap (the reference's margins make tiles overlap).
    One collective: a padded `gather` of each rank's concatenated tiles.
    out: a float32 array of `shape` to assemble into (a caller that produces one mosaic per pair reuses it: a fresh 100 MB
    array costs more in first-touch page faults than the whole assembly).
    collectives: None = only when there is somebody to talk to (world > 1); True = also in a group of ONE rank (the owner count and the
    gather then run through the backend all the same: how tests/test_gpu_rccl.py puts this function's RCCL calls on a 1-GPU box)."""
    import torch
    import torch.distributed as dist
    if str(device) != "cpu" and not torch.cuda.is_available():
        raise RuntimeError("torch sees no GPU: import torch and touch torch.cuda BEFORE the first s2p_amd call "
                           "(the wheel bundles its own HIP runtime; see INTEGRATION.md)")
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    talk = world > 1 or bool(collectives and dist.is_initialized())
    if dynamic and talk:
        who = torch.zeros((2, len(layout)), dtype=torch.int64)  # row 0: sum of (owner + 1), row 1: number of owners
        for i in local_results:
            who[0, i] = rank + 1
            who[1, i] = 1
        who = who.to(device)
        dist.all_reduce(who, group=group)
        who = who.cpu()
        # every tile has exactly one owner (two owners could otherwise sum to a valid third: ranks 0 and 1 give 3 = rank 2)
        if not bool((who[1] == 1).all()):
            bad = [int(i) for i in torch.nonzero(who[1] != 1).flatten().tolist()[:8]]
            raise RuntimeError("gather_mosaic: tiles %s were processed by no rank or by several" % bad)
        who = [int(v) - 1 for v in who[0].tolist()]
    elif dynamic:
        who = [0] * len(layout)
    else:
        who = [owner(i, world) for i in range(len(layout))]
    sizes = [0] * world
    starts = [0] * len(layout)                                  # offset of tile i inside its owner's packed buffer
    for i, (_, _, h, w) in enumerate(layout):
        starts[i] = sizes[who[i]]
        sizes[who[i]] += h * w

    def for_tiles(fn, idx):
        """fn(i) for every tile index: the copies are memory-bound and numpy releases the GIL for them, so a few threads
        move a 100 MB mosaic several times faster than one (a 400-tile Python loop costs 230 ms, this 25-40)."""
        if len(idx) < 16:
            for i in idx:
                fn(i)
            return
        nth = min(8, max(1, len(idx) // 8))
        with ThreadPoolExecutor(max_workers=nth) as ex:
            list(ex.map(lambda chunk: [fn(i) for i in chunk], [idx[k::nth] for k in range(nth)]))

    mine = [i for i in range(len(layout)) if who[i] == rank]
    for i in mine:
        a = local_results[i]
        assert a.shape == tuple(layout[i][2:]), "tile %d: got %s, layout says %s" % (i, a.shape, tuple(layout[i][2:]))
    if talk:
        cap = max(max(sizes), 1)
        buf = torch.empty((cap,), dtype=torch.float32)           # the padding beyond a rank's tiles is never read
        bnp = buf.numpy()

        def pack(i):
            _, _, h, w = layout[i]
            bnp[starts[i]:starts[i] + h * w].reshape(h, w)[...] = local_results[i]
        for_tiles(pack, mine)
        buf = buf.to(device)
        recv = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
        dist.gather(buf, recv, dst=dst, group=group)
        if rank != dst:
            return None
        parts = [o.cpu().numpy() for o in recv]
    elif rank != dst:
        return None
    H, W = shape
    # filled by ONE thread first: it faults the fresh pages in (8 threads faulting one new mapping at once take 10 x longer than
    # the copies themselves) and leaves NaN where no tile lands
    if out is not None:
        assert out.shape == tuple(shape) and out.dtype == np.float32
        mosaic = out
        mosaic.fill(np.nan)
    else:
        mosaic = np.full(shape, np.nan, np.float32)
    disjoint = _tiles_disjoint(layout)                           # no two tiles overlap: any order gives the same mosaic

    def place(i):
        y0, x0, h, w = layout[i]
        src = local_results[i] if not talk else parts[who[i]][starts[i]:starts[i] + h * w].reshape(h, w)
        mosaic[y0:y0 + h, x0:x0 + w] = src
    if disjoint:
        for_tiles(place, list(range(len(layout))))
    else:                                                        # overlapping tiles: later ones overwrite earlier ones, in order
        for i in range(len(layout)):
            place(i)
    return mosaic
