"""s2p_amd/tiles.py -- tile-level data parallelism of the hot path over the GPUs of one node.

The reference's only parallelism is tiles x pairs handed to a multiprocessing.Pool
(s2p/__init__.py:561-562,578-591; s2p/parallel.py:58-110); tiles share nothing while they are
rectified and matched.  Here: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI
on GPUs, "gloo" in the CPU tests), a static round-robin shard of the tile list per rank, several tiles
in flight per GPU on separate HIP streams (one libs2p_hip context per worker thread: the C calls release
the GIL), and NO collective on the data path.  The only exchange is the final gather of the per-rank
result tiles into a mosaic on one rank -- the counterpart of the reference's file-based merge
(s2p/__init__.py:509-525, utils/s2p_mosaic.py).
"""
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np


def shard(items, rank, world_size):
    """Static round-robin ownership: item i belongs to rank i % world_size."""
    return list(items[rank::world_size])


def owner(index, world_size):
    return index % world_size


class Tile:
    """One unit of work: a rectified pair (arrays) + its disparity range + where its result goes in
    the mosaic (y0, x0).  `index` is its position in the global tile list (defines ownership)."""

    def __init__(self, index, im1, im2, disp_min, disp_max, y0=0, x0=0):
        self.index, self.im1, self.im2 = index, im1, im2
        self.disp_min, self.disp_max, self.y0, self.x0 = disp_min, disp_max, y0, x0


def _hip_matcher(algo, device):
    """Matcher bound to one libs2p_hip context per calling thread (= one HIP stream per in-flight tile)."""
    from s2p_amd import _lib
    local = threading.local()

    def run(tile):
        if not hasattr(local, "ctx"):
            import ctypes
            p = ctypes.c_void_p()
            _lib.check(_lib.lib().s2p_hip_ctx_create(device, None, ctypes.byref(p)))
            local.ctx = p
        if algo == "sgbm":
            return _lib.sgbm(tile.im1, tile.im2, tile.disp_min, tile.disp_max, want_cost=False, device=device, ctx=local.ctx)["disp"]
        return _lib.census_sgm(tile.im1, tile.im2, tile.disp_min, tile.disp_max, want_conf=False, device=device, ctx=local.ctx)["disp"]
    return run


def match_tiles(tiles, algo="mgm", device=None, in_flight=2, matcher=None):
    """Run the matcher on this rank's tiles, `in_flight` at a time.  Returns {tile.index: disparity}.
    `matcher` (tile -> array) can be injected (CPU tests of the scheduling logic); by default the HIP
    path is used -- there is no CPU fallback."""
    if matcher is None:
        from s2p_amd import _lib
        if device is None:
            device = _lib.default_device()
        matcher = _hip_matcher(algo, device)
    if in_flight <= 1:
        return {t.index: matcher(t) for t in tiles}
    with ThreadPoolExecutor(max_workers=in_flight) as ex:
        return dict(zip([t.index for t in tiles], ex.map(matcher, tiles)))


def gather_mosaic(local_results, layout, shape, dst=0, group=None, device="cpu"):
    """Gather the per-rank tiles into one float32 mosaic on rank `dst` (None elsewhere).

    local_results: {index: 2-D float32 array} for the tiles this rank owns
    layout: list over ALL tiles of (y0, x0, h, w), index = position in the list (same on every rank)
    shape: (H, W) of the mosaic.  Pixels no tile covers are NaN; later tiles overwrite earlier ones
    where they overlap (the reference's margins make tiles overlap).
    One collective: a padded `gather` of each rank's concatenated tiles."""
    import torch
    import torch.distributed as dist
    if str(device) != "cpu" and not torch.cuda.is_available():
        raise RuntimeError("torch sees no GPU: import torch and touch torch.cuda BEFORE the first s2p_amd call "
                           "(the wheel bundles its own HIP runtime; see INTEGRATION.md)")
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    sizes = [0] * world
    for i, (_, _, h, w) in enumerate(layout):
        sizes[owner(i, world)] += h * w
    cap = max(max(sizes), 1)
    buf = torch.full((cap,), float("nan"), dtype=torch.float32)
    off = 0
    for i, (_, _, h, w) in enumerate(layout):
        if owner(i, world) != rank:
            continue
        a = np.ascontiguousarray(local_results[i], np.float32)
        assert a.shape == (h, w), "tile %d: got %s, layout says %s" % (i, a.shape, (h, w))
        buf[off:off + h * w] = torch.from_numpy(a.reshape(-1))
        off += h * w
    buf = buf.to(device)
    if world > 1:
        out = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
        dist.gather(buf, out, dst=dst, group=group)
    else:
        out = [buf]
    if rank != dst:
        return None
    parts = [o.cpu().numpy() for o in out]
    mosaic = np.full(shape, np.nan, np.float32)
    offs = [0] * world
    for i, (y0, x0, h, w) in enumerate(layout):
        r = owner(i, world)
        mosaic[y0:y0 + h, x0:x0 + w] = parts[r][offs[r]:offs[r] + h * w].reshape(h, w)
        offs[r] += h * w
    return mosaic
