"""tests/e2e.py -- steps 3-7 of s2p.main on the reference's end-to-end inputs (fixtures of tests/golden/make_e2e.py),
with the per-tile work done either by the HIP library (the product: `hip`) or by the CPU oracle (`cpu`: the same
integer pipelines, bit for bit -- used where no GPU is present, and as the checker).

What is NOT the product here and is stated in numpy, because the reference does it in Python around the tile steps:
the tiling bookkeeping, the mean-height registration of the pairs (s2p/__init__.py:326-352), the choice of each
tile's raster window (s2p/__init__.py:445-460) and the first-wins merge of the tile rasters (rasterio.merge,
s2p/__init__.py:509-516).  compare_dsm restates tests/end2end_test.py:21-55 and returns the figures it asserts on.
"""
import numpy as np

from helpers import load_golden


def compare_dsm(computed, expected, absmean_tol, percentile_tol):
    """The reference's acceptance test (tests/end2end_test.py:21-55): same shape, valid count within 1 % (+100),
    |mean difference| <= absmean_tol, 99th percentile of |difference| <= percentile_tol.  Returns the measured
    figures and `ok`; the caller asserts."""
    r = dict(shape_ok=computed.shape == expected.shape)
    if not r["shape_ok"]:
        r.update(ok=False, shapes=(computed.shape, expected.shape))
        return r
    n_c, n_e = int(np.isfinite(computed).sum()), int(np.isfinite(expected).sum())
    diff = computed - expected
    diff = diff[np.isfinite(diff)]
    r.update(n_computed=n_c, n_expected=n_e, count_ok=abs(n_c - n_e) <= 100 + 0.01 * abs(n_e),
             mean=float(np.mean(diff)), p99=float(np.percentile(np.abs(diff), 99)), median_abs=float(np.median(np.abs(diff))))
    r["ok"] = bool(r["count_ok"] and abs(r["mean"]) <= absmean_tol and r["p99"] <= percentile_tol)
    return r


class Cpu:
    """The oracle as a backend (tests only)."""
    name = "cpu"

    def __init__(self, recursion=2):
        from oracle import pyoracle as po
        self.po = po
        self.params = po.census_params(recursion=recursion)           # the 'mgm' call site: median, 5x5 census, P1 8, P2 32

    def rpc(self, tag):
        return self.po.rpc_from_geotiff_tag(tag)

    def run_tiles(self, jobs):
        """jobs: dicts(src1, H1, src2, H2, w, h, dmin, dmax, rpc1, rpc2, ha, hb, bbx, mask_orig) -> [(disp, mask, lonlatalt)]"""
        po, out = self.po, []
        for j in jobs:
            r1, r2 = po.oracle_warp(j["src1"], j["H1"], j["w"], j["h"]), po.oracle_warp(j["src2"], j["H2"], j["w"], j["h"])
            r = po.oracle_census_sgm(r1, r2, j["dmin"], j["dmax"], params=self.params)
            lla = po.oracle_disp_to_lonlatalt(j["rpc1"], j["rpc2"], j["ha"], j["hb"], r["disp"], r["mask"], j["bbx"], j["mask_orig"])[0]
            out.append((r["disp"], r["mask"], lla))
        return out

    def cargarse_basura(self, hm):
        return self.po.oracle_cargarse_basura(hm)

    def localize(self, rpc, heights, off_x, off_y):
        return self.po.oracle_height_map_to_lonlatalt(rpc, heights, off_x, off_y)

    def filter_xyz(self, xyz, r, n, gsd):
        return self.po.oracle_remove_isolated_3d_points(xyz, r, int(np.ceil(r / gsd)), n)

    def plyflatten(self, cloud, xoff, yoff, res, xsize, ysize):
        return self.po.oracle_plyflatten(cloud, xoff, yoff, res, xsize, ysize)

    def height_transfer(self, heights, H, w, h):
        return self.po.oracle_height_transfer(heights, H, w, h)

    def merge_n(self, maps, offsets, threshold):
        return self.po.oracle_merge_n(maps, offsets, "average_if_close", threshold)


class Hip:
    """The product: libs2p_hip.so through the host mirrors of s2p_amd."""
    name = "hip"

    def __init__(self, recursion=2, device=None, in_flight=3):
        from s2p_amd import _lib, triangulation
        from s2p_amd.config import cfg as base
        self._lib, self.tri, self.device, self.in_flight = _lib, triangulation, device, in_flight
        self.cfg = dict(base)
        self.cfg["hip_mgm_recursion"] = recursion

    def rpc(self, tag):
        return self.tri.rpc_from_geotiff_tag(tag)

    def run_tiles(self, jobs):
        """The tiles through tiles.process_queue: one s2p_hip_tile_host call per tile (rectify -> match -> mask ->
        triangulate with the tile resident in HBM), `in_flight` tiles at a time on separate HIP streams."""
        from s2p_amd import tiles
        tj = [tiles.TileJob(k, j["src1"], j["H1"], j["src2"], j["H2"], j["w"], j["h"], j["dmin"], j["dmax"],
                            tri=dict(rpca=j["rpc1"], rpcb=j["rpc2"], ha=j["ha"], hb=j["hb"], msk_orig=j["mask_orig"], bbox=j["bbx"]))
              for k, j in enumerate(jobs)]
        res = tiles.process_queue(tj, tiles.WorkQueue(len(tj)), algo="mgm", device=self.device, in_flight=self.in_flight,
                                  config=self.cfg)
        return [(res[k]["disp"], res[k]["mask"], res[k]["lonlatalt"]) for k in range(len(tj))]

    def cargarse_basura(self, hm):
        return self._lib.cargarse_basura(hm, device=self.device)

    def localize(self, rpc, heights, off_x, off_y):
        return self._lib.height_map_to_lonlatalt(rpc, heights, off_x, off_y, device=self.device)

    def filter_xyz(self, xyz, r, n, gsd):
        xyz = np.ascontiguousarray(xyz, np.float64).copy()
        self.tri.filter_xyz(xyz, r, n, gsd, device=self.device)
        return xyz

    def plyflatten(self, cloud, xoff, yoff, res, xsize, ysize):
        return self._lib.plyflatten(cloud, xoff, yoff, res, xsize, ysize, device=self.device)

    def height_transfer(self, heights, H, w, h):
        return self._lib.height_transfer(heights, H, w, h, device=self.device)

    def merge_n(self, maps, offsets, threshold):
        return self._lib.merge_n(maps, offsets, "average_if_close", threshold, device=self.device)


def _int_range(fx, i, t):
    lo, hi = fx["disp_range_%d_%d" % (i, t)]
    return int(np.floor(lo)), int(np.ceil(hi))            # compute_disparity_map's rounding (s2p/block_matching.py:70-74)


def tile_cloud_roi(xyz, res):
    """The raster window plys_to_dsm gives a tile from its own cloud (s2p/__init__.py:445-460)."""
    p = xyz[np.isfinite(xyz).all(axis=1)]
    xmin, ymin = p[:, 0].min(), p[:, 1].min()
    xmax, ymax = p[:, 0].max(), p[:, 1].max()
    xoff = np.floor(xmin / res) * res
    xsize = int(1 + np.floor((xmax - xoff) / res))
    yoff = np.ceil(ymax / res) * res
    ysize = int(1 - np.floor((ymin - yoff) / res))
    return xoff, yoff, xsize, ysize


def merge_first(rasters, res):
    """rasterio.merge.merge(method='first') of north-up float rasters given as (xoff, yoff, array): union of the
    bounds, earlier rasters win, later ones fill what is still NaN (s2p/__init__.py:509-516)."""
    x0 = min(r[0] for r in rasters)
    y0 = max(r[1] for r in rasters)
    x1 = max(r[0] + res * r[2].shape[1] for r in rasters)
    y1 = min(r[1] - res * r[2].shape[0] for r in rasters)
    W, H = int(round((x1 - x0) / res)), int(round((y0 - y1) / res))
    out = np.full((H, W), np.nan, np.float32)
    for xoff, yoff, a in rasters:
        c, r_ = int(round((xoff - x0) / res)), int(round((y0 - yoff) / res))
        v = out[r_:r_ + a.shape[0], c:c + a.shape[1]]
        v[np.isnan(v)] = a[np.isnan(v)]
    return (x0, y0), out


def _jobs(fx, be, i, rpcs, pad):
    """One job per tile of pair i: windows, homographies, range, triangulation inputs (bbox padded by `pad` as
    triangulation.height_map does, s2p/triangulation.py:367-374)."""
    A, jobs = fx["A_%d" % i], []
    for t, (x, y, w, h) in enumerate(fx["tiles"].tolist()):
        H1, H2 = fx["H_ref_%d_%d" % (i, t)], fx["H_sec_%d_%d" % (i, t)]
        ww, hh = (int(v) for v in fx["size_%d_%d" % (i, t)])
        lo, hi = _int_range(fx, i, t)
        jobs.append(dict(src1=fx["img_0"], H1=H1, src2=fx["img_%d" % i], H2=H2, w=ww, h=hh, dmin=lo, dmax=hi,
                         rpc1=rpcs[0], rpc2=rpcs[i], ha=H1, hb=H2 @ np.linalg.inv(A),      # pointing correction (:113-114)
                         bbx=(x - pad, x + w + 2 * pad, y - pad, y + h + 2 * pad) if pad else (x, x + w, y, y + h),
                         mask_orig=np.ones((h + 2 * pad, w + 2 * pad), np.uint8)))
    return jobs


def rasterize(be, clouds, res, grid=None):
    """plys_to_dsm per tile over the neighbourhood clouds (every tile of a 2 x 2 tiling is everybody's neighbour) and
    global_dsm's first-wins merge; `grid` = (xoff, yoff, xsize, ysize) rasterises on a given grid instead."""
    allc = np.concatenate(clouds)
    if grid is not None:
        xoff, yoff, xs, ys = grid
        return (xoff, yoff), be.plyflatten(allc, xoff, yoff, res, xs, ys)[:, :, 0]
    rasters = []
    for c in clouds:
        xoff, yoff, xs, ys = tile_cloud_roi(c, res)
        rasters.append((xoff, yoff, be.plyflatten(allc, xoff, yoff, res, xs, ys)[:, :, 0]))
    return merge_first(rasters, res)


def run_pair(fx, be):
    """input_pair (s2p.main with two images): rectify -> match -> triangulate per tile; CRS; 3-D filter; cloud per tile;
    rasterise; merge.  Returns ((xoff, yoff), dsm, per-tile disparity maps)."""
    from s2p_amd import geographiclib
    res, crs = float(fx["dsm_resolution"]), str(fx["out_crs"])
    rpcs = [be.rpc(fx["rpc_0"]), be.rpc(fx["rpc_1"])]
    results = be.run_tiles(_jobs(fx, be, 1, rpcs, 0))
    clouds = []
    for disp, mask, lla in results:
        xyz = geographiclib.lonlatalt_to_utm(lla, crs)
        rr, nn = fx["filtering"]
        xyz = be.filter_xyz(xyz, float(rr), int(nn), float(fx["gsd"])).reshape(-1, 3)
        clouds.append(xyz[np.isfinite(xyz).all(axis=1)])
    origin, dsm = rasterize(be, clouds, res)
    return origin, dsm, [r[0] for r in results]


def run_triplet(fx, be, fusion_thresh=3.0, basura=True):
    """input_triplet (s2p.main with three images): per pair rectify -> match -> height map on the reference image's
    grid (triangulation.height_map); mean-height registration of the pairs; cargarse_basura + merge_n per tile
    (heights_fusion); localisation of the fused map, CRS, cloud per tile (heights_to_ply); rasterise on the grid of the
    expected DSM.  Returns dict(hm1: mosaic of pair 1's height maps as the files are left (after cargarse_basura),
    fused: mosaic of the fused maps, dsm: the raster)."""
    from s2p_amd import geographiclib
    tiles = fx["tiles"].tolist()
    rx, ry, rw, rh = (int(v) for v in fx["roi"])
    rpcs = [be.rpc(fx["rpc_%d" % k]) for k in range(3)]
    hm = {}
    for i in (1, 2):
        results = be.run_tiles(_jobs(fx, be, i, rpcs, 1))
        for t, ((x, y, w, h), (disp, mask, lla)) in enumerate(zip(tiles, results)):
            T = np.array([[1., 0., x], [0., 1., y], [0., 0., 1.]])
            hm[(i, t)] = be.height_transfer(lla[:, :, 2], fx["H_ref_%d_%d" % (i, t)] @ T, w, h).astype(np.float32)   # height_map.tif
    # mean_heights / global_mean_heights (s2p/__init__.py:326-352): on the maps as triangulated
    local = []
    for t in range(len(tiles)):
        maps = np.stack([hm[(i, t)] for i in (1, 2)], axis=2).astype(np.float64)
        valid = maps.sum(axis=2)
        valid = valid + 1 - valid
        with np.errstate(all="ignore"):
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                local.append([np.nanmean(valid * maps[:, :, k]) for k in range(2)])
    glob = np.nanmean(np.array(local), axis=0)
    # heights_fusion (s2p/__init__.py:355-385): cargarse_basura in place on each pair's map, then merge_n
    if basura:
        hm = {k: be.cargarse_basura(v) for k, v in hm.items()}
    fused = {t: be.merge_n([hm[(1, t)], hm[(2, t)]], [float(v) for v in glob], fusion_thresh) for t in range(len(tiles))}
    # heights_to_ply (:388-431): localise the fused map with the reference image's RPC, CRS, cloud (no 3-D filter in this config)
    clouds = []
    for t, (x, y, w, h) in enumerate(tiles):
        xyz = geographiclib.lonlatalt_to_utm(be.localize(rpcs[0], fused[t], x, y), str(fx["out_crs"])).reshape(-1, 3)
        clouds.append(xyz[np.isfinite(xyz).all(axis=1)])
    ys, xs = fx["dsm"].shape
    _, dsm = rasterize(be, clouds, float(fx["dsm_resolution"]), grid=(float(fx["dsm_origin"][0]), float(fx["dsm_origin"][1]), xs, ys))
    m1 = np.full((rh, rw), np.nan, np.float32)
    mf = np.full((rh, rw), np.nan, np.float32)
    for t, (x, y, w, h) in enumerate(tiles):
        m1[y - ry:y - ry + h, x - rx:x - rx + w] = hm[(1, t)]
        mf[y - ry:y - ry + h, x - rx:x - rx + w] = fused[t]
    return dict(hm1=m1, fused=mf, dsm=dsm, offsets=glob)


def load(name):
    return load_golden(name)


def _file_tile(args):
    """One tile x pair the way a worker of the reference's Pools handles it, file by file through the drop-in mirrors:
    rectification (s2p/__init__.py:101-159: image_apply_homography x 2), stereo_matching (:166-196: compute_disparity_map), the
    triangulation's library call (:199-239).  Runs in a forked Pool worker: every GPU call travels through the broker."""
    import os
    import sys
    d, k, j = args
    from s2p_amd import block_matching as bm, common, triangulation
    from s2p_amd import io as rio
    p = lambda n: os.path.join(d, "tile_%d_%s" % (k, n))
    so, sys.stdout = sys.stdout, open(os.devnull, "w")
    try:
        common.image_apply_homography(p("rectified_ref.tif"), j["im1"], j["H1"], j["w"], j["h"])
        common.image_apply_homography(p("rectified_sec.tif"), j["im2"], j["H2"], j["w"], j["h"])
        bm.compute_disparity_map(p("rectified_ref.tif"), p("rectified_sec.tif"), p("rectified_disp.tif"), p("rectified_mask.png"), "mgm",
                                 j["dmin"], j["dmax"], timeout=600)
    finally:
        sys.stdout.close()
        sys.stdout = so
    disp = rio.read_image(p("rectified_disp.tif"))
    mask = rio.read_image(p("rectified_mask.png"), np.uint8)
    lla, _ = triangulation.disp_to_lonlatalt(j["rpc1"], j["rpc2"], j["ha"], j["hb"], disp, mask, j["bbx"], j["mask_orig"])
    from s2p_amd import _lib
    return disp, mask, lla, len(_lib._ctx)


class FilePool(Hip):
    """The product as the reference's orchestrator drives it: a forked multiprocessing.Pool whose workers call the FILE-level mirrors
    (s2p/parallel.py:76-110); the tiles' GPU work goes through the device's broker.  The steps after the tiles run in this process."""
    name = "filepool"

    def __init__(self, workdir, workers=4, recursion=2):
        super().__init__(recursion=recursion)
        self.workdir, self.workers = workdir, workers
        self.worker_contexts = []

    def run_tiles(self, jobs):
        import multiprocessing as mp
        import os
        from s2p_amd import io as rio
        assert self.cfg["hip_mgm_recursion"] == 2            # the workers run the shim's own default
        files = {}
        args = []
        for k, j in enumerate(jobs):
            jj = dict(j)
            for key in ("src1", "src2"):
                a = j[key]
                if id(a) not in files:
                    files[id(a)] = os.path.join(self.workdir, "img_%d_%d.tif" % (len(self.worker_contexts), len(files)))
                    rio.write_image(files[id(a)], np.ascontiguousarray(a, np.float32))
                jj["im1" if key == "src1" else "im2"] = files[id(a)]
                del jj[key]
            jj["rpc1"], jj["rpc2"] = bytes(j["rpc1"]), bytes(j["rpc2"])      # (ctypes structs do not pickle)
            args.append((self.workdir, len(self.worker_contexts) + k, jj))
        with mp.get_context("fork").Pool(self.workers) as pool:
            res = pool.map(_file_tile_unpack, args)
        self.worker_contexts += [r[3] for r in res]
        return [(r[0], r[1], r[2]) for r in res]


def _file_tile_unpack(args):
    from s2p_amd import _lib
    d, k, j = args
    j = dict(j)
    j["rpc1"], j["rpc2"] = _lib.RpcStruct.from_buffer_copy(j["rpc1"]), _lib.RpcStruct.from_buffer_copy(j["rpc2"])
    return _file_tile((d, k, j))
