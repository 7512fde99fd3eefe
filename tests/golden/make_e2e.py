#!/usr/bin/env python3
"""tests/golden/make_e2e.py -- fixtures for the end-to-end DSM tests (build container only: reads /root/reference).

The reference's only out-of-sample evidence for its matcher are the artefacts of tests/end2end_test.py:
    tests/data/input_pair     -> expected_output/pair/dsm.tif                          (tolerances 0.025 m / 1 m)
    tests/data/input_triplet  -> expected_output/triplet/{height_map.tif, dsm.tif}     (tolerances 0.05 m / 2 m)
This script prepares, per dataset, what the GPU path needs to redo steps 3-7 of s2p.main on those inputs and stores it
as data (tests/golden/e2e_pair.npz, e2e_triplet.npz): the rasters, the RPC tags, the tiles of the reference's tiling
(ROI 150,150,700,700, tile_size 300 -> 2 x 2 tiles of 350 x 350: s2p/initialization.py:169-183), per tile and pair the
two rectifying homographies and the disparity range, the pointing correction, and the expected rasters with their grid.

What the reference computes upstream of the hot path and how it is obtained here:
  * virtual matches from the RPCs (s2p/rpc_utils.py:356-376) -- restated below on the reference's own rpc_projection /
    rpc_localization (c/rpc.c, compiled where it lies: oracle/_ref/libdisp_to_h_ref.so);
  * affine fundamental matrix and rectifying similarities -- the reference's own s2p/estimation.py, imported from
    /root/reference (it is numpy-only);
  * the shear / translation registration and the margins of rectify_pair (s2p/rectification.py:325-375) -- restated;
  * the POINTING CORRECTION (steps 1-2 of s2p.main) needs SIFT keypoints: 3rdparty/sift is an empty submodule.  The
    reference's correction is, by construction, a translation of image 2 along the normal of the epipolar direction
    (pointing_accuracy.local_translation: the median of the matches' error vectors, which are all collinear with that
    normal because F is affine), i.e. exactly the residual VERTICAL offset of the rectified pair.  It is measured here
    from the images themselves (dense matching with the CPU oracle, then a least-squares vertical shift between the
    locally normalised rectified images, iterated until the residual vanishes).  For input_pair the reference holds the
    correction it computed (tests/data/input_triangulation/global_pointing_pair_1.txt): the measurement reproduces its
    translation to a few hundredths of a pixel (printed below), and the stored matrix is what the fixture carries;
  * the disparity range ('sift' method: min / max over the SIFT matches + 20 %) -- from the same dense matching: 0.1 /
    99.9 percentiles + 20 %; the horizontal registration (mean disparity of the SIFT matches) likewise.

    python tests/golden/make_e2e.py [pair] [triplet]
"""
import ctypes
import importlib.util
import os
import sys

import numpy as np
from scipy import ndimage

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import pyoracle as po  # noqa: E402

REF = "/root/reference"
DATA = os.path.join(REF, "tests", "data")


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


est = _load(os.path.join(REF, "s2p", "estimation.py"), "ref_estimation")


class Cam:
    """RPC camera on the reference's own C functions (c/rpc.c:464-472)."""

    def __init__(self, tag):
        self.tag = np.array(tag, np.float64)
        self.s = po.rpc_from_geotiff_tag(self.tag)
        self.lib = ctypes.CDLL(po.REF_TRI_SO)
        self.alt_offset, self.alt_scale = float(self.tag[6]), float(self.tag[11])
        self.row_offset, self.col_offset = float(self.tag[2]), float(self.tag[3])

    def _each(self, fn, a, b, c):
        a, b, c = np.broadcast_arrays(np.asarray(a, np.float64), np.asarray(b, np.float64), np.asarray(c, np.float64))
        out = np.empty(a.shape + (2,))
        o2, i3 = (ctypes.c_double * 2)(), (ctypes.c_double * 3)()
        for k in np.ndindex(a.shape):
            i3[:] = [a[k], b[k], c[k]]
            fn(o2, ctypes.byref(self.s), i3)
            out[k] = o2[:]
        return out[..., 0], out[..., 1]

    def projection(self, lon, lat, alt):
        return self._each(self.lib.rpc_projection, lon, lat, alt)

    def localization(self, col, row, alt):
        return self._each(self.lib.rpc_localization, col, row, alt)


def T(x, y):
    return np.array([[1., 0., x], [0., 1., y], [0., 0., 1.]])


def apply_h(H, pts):
    p = np.column_stack([np.asarray(pts, np.float64), np.ones(len(pts))]) @ np.asarray(H).T
    return p[:, :2] / p[:, 2:]


def bbox(pts):
    x0, y0 = pts.min(0)
    x1, y1 = pts.max(0)
    return x0, y0, x1 - x0, y1 - y0


def gcp(cam, x, y, w, h, m, M, n):
    """s2p/rpc_utils.py:263-316: n^3 points on a grid slightly inside the ROI, at n altitudes."""
    cols = np.linspace(x + w / (2. * n), x + (2 * n - 1.) * w / (2 * n), n)
    rows = np.linspace(y + h / (2. * n), y + (2 * n - 1.) * h / (2 * n), n)
    alts = np.linspace(m, M, n)
    a, r, c = np.meshgrid(alts, rows, cols, indexing="ij")
    lon, lat = cam.localization(c.ravel(), r.ravel(), a.ravel())
    return lon, lat, a.ravel()


def matches_from_rpc(c1, c2, x, y, w, h, n=5):
    """s2p/rpc_utils.py:356-376 with the coarse altitude range (no SRTM, no exogenous DEM: :76-89, :190-197)."""
    m, M = c1.alt_offset - c1.alt_scale, c1.alt_offset + c1.alt_scale
    lon, lat, alt = gcp(c1, x, y, w, h, m, M, n)
    x1, y1 = c1.projection(lon, lat, alt)
    x2, y2 = c2.projection(lon, lat, alt)
    return np.column_stack([x1, y1, x2, y2])


def rectifying_pair(c1, c2, x, y, w, h, A):
    """H1, H2 up to the horizontal registration: s2p/rectification.py:325-351 (+ :262-279)."""
    m = matches_from_rpc(c1, c2, x, y, w, h)
    m[:, 2:] = apply_h(np.linalg.inv(A), m[:, 2:])
    F = est.affine_fundamental_matrix(m)
    S1, S2 = est.rectifying_similarities_from_affine_fundamental_matrix(F, False)
    x0, y0 = bbox(apply_h(S1, [[x, y], [x + w, y], [x + w, y + h], [x, y + h]]))[:2]
    H1, H2 = T(-x0, -y0) @ S1, T(-x0, -y0) @ S2
    # register_with_shear (:343-351, :52-86): ground points at the mean altitude of the range
    a = c1.alt_offset
    lon, lat, alt = gcp(c1, x, y, w, h, a, a, 4)
    g = np.unique(np.column_stack(c1.projection(lon, lat, alt) + c2.projection(lon, lat, alt)), axis=0)
    p1, p2 = apply_h(H1, g[:, :2]), apply_h(H2, g[:, 2:])
    abc = np.linalg.lstsq(np.column_stack([p2[:, 0], p2[:, 1], np.ones(len(p2))]), p1[:, 0], rcond=None)[0]
    H2 = np.array([[abc[0], abc[1], abc[2]], [0, 1, 0], [0, 0, 1]]) @ H2
    return H1, H2


def final_frame(H1, H2, x, y, w, h, disp_m, disp_M, hmargin, vmargin):
    """s2p/rectification.py:366-380: margins, size of the rectified tile."""
    hm = int(np.ceil(max([hmargin, abs(disp_m), abs(disp_M)])))
    H1, H2 = T(hm, vmargin) @ H1, T(hm, vmargin) @ H2
    x0, y0, w0, h0 = bbox(apply_h(H1, [[x, y], [x + w, y], [x + w, y + h], [x, y + h]]))
    assert np.allclose(np.round([x0, y0]), [hm, vmargin], atol=.01)
    return H1, H2, int(w0 + 2 * hm), int(h0 + 2 * vmargin)


def coarse_offset(im1, im2, H1, H2, roi, zoom=4, search=1400):
    """Horizontal offset of image 2 in the rectified frame of the whole ROI, at 1 / zoom resolution.  The virtual
    matches span the RPC's altitude range (1295 +- 1315 m for input_pair) and the shear registration sits at its
    middle, while the scene lies near 2300 m: hundreds of pixels, which the reference absorbs with the mean disparity
    of its SIFT matches (register_horizontally_translation, s2p/rectification.py:89-136).  Measured with the tile's
    own H1, H2 (the sign of the affine F decides whether the rectified frame is turned by 180 degrees) on the whole ROI."""
    x, y, w, h = roi
    Z = np.diag([1. / zoom, 1. / zoom, 1.])
    s = search // zoom
    b1, b2 = (ndimage.gaussian_filter(a.astype(np.float32), zoom / 2.) for a in (im1, im2))
    x0, y0, w0, h0 = bbox(apply_h(Z @ H1, [[x, y], [x + w, y], [x + w, y + h], [x, y + h]]))
    ww, hh = int(w0) + 2 * s, int(h0)
    P = T(s - x0, -y0) @ Z
    r1, r2 = po.oracle_warp(b1, P @ H1, ww, hh), po.oracle_warp(b2, P @ H2, ww, hh)
    d = po.oracle_census_sgm(r1, r2, -s, s, params=po.census_params(recursion=0))["disp"]
    v = d[np.isfinite(d)]
    print("  coarse offset: %.0f px (1/%d scale: %.0f %% valid, quartiles %.0f / %.0f)" % (
        zoom * np.median(v), zoom, 100 * np.isfinite(d).mean(), zoom * np.percentile(v, 25), zoom * np.percentile(v, 75)))
    return float(zoom * np.median(v))


def coarse_match(im1, im2, H1, H2, x, y, w, h, t0=0., search=120):
    """Dense disparity of the tile on the CPU oracle over a wide range (the stand-in for the SIFT matches)."""
    H2 = T(-t0, 0) @ H2
    G1, G2, ww, hh = final_frame(H1, H2, x, y, w, h, -search, search, 0, 5)
    r1, r2 = po.oracle_warp(im1, G1, ww, hh), po.oracle_warp(im2, G2, ww, hh)
    r = po.oracle_census_sgm(r1, r2, -search, search, params=po.census_params(recursion=0))
    return r1, r2, r["disp"], G1, G2


def vertical_residual(r1, r2, disp):
    """Least-squares vertical shift e with  r2(x + d, y) ~ r1(x, y - e)  on locally normalised images, robustly
    (median of 48 x 48 blocks)."""
    hh, ww = r1.shape
    ys, xs = np.mgrid[0:hh, 0:ww].astype(np.float64)
    ok = np.isfinite(disp)
    ok &= np.isfinite(r1)
    r1 = np.nan_to_num(r1)
    w2 = ndimage.map_coordinates(np.nan_to_num(r2), [ys, xs + np.where(ok, disp, 0)], order=3, mode="nearest")
    ok &= ndimage.map_coordinates(np.isfinite(r2).astype(np.float64), [ys, xs + np.where(ok, disp, 0)], order=1, mode="constant") > 0.999

    def norm(a):
        m = ndimage.uniform_filter(a, 9)
        s = np.sqrt(np.maximum(ndimage.uniform_filter(a * a, 9) - m * m, 1e-6))
        return (a - m) / s
    a, b = norm(r1.astype(np.float64)), norm(w2)
    gy = np.gradient(0.5 * (a + b), axis=0)
    ok &= ndimage.binary_erosion(ok, iterations=6)
    est_ = []
    for by in range(0, hh - 47, 48):
        for bx in range(0, ww - 47, 48):
            s = np.s_[by:by + 48, bx:bx + 48]
            k = ok[s]
            if k.sum() > 800:
                est_.append(-(gy[s][k] * (b[s][k] - a[s][k])).sum() / (gy[s][k] ** 2).sum())
    return float(np.median(est_)), len(est_)


def measure_pointing(im1, im2, c1, c2, tiles, roi, A0=None, iters=4):
    """Translation of image 2 along the epipolar normal that cancels the vertical residual of the rectified pair,
    measured over all tiles (the counterpart of s2p/pointing_accuracy.py:63-103 + global_from_local with one value)."""
    A = np.eye(3) if A0 is None else A0.copy()
    t0 = {}
    for it in range(iters):
        es = []
        for (x, y, w, h) in tiles:
            H1, H2 = rectifying_pair(c1, c2, x, y, w, h, A)
            if it == 0:
                t0[(x, y)] = coarse_offset(im1, im2, H1, H2, roi)
            r1, r2, d, G1, G2 = coarse_match(im1, im2, H1, H2, x, y, w, h, t0[(x, y)])
            e, n = vertical_residual(r1, r2, d)
            # a vertical offset e of the rectified image 2 = the image-2 vector  G2_lin^-1 (0, e)
            v = np.linalg.inv((G2 @ np.linalg.inv(A))[:2, :2]) @ np.array([0., e])
            es.append(v)
            print("    it %d tile %s: residual %+.3f px over %d blocks" % (it, (x, y, w, h), e, n))
        v = np.median(np.array(es), axis=0)
        # image-2 content sits v too far along the normal: corrected point = A p, the reference's convention
        # (s2p/pointing_accuracy.py:99-102: A = translation(-median error vector))
        A = T(-v[0], -v[1]) @ A
        print("  iteration %d: step (%+.4f, %+.4f) -> A translation (%+.4f, %+.4f)" % (it, -v[0], -v[1], A[0, 2], A[1, 2]))
        if np.hypot(*v) < 0.01:
            break
    return A, t0


def tiles_of(roi, tile_size):
    """s2p/initialization.py:160-205."""
    rx, ry, rw, rh = roi
    tw = int(np.ceil(rw / int(np.round(rw / min(rw, tile_size)))))
    th = int(np.ceil(rh / int(np.round(rh / min(rh, tile_size)))))
    return [(x, y, min(tw, rx + rw - x), min(th, ry + rh - y)) for y in range(ry, ry + rh, th) for x in range(rx, rx + rw, tw)]


def read_tif(path):
    from PIL import Image
    im = Image.open(path)
    return np.array(im), im.tag_v2


def gsd_from_rpc(cam):
    """s2p/rpc_utils.py:477-495 (geocentric distance between two neighbouring pixel centres at z = 0)."""
    def ecef(lon, lat, alt):
        a, f = 6378137.0, 1 / 298.257223563
        e2 = f * (2 - f)
        lo, la = np.radians(lon), np.radians(lat)
        N = a / np.sqrt(1 - e2 * np.sin(la) ** 2)
        return np.array([(N + alt) * np.cos(la) * np.cos(lo), (N + alt) * np.cos(la) * np.sin(lo), (N * (1 - e2) + alt) * np.sin(la)])
    c, r = cam.col_offset, cam.row_offset
    p = [ecef(*[float(v) for v in cam.localization(c + k, r, 0.)], 0.) for k in (0, 1)]
    return float(np.linalg.norm(p[1] - p[0]))


def prepare(name, images, roi, tile_size, hmargin, vmargin, A_known=None):
    ims, cams = [], []
    for p in images:
        a, tags = read_tif(os.path.join(DATA, name, p))
        ims.append(a)
        cams.append(Cam(tags[50844]))
    tiles = tiles_of(roi, tile_size)
    out = dict(tiles=np.array(tiles, np.int32), roi=np.array(roi, np.int32), gsd=np.float64(gsd_from_rpc(cams[0])),
               margins=np.array([hmargin, vmargin], np.int32))
    for k, (a, c) in enumerate(zip(ims, cams)):
        out["img_%d" % k] = a
        out["rpc_%d" % k] = c.tag
    for i in range(1, len(ims)):
        print("%s pair %d" % (name, i))
        A, t0 = measure_pointing(ims[0], ims[i], cams[0], cams[i], tiles, roi)
        if A_known is not None and i == 1:
            print("  measured translation (%+.4f, %+.4f) vs the reference's stored correction (%+.4f, %+.4f)"
                  % (A[0, 2], A[1, 2], A_known[0, 2], A_known[1, 2]))
            out["A_measured_%d" % i] = A
            A = A_known
        out["A_%d" % i] = A
        for t, (x, y, w, h) in enumerate(tiles):
            H1, H2 = rectifying_pair(cams[0], cams[i], x, y, w, h, A)
            r1, r2, d, G1, G2 = coarse_match(ims[0], ims[i], H1, H2, x, y, w, h, t0[(x, y)])
            v = d[np.isfinite(d)]
            H2 = T(-t0[(x, y)] - float(v.mean()), 0) @ H2                             # register_horizontally_translation, 'center'
            v = v - v.mean()
            lo, hi = np.floor(np.percentile(v, 0.1)), np.ceil(np.percentile(v, 99.9))
            lo = lo - 0.2 * (hi - lo)                                     # disp_range_extra_margin, in the order of :156-158
            hi = hi + 0.2 * (hi - lo)
            lo, hi = min(-3, lo), max(3, hi)
            F1, F2, ww, hh = final_frame(H1, H2, x, y, w, h, lo, hi, hmargin, vmargin)
            out["H_ref_%d_%d" % (i, t)] = F1
            out["H_sec_%d_%d" % (i, t)] = F2
            out["size_%d_%d" % (i, t)] = np.array([ww, hh], np.int32)
            out["disp_range_%d_%d" % (i, t)] = np.array([lo, hi])
            print("  tile %d %s: rectified %d x %d, disparity range [%.1f, %.1f]" % (t, (x, y, w, h), ww, hh, lo, hi))
    return out


def expected(path):
    a, tags = read_tif(path)
    d = dict(raster=a.astype(np.float32))
    if 33922 in tags:
        tp, sc = tags[33922], tags[33550]
        d["origin"] = np.array([tp[3], tp[4]], np.float64)
        d["resolution"] = np.float64(sc[0])
    return d


def main():
    assert po.have_ref_tri(), "make -C oracle ref_tri"
    which = sys.argv[1:] or ["pair", "triplet"]
    if "pair" in which:
        A = np.loadtxt(os.path.join(DATA, "input_triangulation", "global_pointing_pair_1.txt"))
        out = prepare("input_pair", ["img_01.tif", "img_02.tif"], (150, 150, 700, 700), 300, 20, 5, A_known=A)
        e = expected(os.path.join(DATA, "expected_output", "pair", "dsm.tif"))
        out.update(dsm=e["raster"], dsm_origin=e["origin"], dsm_resolution=e["resolution"], out_crs=np.array("epsg:32740"),
                   filtering=np.array([5., 50.]))
        np.savez_compressed(os.path.join(HERE, "e2e_pair.npz"), **out)
        print("e2e_pair.npz %.0f KB" % (os.path.getsize(os.path.join(HERE, "e2e_pair.npz")) / 1024))
    if "triplet" in which:
        out = prepare("input_triplet", ["img_02.tif", "img_01.tif", "img_03.tif"], (150, 150, 700, 700), 300, 20, 5)
        e = expected(os.path.join(DATA, "expected_output", "triplet", "dsm.tif"))
        hm = expected(os.path.join(DATA, "expected_output", "triplet", "height_map.tif"))
        out.update(dsm=e["raster"], dsm_origin=e["origin"], dsm_resolution=e["resolution"], out_crs=np.array("epsg:32631"),
                   height_map_pair_1=hm["raster"])
        np.savez_compressed(os.path.join(HERE, "e2e_triplet.npz"), **out)
        print("e2e_triplet.npz %.0f KB" % (os.path.getsize(os.path.join(HERE, "e2e_triplet.npz")) / 1024))


if __name__ == "__main__":
    main()
