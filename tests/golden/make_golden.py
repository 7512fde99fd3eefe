#!/usr/bin/env python3
"""tests/golden/make_golden.py -- regenerates the golden vectors in this directory.

Run ONLY in the container that has /root/reference: it calls the REAL reference matcher
(oracle/_ref/libsgbm_ref.so, built by `make -C oracle ref` from /root/reference/3rdparty/sgbm) on
seeded synthetic inputs and stores inputs + reference outputs as compressed .npz fixtures.
The fixtures are data (inputs / expected outputs), no reference source text.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
from scipy.ndimage import gaussian_filter

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import pyoracle as po  # noqa: E402


def synth_pair(seed, H, W, disp_fn, gain=1.05, nan=False, sigma=1.0):
    """Blurred-noise rectified pair in the s2p convention im1(x) <-> im2(x + d(x,y))."""
    rng = np.random.default_rng(seed)
    pad = 256
    base = gaussian_filter(rng.uniform(0, 1000, (H, W + 2 * pad)), sigma).astype(np.float32)
    im1 = base[:, pad:pad + W].copy()
    xs = np.arange(W, dtype=np.float64)[None, :] + np.zeros((H, 1))
    ys = np.arange(H, dtype=np.float64)[:, None] + np.zeros((1, W))
    d = disp_fn(xs, ys)
    src = pad + xs - d
    x0 = np.floor(src).astype(int)
    fr = (src - x0).astype(np.float32)
    rows = np.arange(H)[:, None]
    im2 = (gain * ((1 - fr) * base[rows, x0] + fr * base[rows, x0 + 1])).astype(np.float32)
    if nan:
        im1[rng.uniform(size=im1.shape) < 0.01] = np.nan
        im2[5:9, 10:30] = np.nan
    return im1, im2


CASES = {
    # name: (seed, H, W, dmin, dmax, nan, full-dump?, disparity field)
    "sgbm_tiny_sym": (1, 48, 64, -16, 16, False, True, lambda x, y: 6 * np.sin(x / 23.) * np.cos(y / 19.)),
    "sgbm_tiny_nan": (2, 40, 70, -7, 21, True, True, lambda x, y: 7 + 5 * np.sin(x / 29.) * np.cos(y / 17.)),
    "sgbm_pos_range": (3, 33, 50, 5, 30, False, True, lambda x, y: 17 + 4 * np.sin(x / 15.)),
    "sgbm_neg_range_oob": (10, 100, 150, -50, -10, False, False, lambda x, y: -30 + 6 * np.sin(x / 31.) * np.cos(y / 23.)),
    "sgbm_asym": (7, 96, 128, -20, 50, False, False, lambda x, y: 15 + 12 * np.sin(x / 41.) * np.cos(y / 37.)),
    "sgbm_256_d64": (9, 256, 256, -32, 32, False, False, lambda x, y: 20 * np.sin(2 * np.pi * x / 256.) * np.cos(2 * np.pi * y / 256.)),
}


REFDATA = "/root/reference/tests/data"


def crop_for(H, w, h, sw, sh, margin=16):
    """Source window (x0, y0, x1, y1) that H^-1 maps the w x h output grid into, plus a margin."""
    Hi = np.linalg.inv(H)
    c = np.array([[0, 0, 1], [w, 0, 1], [0, h, 1], [w, h, 1]], np.float64).T
    p = Hi @ c
    p = p[:2] / p[2]
    x0 = max(int(np.floor(p[0].min())) - margin, 0)
    y0 = max(int(np.floor(p[1].min())) - margin, 0)
    x1 = min(int(np.ceil(p[0].max())) + margin, sw)
    y1 = min(int(np.ceil(p[1].max())) + margin, sh)
    return x0, y0, x1, y1


def reference_tile_fixtures():
    """The one disparity-level artefact the reference's tests hold: inputs of
    tests/triangulation_test.py.  Data files only (rasters and the two 3x3 matrices):
      warp_tile.npz : crop of input_pair/img_01.tif + H_ref  ->  rectified_ref.tif      (`homography` output)
      mgm_tile.npz  : rectified_ref.tif + crop of img_02.tif + H_sec -> rectified_disp.tif (`mgm` output), mask
    Crops keep the fixtures small; the crop offset is folded into the matrices (H' = H . T(x0, y0))."""
    from PIL import Image
    t = os.path.join(REFDATA, "input_triangulation", "pair_1")
    ref = np.array(Image.open(os.path.join(t, "rectified_ref.tif"))).astype(np.float32)
    disp = np.array(Image.open(os.path.join(t, "rectified_disp.tif"))).astype(np.float32)
    mask = np.array(Image.open(os.path.join(t, "rectified_mask.png"))).astype(np.uint8)
    Href = np.loadtxt(os.path.join(t, "H_ref.txt"))
    Hsec = np.loadtxt(os.path.join(t, "H_sec.txt"))
    im1 = np.array(Image.open(os.path.join(REFDATA, "input_pair", "img_01.tif")))
    im2 = np.array(Image.open(os.path.join(REFDATA, "input_pair", "img_02.tif")))
    h, w = ref.shape
    for name, im, H, extra in (("warp_tile", im1, Href, dict(expected=ref)),
                               ("mgm_tile", im2, Hsec, dict(ref=ref, disp=disp, mask=mask))):
        x0, y0, x1, y1 = crop_for(H, w, h, im.shape[1], im.shape[0])
        T = np.array([[1, 0, x0], [0, 1, y0], [0, 0, 1]], np.float64)
        out = dict(src=np.ascontiguousarray(im[y0:y1, x0:x1]), H=H @ T, size=np.array([w, h], np.int32),
                   crop=np.array([x0, y0, x1, y1], np.int32), **extra)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print("%-22s src crop %s %s  %.0f KB" % (name, out["src"].shape, out["src"].dtype, os.path.getsize(path) / 1024))


def input_pair_fixture():
    """input_pair.npz: the two rasters of tests/data/input_pair (the reference's end-to-end test input) with the two
    rectifying homographies its tests hold for that pair (tests/data/input_triangulation/pair_1/H_ref.txt, H_sec.txt,
    full-image coordinates) and the tile [x, y, w, h] they were computed for.  Data files only.  Used for BASELINE
    configs[2]: the full ROI tiled 512 x 512 in the image, every tile rectified into the frame of these homographies
    (integer shifts of it, so that the stored rectified_disp.tif of mgm_tile.npz overlays exactly) and matched by
    `mgm_multi` over 192 disparities."""
    from PIL import Image
    t = os.path.join(REFDATA, "input_triangulation", "pair_1")
    out = dict(img_01=np.array(Image.open(os.path.join(REFDATA, "input_pair", "img_01.tif"))),
               img_02=np.array(Image.open(os.path.join(REFDATA, "input_pair", "img_02.tif"))),
               H_ref=np.loadtxt(os.path.join(t, "H_ref.txt")), H_sec=np.loadtxt(os.path.join(t, "H_sec.txt")),
               tile=np.array([500, 150, 350, 350], np.int32))
    path = os.path.join(HERE, "input_pair.npz")
    np.savez_compressed(path, **out)
    print("%-22s %s %s + %s %s  %.0f KB" % ("input_pair", out["img_01"].shape, out["img_01"].dtype, out["img_02"].shape,
                                           out["img_02"].dtype, os.path.getsize(path) / 1024))


def triangulation_fixture():
    """tri_tile.npz: inputs of the triangulation step for the reference's tile (the data files of
    tests/data/input_triangulation + the RPC tag of the two input images) and the output of the reference's own
    disp_to_lonlatalt (oracle/_ref/libdisp_to_h_ref.so), every 4th pixel."""
    from PIL import Image
    t = os.path.join(REFDATA, "input_triangulation")
    out = dict(
        rpc1=np.array(Image.open(os.path.join(REFDATA, "input_pair", "img_01.tif")).tag_v2[50844], np.float64),
        rpc2=np.array(Image.open(os.path.join(REFDATA, "input_pair", "img_02.tif")).tag_v2[50844], np.float64),
        H_ref=np.loadtxt(os.path.join(t, "pair_1", "H_ref.txt")), H_sec=np.loadtxt(os.path.join(t, "pair_1", "H_sec.txt")),
        A=np.loadtxt(os.path.join(t, "global_pointing_pair_1.txt")),
        mask_orig=np.array(Image.open(os.path.join(t, "mask.png"))).astype(np.uint8),
        mask_rect=np.array(Image.open(os.path.join(t, "pair_1", "rectified_mask.png"))).astype(np.uint8),
        tile=np.array([500, 150, 350, 350], np.int32))
    disp = np.array(Image.open(os.path.join(t, "pair_1", "rectified_disp.tif"))).astype(np.float32)
    r1, r2 = po.rpc_from_geotiff_tag(out["rpc1"]), po.rpc_from_geotiff_tag(out["rpc2"])
    x, y, w, h = (int(v) for v in out["tile"])
    lla, err = po.ref_disp_to_lonlatalt(r1, r2, out["H_ref"], out["H_sec"] @ np.linalg.inv(out["A"]), disp,
                                        out["mask_rect"], (x, x + w, y, y + h), out["mask_orig"])
    out["lonlatalt_4"] = lla[::4, ::4].copy()
    out["err_4"] = err[::4, ::4].copy()
    path = os.path.join(HERE, "tri_tile.npz")
    np.savez_compressed(path, **out)
    print("%-22s valid=%.3f alt=[%.1f, %.1f]  %.0f KB" % ("tri_tile", np.isfinite(err).mean(), np.nanmin(lla[..., 2]),
                                                            np.nanmax(lla[..., 2]), os.path.getsize(path) / 1024))


def fusion_fixture():
    """fusion_stack.npz: seeded stacks of height maps + offsets and the output of the array core of
    fusion.merge_n (s2p/fusion.py:46-68) evaluated with the reference's OWN average_if_close: the function is
    compiled here from /root/reference/s2p/fusion.py (its module cannot be imported: rasterio is absent) and
    handed to the same np.apply_along_axis call the reference makes."""
    import ast
    tree = ast.parse(open("/root/reference/s2p/fusion.py").read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "average_if_close"]
    ns = {"np": np}
    exec(compile(ast.Module(body=fn, type_ignores=[]), "s2p/fusion.py", "exec"), ns)
    rng = np.random.default_rng(77)
    out = {}
    for k, (n, thr) in enumerate(((2, 3.0), (3, 1.0), (5, 2.5), (9, 4.0))):
        base = rng.uniform(20, 60, (37, 53))
        st = [(base + rng.normal(0, 1.0, base.shape) + 7.0 * i).astype(np.float32) for i in range(n)]
        for a in st:
            a[rng.uniform(size=a.shape) < 0.15] = np.nan
            a[rng.uniform(size=a.shape) < 0.01] = np.inf
        st[0][:3] = np.nan
        for a in st:
            a[3:5, :7] = np.nan
        offs = [7.0 * i + float(rng.normal(0, 0.3)) for i in range(n)]
        out["stack%d" % k] = np.stack(st)
        out["offsets%d" % k] = np.array(offs)
        out["threshold%d" % k] = np.float64(thr)
        out["expected%d" % k] = po.oracle_merge_n(st, offs, "average_if_close", thr, fn=ns["average_if_close"])
    path = os.path.join(HERE, "fusion_stack.npz")
    np.savez_compressed(path, **out)
    print("%-22s %d stacks, finite=%.3f  %.0f KB" % ("fusion_stack", 4, np.isfinite(out["expected1"]).mean(), os.path.getsize(path) / 1024))


def filter3d_fixture():
    """filter3d.npz: a gridded cloud + keypoint matches and the outputs of the reference's own count_3d_neighbors,
    remove_isolated_3d_points and stereo_corresp_to_lonlatalt (oracle/_ref/libdisp_to_h_ref.so = c/disp_to_h.c:43-67,
    143-230 compiled from the reference tree)."""
    sys.path.insert(0, os.path.dirname(HERE))
    from helpers import synth_cloud
    xyz = synth_cloud(31, 72, 96)
    out = dict(xyz=xyz, params=np.array([1.0, 2, 9, 1]))          # r, p, n, q
    out["count"] = po.ref_count_3d_neighbors(xyz, 1.0, 2)
    out["removed"] = np.isnan(po.ref_remove_isolated_3d_points(xyz, 1.0, 2, 9, 1)[:, :, 0])
    t = np.load(os.path.join(HERE, "tri_tile.npz"))
    r1, r2 = po.rpc_from_geotiff_tag(t["rpc1"]), po.rpc_from_geotiff_tag(t["rpc2"])
    rng = np.random.default_rng(5)
    pts1 = np.stack([rng.uniform(400, 900, 40), rng.uniform(100, 500, 40)], axis=1).astype(np.float32)
    # plausible matches: project the 3-D point of pts1 at a random altitude into image 2 is not needed here --
    # rpc_height is defined for any pair of points; a small offset keeps the residuals moderate
    pts2 = (pts1 + np.stack([rng.uniform(-30, 30, 40), rng.uniform(-0.5, 0.5, 40)], axis=1)).astype(np.float32)
    lla, err = po.ref_stereo_corresp_to_lonlatalt(r1, r2, pts1, pts2)
    out.update(pts1=pts1, pts2=pts2, corresp_lonlatalt=lla, corresp_err=err)
    path = os.path.join(HERE, "filter3d.npz")
    np.savez_compressed(path, **out)
    print("%-22s count max %d, removed %.3f (nan in %.3f), corresp alt [%.0f, %.0f]  %.0f KB" % (
        "filter3d", out["count"].max(), out["removed"].mean(), np.isnan(xyz[:, :, 0]).mean(), lla[:, 2].min(), lla[:, 2].max(),
        os.path.getsize(path) / 1024))


def plyflatten_fixture(only=False):
    """plyflatten_crop.npz: a 160 x 160 window of the reference's rasterisation golden -- the points of
    tests/data/input_ply/cloud.ply that fall into the window (x, y, z float64, r, g, b uint8) and the same window of
    tests/data/expected_output/plyflatten/dsm_40cm.tiff (tests/rasterization_test.py:13-28), with the full-raster roi
    that plyflatten_from_plyfiles_list derives from the whole cloud.  Both are data files of the reference's tests."""
    from PIL import Image
    ref = "/root/reference/tests/data"
    exp = np.array(Image.open(os.path.join(ref, "expected_output/plyflatten/dsm_40cm.tiff")))
    raw = open(os.path.join(ref, "input_ply/cloud.ply"), "rb").read()
    hdr = raw.index(b"end_header\n") + len(b"end_header\n")
    n = int([l for l in raw[:hdr].decode().splitlines() if l.startswith("element vertex")][0].split()[-1])
    dt = np.dtype([("x", "<f8"), ("y", "<f8"), ("z", "<f8"), ("r", "u1"), ("g", "u1"), ("b", "u1")])
    pts = np.frombuffer(raw[hdr:], dtype=dt, count=n)
    res = 0.4
    xoff = np.floor(pts["x"].min() / res) * res
    yoff = np.ceil(pts["y"].max() / res) * res
    xsize = int(1 + np.floor((pts["x"].max() - xoff) / res))
    ysize = int(1 - np.floor((pts["y"].min() - yoff) / res))
    assert exp.shape == (ysize, xsize)
    r0, c0, hh, ww = 150, 120, 160, 160
    i = np.floor((pts["x"] - xoff) / res).astype(int)
    j = np.floor((yoff - pts["y"]) / res).astype(int)
    keep = (i >= c0) & (i < c0 + ww) & (j >= r0) & (j < r0 + hh)
    sub = pts[keep]                                            # input order preserved: the running mean depends on it
    out = dict(xyz=np.stack([sub["x"], sub["y"], sub["z"]], 1), rgb=np.stack([sub["r"], sub["g"], sub["b"]], 1),
               roi=np.array([xoff, yoff, xsize, ysize]), resolution=np.array(res), window=np.array([r0, c0, hh, ww]),
               expected=exp[r0:r0 + hh, c0:c0 + ww].copy(), comments=np.array("projection: CRS epsg:32740"))
    path = os.path.join(HERE, "plyflatten_crop.npz")
    np.savez_compressed(path, **out)
    print("%-22s %d of %d points, window %dx%d finite=%.3f  %.0f KB" % ("plyflatten_crop", keep.sum(), n, ww, hh,
          np.isfinite(out["expected"]).mean(), os.path.getsize(path) / 1024))


def main():
    if "plyflatten" in sys.argv[1:]:
        return plyflatten_fixture()
    if "input_pair" in sys.argv[1:]:
        return input_pair_fixture()
    assert po.have_ref(), "build the reference first: make -C oracle ref"
    fusion_fixture()
    plyflatten_fixture()
    if po.have_ref_tri():
        filter3d_fixture()
    reference_tile_fixtures()
    input_pair_fixture()
    if po.have_ref_tri():
        triangulation_fixture()
    for name, (seed, H, W, dmin, dmax, nan, full, fn) in CASES.items():
        im1, im2 = synth_pair(seed, H, W, fn, nan=nan)
        r = po.ref_sgbm(im1, im2, dmin, dmax, dump="full" if full else True)
        out = dict(im1=im1, im2=im2, params=np.array([dmin, dmax, 3, 8, 32, 1], np.int32),
                   geom=np.array(r["geom"], np.int32), rminmax=np.array(r["rminmax"], np.float32))
        for k in ("q1", "q2", "C", "S", "disp_raw", "disp_med", "disp_fin", "cost_raw", "disp", "cost"):
            if k in r:
                out[k] = r[k]
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print("%-22s %4dx%-4d d=[%d,%d] valid=%.3f  %.0f KB" % (
            name, W, H, dmin, dmax, np.isfinite(r["disp"]).mean(), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
