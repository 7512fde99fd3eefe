"""s2p_amd/common.py -- drop-in for the resampling entry of s2p/common.py.

``image_apply_homography(out, im, H, w, h)`` keeps the reference's signature and file contract
(s2p/common.py:159-180): the reference formats the 9 coefficients into a command line and forks the
`homography` binary; here the source window is decoded, ONE call goes into libs2p_hip.so
(s2p_hip_warp_host: quintic B-spline resampling on the GPU) and the float32 TIFF is written.
"""
import numpy as np

from s2p_amd import _lib
from s2p_amd import io as rio

MARGIN = 16      # source pixels kept around the needed window (quintic prefilter decays as 0.43^k)


def matrix_translation(x, y):           # s2p/common.py:97-101
    t = np.eye(3)
    t[0, 2] = x
    t[1, 2] = y
    return t


def points_apply_homography(H, pts):    # s2p/common.py:183-211
    pts = np.asarray(pts, dtype=np.float64)
    if pts.ndim != 2 or pts.shape[1] < 2:
        raise ValueError("The input must be a numpy array of 2D points, one point per line")
    p = np.hstack((pts[:, 0:2], np.ones((len(pts), 1))))
    q = (np.asarray(H, np.float64) @ p.T).T
    return q[:, 0:2] / q[:, 2:3]


def bounding_box2D(pts):                # s2p/common.py:214-221
    pts = np.asarray(pts)
    mn, mx = pts.min(axis=0), pts.max(axis=0)
    return mn[0], mn[1], mx[0] - mn[0], mx[1] - mn[1]


def source_window(H, w, h, sw, sh, margin=MARGIN):
    """Window of the source raster that H^-1 maps the [0,w]x[0,h] output grid into (+ margin)."""
    Hi = np.linalg.inv(np.asarray(H, np.float64))
    corners = points_apply_homography(Hi, [[0, 0], [w, 0], [0, h], [w, h]])
    x, y, bw, bh = bounding_box2D(corners)
    x0 = int(max(np.floor(x) - margin, 0))
    y0 = int(max(np.floor(y) - margin, 0))
    x1 = int(min(np.ceil(x + bw) + margin + 1, sw))
    y1 = int(min(np.ceil(y + bh) + margin + 1, sh))
    if x1 <= x0 or y1 <= y0:            # the output does not see the source at all
        return 0, 0, min(sw, 1), min(sh, 1)
    return x0, y0, x1, y1


MAX_ZOOM_OUT = 1.5   # source pixels per output pixel (largest singular value of the Jacobian of H^-1) up to which no anti-alias filter is needed


def zoom_out_factor(H, w, h):
    """How many source pixels one output pixel spans at most, anywhere on the [0, w] x [0, h] output grid: the largest singular
    value of the Jacobian of x -> H^-1 x at the four corners and the centre.  1 = same sampling density; 2 = a 2 x zoom-out."""
    Hi = np.linalg.inv(np.asarray(H, np.float64).reshape(3, 3))
    worst = 0.0
    for x, y in ((0, 0), (w, 0), (0, h), (w, h), (w / 2.0, h / 2.0)):
        p = Hi @ np.array([x, y, 1.0])
        # d(p_i / p_2) / dx_j = (Hi[i, j] p_2 - p_i Hi[2, j]) / p_2^2
        J = (Hi[:2, :2] * p[2] - np.outer(p[:2], Hi[2, :2])) / (p[2] * p[2])
        worst = max(worst, float(np.linalg.svd(J, compute_uv=False)[0]))
    return worst


def image_apply_homography(out, im, H, w, h):
    """
    Applies an homography to an image (HIP, MI355X).

    Args: identical to s2p.common.image_apply_homography (s2p/common.py:159-176)
        out: path to the output image file
        im: path to the input image file
        H: numpy array containing the 3x3 homography matrix
        w, h: dimensions (width and height) of the output image

    The output image is defined on the domain [0, w] x [0, h]. Its pixels
    intensities are defined by out(x) = im(H^{-1}(x)).
    """
    w, h = int(w), int(h)               # the reference formats them with "%d" (:180): truncation
    H = np.asarray(H, dtype=np.float64).reshape(3, 3)
    # The resampler INTERPOLATES (quintic B-spline, pinned on the reference's stored rectified tile at zoom 1.00); it has no anti-alias
    # filter, and whether the absent `homography` binary has one for zoom-outs is unknown.  Every call s2p makes is a rectifying
    # similarity with zoom ~ 1 (s2p/rectification.py:242-278), where none is needed -- tests/test_oracle_tile.py checks the
    # interpolation against analytic answers for zooms 0.5 .. 2 on band-limited images.  Beyond MAX_ZOOM_OUT source pixels per output
    # pixel an image with content up to its Nyquist rate would alias: refused, so the caller can fall back to the reference's binary.
    z = zoom_out_factor(H, w, h)
    if z > MAX_ZOOM_OUT:
        raise NotImplementedError("image_apply_homography: H samples the source %.2f pixels apart (zoom-out): beyond %.1f the resampler "
                                  "would need an anti-alias filter, which is not modelled (the `homography` binary's source is absent from the "
                                  "reference tree); s2p's rectifying homographies have zoom ~ 1" % (z, MAX_ZOOM_OUT))
    sw, sh = rio.image_size(im)
    x0, y0, x1, y1 = source_window(H, w, h, sw, sh)
    src = rio.read_window(im, x0, y0, x1, y1)
    Hc = H @ matrix_translation(x0, y0)  # out(x) = crop((H T)^-1 x)
    print("\nRUN (libs2p_hip): homography %s -h \"%s\" %s %d %d" % (im, " ".join(str(v) for v in H.flatten()), out, w, h))
    dst = _lib.warp(src, Hc, w, h)
    rio.write_image(out, dst)


def cargarse_basura(inputf, outputf):
    """
    Remove spurious heights from a height map file (HIP, MI355X): s2p.common.cargarse_basura (s2p/common.py:224-235),
    same arguments (two paths, which may be equal: heights_fusion filters each pair's height_map.tif in place,
    s2p/__init__.py:362-365).  The reference runs morphoop x 4, plambda and remove_small_cc as subprocesses through
    three temporary TIFFs; here the map makes one round trip to the GPU (s2p_hip_cargarse_basura_host).
    """
    a = rio.read_image(inputf)
    print("\nRUN (libs2p_hip): cargarse_basura %s %s" % (inputf, outputf))
    rio.write_image(outputf, _lib.cargarse_basura(a))
