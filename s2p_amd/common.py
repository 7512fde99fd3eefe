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
