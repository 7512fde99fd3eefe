"""s2p_amd/rectification.py -- the resampling tail of s2p.rectification.rectify_pair.

Only lines 366-382 of s2p/rectification.py belong to the hot path: the final
horizontal margin, the translated homographies, the output size and the two resampling calls.  The
estimation of H1, H2 and of the disparity range above them (RPC virtual matches, affine fundamental
matrix, registration: s2p/rectification.py:310-364) is tiny CPU linear algebra that stays with the
reference; its results are this function's inputs.
"""
import numpy as np

from s2p_amd import common


def rectify_tail(im1, im2, out1, out2, H1, H2, x, y, w, h, disp_m, disp_M, hmargin=0, vmargin=0):
    """
    Resample the ROI of both images with the rectifying homographies (HIP, MI355X).

    Args (names as in s2p.rectification.rectify_pair, s2p/rectification.py:281-282):
        im1, im2: paths to the two GeoTIFF images
        out1, out2: paths to the output rectified crops
        H1, H2: 3x3 rectifying homographies as returned by rectification_homographies / register_*
        x, y, w, h: ROI in the first image
        disp_m, disp_M: horizontal disparity range from disparity_range()
        hmargin, vmargin: margins added on the sides of the rectified images

    Returns:
        H1, H2 (translated by the margins), disp_m, disp_M  -- what rectify_pair returns (:382)
    """
    # the horizontal margin must cover the largest disparity magnitude (s2p/rectification.py:366-367)
    hmargin = int(np.ceil(max(float(hmargin), abs(float(disp_m)), abs(float(disp_M)))))
    shift = common.matrix_translation(hmargin, vmargin)
    H1, H2 = shift @ np.asarray(H1, np.float64), shift @ np.asarray(H2, np.float64)      # (:368-369)

    # size of the rectified ROI of image 1; its corner must land on (hmargin, vmargin) (:371-376)
    corners = np.array([[x, y], [x + w, y], [x + w, y + h], [x, y + h]], np.float64)
    left, top, w0, h0 = common.bounding_box2D(common.points_apply_homography(H1, corners))
    if not np.allclose(np.round([left, top]), [hmargin, vmargin], rtol=1e-7, atol=.01):
        raise AssertionError("H1 does not map the ROI onto the margins: corner (%g, %g), expected (%d, %d)"
                             % (left, top, hmargin, vmargin))

    # both crops have the same size, floats truncated by the resampling entry as "%d" does (:378-380, common.py:180)
    for dst, src, H in ((out1, im1, H1), (out2, im2, H2)):
        common.image_apply_homography(dst, src, H, w0 + 2 * hmargin, h0 + 2 * vmargin)
    return H1, H2, disp_m, disp_M
