"""s2p_amd/fusion.py -- drop-in for fusion.merge_n (s2p/fusion.py:26-68), the pixelwise merge of the
registered height maps of a tile's pairs (tri-stereo tail, called from heights_fusion, s2p/fusion.py:71-105).

The reference stacks the maps in a float64 (h, w, n) array and calls a Python function per pixel through
np.apply_along_axis; here the stack goes through ONE call into libs2p_hip.so (s2p_hip_merge_n_host, one GPU
thread per pixel, the same float64 arithmetic in numpy's evaluation order)."""
import os

import numpy as np

from s2p_amd import _lib
from s2p_amd import io as rio


def merge_n(output, inputs, offsets, averaging='average_if_close', threshold=1, debug=False):
    """
    Merge n images of equal sizes by taking the median/mean/min/max pixelwise (HIP, MI355X).

    Args (as s2p.fusion.merge_n, s2p/fusion.py:26-40):
        inputs: list of paths to the input images
        output: path to the output image
        averaging: 'average_if_close' or the name of a numpy reduction ('np.nanmedian', 'np.median',
            'np.nanmean', 'np.mean', 'np.nanmin', 'np.nanmax', 'np.min', 'np.max')
        threshold: max - min above which average_if_close rejects a pixel
        debug: the reference's cfg['debug'] (:50-52): also write <input>_registered.tif
    """
    assert len(inputs) == len(offsets)
    if not inputs:
        return
    imgs = [rio.read_image(p, np.float32) for p in inputs]
    if debug:
        for p, a, o in zip(inputs, imgs, offsets):
            rio.write_image('{}_registered.tif'.format(os.path.splitext(p)[0]),
                            (a.astype(np.float64) - o + np.mean(offsets)).astype(np.float32))
    avg = _lib.merge_n(imgs, offsets, averaging, threshold)
    rio.update_image(inputs[0], output, avg)      # copy an input file to keep its metadata, then replace the band (:64-68)
