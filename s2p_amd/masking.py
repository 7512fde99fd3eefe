"""s2p_amd/masking.py -- drop-in for the mask erosion that follows the matcher
(s2p/masking.py:87-97, called at s2p/__init__.py:189-190): the reference forks
`morsi disk<radius> erosion msk out`; here one call into libs2p_hip.so."""
from s2p_amd import _lib
from s2p_amd import io as rio
import numpy as np


def erosion(out, msk, radius):
    """
    Erodes the accepted regions (ie eliminates more pixels)

    Args:
        out: path to the ouput mask image file
        msk: path to the input mask image file
        radius (in pixels): size of the disk used for the erosion
    """
    if radius >= 2:
        m = rio.read_image(msk, np.uint8)
        rio.write_image(out, _lib.erode_mask(m, int(radius)))
