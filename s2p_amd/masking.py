"""s2p_amd/masking.py -- drop-in for the mask erosion that follows the matcher
(s2p/masking.py:87-97, called at s2p/__init__.py:189-190): the reference forks
`morsi disk<radius> erosion msk out`; here one call into libs2p_hip.so."""
from s2p_amd import _lib
from s2p_amd import io as rio
import numpy as np


def erosion(out, msk, radius):
    """Shrink the valid (non-zero) area of a 0/1 mask file by a disk of `radius` pixels (HIP, MI355X).

    Same signature and file contract as s2p.masking.erosion: `msk` is read, the eroded mask is written to
    `out` (the two may be the same path, as at s2p/__init__.py:189-190); radii below 2 leave the file untouched,
    as in the reference (s2p/masking.py:96).
    """
    if radius >= 2:
        m = rio.read_image(msk, np.uint8)
        rio.write_image(out, _lib.erode_mask(m, int(radius)))
