"""s2p_amd -- MI355X-native stereo-matching hot path for the S2P pipeline.

Drop-in mirrors of the reference functions that own the path today:
  s2p_amd.block_matching.compute_disparity_map   (s2p/block_matching.py:35-336)
  s2p_amd.common.image_apply_homography          (s2p/common.py:159-180)
backed by hand-written gfx950 kernels in libs2p_hip.so (C ABI: include/s2p_hip.h).
Importing this package does not touch the GPU.
"""
__version__ = "0.1.0"
