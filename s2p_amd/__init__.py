"""s2p_amd -- MI355X-native stereo-matching hot path for the S2P pipeline.

Drop-in mirrors of the reference functions that own the path today:
  s2p_amd.block_matching.compute_disparity_map   (s2p/block_matching.py:35-336)
  s2p_amd.common.image_apply_homography          (s2p/common.py:159-180)
and of the steps either side of the matcher:
  s2p_amd.rectification.rectify_tail (the resampling end of rectify_pair), s2p_amd.masking.erosion,
  s2p_amd.triangulation.{disp_to_lonlatalt, height_map, stereo_corresp_to_lonlatalt, filter_xyz, ...}
  s2p_amd.fusion.merge_n                         (s2p/fusion.py:26-68)
  s2p_amd.rasterization.plyflatten_from_plyfiles_list   (the plyflatten package, s2p/__init__.py:462-466)
  s2p_amd.tiles.process_tiles                    (several tiles in flight, sharded over ranks)
backed by hand-written gfx950 kernels in libs2p_hip.so (C ABI: include/s2p_hip.h).
Importing this package does not touch the GPU.
"""
__version__ = "0.1.0"
