"""s2p_amd/triangulation.py -- drop-in for the C call inside s2p.triangulation.disp_to_xyz.

The reference binds lib/disp_to_h.so with ctypes and calls `disp_to_lonlatalt`
(s2p/triangulation.py:117-145; C code c/disp_to_h.c:70-140 + c/rpc.c).  libs2p_hip.so exports the
same symbol with the same argument list, so the reference module works unchanged once its `lib_path`
(s2p/triangulation.py:18-20) points at it.  This module is the same binding for callers that do not
import s2p: `RPCStruct` mirrors the reference class (:23-82) and `disp_to_lonlatalt` is the part of
`disp_to_xyz` before the CRS conversion (which stays with pyproj in the reference, :148-162).
"""
import ctypes

import numpy as np

from s2p_amd import _lib
from s2p_amd import broker
from s2p_amd import ply


class RPCStruct(_lib.RpcStruct):
    """ctypes version of the RPC C struct defined in c/rpc.h (same fields, same order as the reference)."""

    def __init__(self, rpc=None, delta=1.0):
        """rpc: an rpcm.RPCModel-like object (col_offset, row_num, ... attributes), s2p/triangulation.py:47-82."""
        super().__init__()
        if rpc is None:
            return
        self.offset[:] = [rpc.col_offset, rpc.row_offset, rpc.alt_offset]
        self.ioffset[:] = [rpc.lon_offset, rpc.lat_offset, rpc.alt_offset]
        self.scale[:] = [rpc.col_scale, rpc.row_scale, rpc.alt_scale]
        self.iscale[:] = [rpc.lon_scale, rpc.lat_scale, rpc.alt_scale]
        self.inumx[:] = list(rpc.col_num)
        self.idenx[:] = list(rpc.col_den)
        self.inumy[:] = list(rpc.row_num)
        self.ideny[:] = list(rpc.row_den)
        if hasattr(rpc, 'lat_num'):
            self.numx[:] = list(rpc.lon_num)
            self.denx[:] = list(rpc.lon_den)
            self.numy[:] = list(rpc.lat_num)
            self.deny[:] = list(rpc.lat_den)
        else:
            for a in (self.numx, self.denx, self.numy, self.deny):
                a[:] = [np.nan] * 20
        self.delta = delta          # initialization factor for iterative localization


def rpc_from_geotiff_tag(tag, delta=1.0):
    """RPCStruct from the 92 doubles of a GeoTIFF RPCCoefficientTag (50844): ERR_BIAS, ERR_RAND, LINE_OFF, SAMP_OFF,
    LAT_OFF, LONG_OFF, HEIGHT_OFF, LINE_SCALE, SAMP_SCALE, LAT_SCALE, LONG_SCALE, HEIGHT_SCALE, LINE_NUM_COEFF[20],
    LINE_DEN_COEFF[20], SAMP_NUM_COEFF[20], SAMP_DEN_COEFF[20] -- what rpcm reads from the image and RPCStruct.__init__
    copies (s2p/triangulation.py:47-82).  Such a model has no direct (image -> ground) polynomials: they are set to NaN
    and the kernels localise iteratively, as the reference does."""
    v = [float(x) for x in tag]
    if len(v) != 92:
        raise ValueError("RPCCoefficientTag holds 92 doubles, got %d" % len(v))
    r = RPCStruct()
    line_off, samp_off, lat_off, lon_off, h_off, line_sc, samp_sc, lat_sc, lon_sc, h_sc = v[2:12]
    r.offset[:] = [samp_off, line_off, h_off]
    r.ioffset[:] = [lon_off, lat_off, h_off]
    r.scale[:] = [samp_sc, line_sc, h_sc]
    r.iscale[:] = [lon_sc, lat_sc, h_sc]
    r.inumy[:] = v[12:32]
    r.ideny[:] = v[32:52]
    r.inumx[:] = v[52:72]
    r.idenx[:] = v[72:92]
    for a in (r.numx, r.denx, r.numy, r.deny):
        a[:] = [np.nan] * 20
    r.delta = delta
    return r


def disp_to_lonlatalt(rpc1, rpc2, H1, H2, disp, mask_rect, img_bbx, mask_orig, A=None, device=None):
    """
    3-D (lon, lat, alt) map from a disparity map, using RPC camera models (HIP, MI355X).

    Args (as s2p.triangulation.disp_to_xyz, s2p/triangulation.py:85-108, without out_crs):
        rpc1, rpc2: RPCStruct instances, or rpcm.RPCModel-like objects
        H1, H2: 3x3 rectifying homographies
        disp, mask_rect: (h, w) disparity and mask maps
        img_bbx: col_min, col_max, row_min, row_max of the unrectified image domain
        mask_orig: unrectified image validity domain
        A: 3x3 pointing correction for im2

    Returns: lonlatalt (h, w, 3) float64, err (h, w) float32
    """
    r1 = rpc1 if isinstance(rpc1, ctypes.Structure) else RPCStruct(rpc1)
    r2 = rpc2 if isinstance(rpc2, ctypes.Structure) else RPCStruct(rpc2)
    H1 = np.asarray(H1, np.float64)
    H2 = np.asarray(H2, np.float64)
    if A is not None:                                   # apply pointing correction (:113-114)
        H2 = np.dot(H2, np.linalg.inv(A))
    return _disp_to_lonlatalt_arrays(r1, r2, H1, H2, np.ascontiguousarray(disp, np.float32), np.ascontiguousarray(mask_rect, np.float32),
                                     np.asarray(img_bbx, np.float32), np.ascontiguousarray(mask_orig, np.float32), device=device)


@broker.remote()
def _disp_to_lonlatalt_arrays(r1, r2, H1, H2, disp, msk_rect, img_bbx, msk_orig, device=None):
    """The library call of disp_to_lonlatalt on ready-made arguments (what travels to the GPU broker from a Pool worker)."""
    h, w = disp.shape
    hh, ww = msk_orig.shape
    lonlatalt = np.zeros((h, w, 3), np.float64)
    err = np.zeros((h, w), np.float32)
    Ha = np.ascontiguousarray(H1.reshape(9))
    Hb = np.ascontiguousarray(H2.reshape(9))
    bbx = np.asarray(img_bbx, np.float32)
    P = ctypes.c_void_p
    c = _lib.context(device)
    with _lib._held(c):
        _lib.check(_lib.lib().s2p_hip_disp_to_lonlatalt_host(
            c, lonlatalt.ctypes.data_as(P), err.ctypes.data_as(P), disp.ctypes.data_as(P), None,
            msk_rect.ctypes.data_as(P), w, h, msk_orig.ctypes.data_as(P), ww, hh,
            Ha.ctypes.data_as(P), Hb.ctypes.data_as(P), ctypes.byref(r1), ctypes.byref(r2), bbx.ctypes.data_as(P)))
    return lonlatalt, err


def _to_crs(lonlatalt, out_crs):
    """The CRS step of disp_to_xyz / stereo_corresp_to_xyz (s2p/triangulation.py:148-162, 261-270): lon / lat / alt stay as they are
    for None / epsg:4979 / epsg:4326, a WGS 84 UTM zone ("epsg:326xx" / "epsg:327xx", the reference's default output CRS) is computed
    here (s2p_amd/geographiclib.py); any other CRS raises NotImplementedError so that the caller keeps pyproj for it."""
    crs = None if out_crs is None else (str(out_crs).lower() if not hasattr(out_crs, "to_string") else out_crs.to_string().lower())
    if crs in (None, "epsg:4979", "epsg:4326"):
        return lonlatalt
    from s2p_amd import geographiclib
    return geographiclib.lonlatalt_to_utm(lonlatalt, crs)


def disp_to_xyz(rpc1, rpc2, H1, H2, disp, mask_rect, img_bbx, mask_orig, A=None, out_crs=None):
    """
    Compute a 3D coordinates map from a disparity map, using RPC camera models (HIP, MI355X): s2p.triangulation.disp_to_xyz
    (s2p/triangulation.py:85-162), same arguments and return values -- the one import a maintainer swaps so that the
    triangulation step of the orchestrator's Pools goes through the GPU broker instead of binding the C symbol in every worker.

    Returns: xyz (h, w, 3) float64 in `out_crs`, err (h, w) float32.
    """
    lla, err = disp_to_lonlatalt(rpc1, rpc2, H1, H2, disp, mask_rect, img_bbx, mask_orig, A=A)
    return _to_crs(lla, out_crs), err


def stereo_corresp_to_xyz(rpc1, rpc2, pts1, pts2, out_crs=None):
    """s2p.triangulation.stereo_corresp_to_xyz (s2p/triangulation.py:220-272), same arguments: xyz (n, 3) float64, err (n,) float32."""
    lla, err = stereo_corresp_to_lonlatalt(rpc1, rpc2, pts1, pts2)
    return _to_crs(lla.reshape(1, -1, 3), out_crs).reshape(-1, 3), err.reshape(-1, 1)        # (the reference's err is (n, 1), :255)


def height_map(x, y, w, h, rpc1, rpc2, H1, H2, disp, mask, mask_orig, A=None, device=None):
    """
    Altitude map on the grid of the original reference image from a disparity map on the rectified grid (HIP, MI355X):
    s2p.triangulation.height_map (s2p/triangulation.py:346-389), same arguments.

    Args:
        x, y, w, h: rectangular AOI in the original image
        rpc1, rpc2: RPCStruct instances, or rpcm.RPCModel-like objects
        H1, H2: 3x3 rectifying homographies (affine: scipy's affine_transform refuses a projective H1)
        disp, mask: disparity and mask maps on the rectified grid
        mask_orig: unrectified image validity domain
        A: 3x3 pointing correction for im2

    Returns: (h, w) float64 height map
    """
    p = 1                                               # padding of mask_orig against border effects (:367-371)
    lonlatalt, _ = disp_to_lonlatalt(rpc1, rpc2, H1, H2, disp, mask,
                                     img_bbx=(x - p, x + w + 2 * p, y - p, y + h + 2 * p),
                                     mask_orig=np.pad(mask_orig, p, constant_values=1), A=A, device=device)
    heights = lonlatalt[:, :, 2]
    T = np.array([[1.0, 0.0, x], [0.0, 1.0, y], [0.0, 0.0, 1.0]])                  # common.matrix_translation(x, y)
    return _lib.height_transfer(heights, np.dot(np.asarray(H1, np.float64), T), w, h, device=device)


def height_map_to_xyz(heights, rpc, off_x=0, off_y=0, out_crs=None, device=None):
    """
    3-D coordinates map from a height map, using an RPC camera model (HIP, MI355X): s2p.triangulation.height_map_to_xyz
    (s2p/triangulation.py:165-219), same arguments.

    Args:
        heights: path to the height map file (as in the reference), or the (h, w) array itself
        rpc: RPCStruct instance, or rpcm.RPCModel-like object
        off_{x,y}: coordinates of the origin of the crop in the pixel coordinates of the full image
        out_crs: None / "epsg:4979" (lon, lat, alt) or a WGS 84 UTM zone as "epsg:326xx" / "epsg:327xx" (the reference's
            default output CRS, s2p/initialization.py:133-138: s2p_amd/geographiclib.py); any other CRS stays with pyproj
            (NotImplementedError)

    Returns: xyz (h, w, 3) float64; NaN where the height is NaN (the reference leaves lon / lat of those pixels
        uninitialised and relies on their NaN altitude, :196-199)
    """
    if isinstance(heights, (str, bytes)) or hasattr(heights, "__fspath__"):
        from s2p_amd import io as rio
        heights = rio.read_image(heights)
    r = rpc if isinstance(rpc, ctypes.Structure) else RPCStruct(rpc)
    lla = _lib.height_map_to_lonlatalt(r, heights, off_x, off_y, device=device)
    return _to_crs(lla, out_crs)


def stereo_corresp_to_lonlatalt(rpc1, rpc2, pts1, pts2, device=None):
    """3-D (lon, lat, alt) points from keypoint matches (HIP): the C call inside s2p.triangulation.stereo_corresp_to_xyz
    (s2p/triangulation.py:220-258; c/disp_to_h.c:43-67).  pts1, pts2: (n, 2) arrays.  Returns (n, 3) float64, (n,) float32."""
    # the reference wraps the rpcm objects with delta = 0.1 here (s2p/triangulation.py:240-241; 1.0 in disp_to_xyz, :111-112):
    # delta sets the start and the first step of the iterative localisation (c/rpc.c:378-412)
    r1 = rpc1 if isinstance(rpc1, ctypes.Structure) else RPCStruct(rpc1, delta=0.1)
    r2 = rpc2 if isinstance(rpc2, ctypes.Structure) else RPCStruct(rpc2, delta=0.1)
    a = np.ascontiguousarray(pts1, np.float32)
    b = np.ascontiguousarray(pts2, np.float32)
    assert a.shape == b.shape and a.ndim == 2 and a.shape[1] == 2
    return _stereo_corresp_arrays(r1, r2, a, b, device=device)


@broker.remote()
def _stereo_corresp_arrays(r1, r2, a, b, device=None):
    n = len(a)
    lonlatalt = np.zeros((n, 3), np.float64)
    err = np.zeros(n, np.float32)
    P = ctypes.c_void_p
    c = _lib.context(device)
    with _lib._held(c):
        _lib.check(_lib.lib().s2p_hip_stereo_corresp_to_lonlatalt_host(
            c, lonlatalt.ctypes.data_as(P), err.ctypes.data_as(P), a.ctypes.data_as(P), b.ctypes.data_as(P), n,
            ctypes.byref(r1), ctypes.byref(r2)))
    return lonlatalt, err


@broker.remote()
def count_3d_neighbors(xyz, r, p, device=None):
    """Count 3D neighbors of a gridded set of 3D points (HIP); s2p/triangulation.py:275-301, c/disp_to_h.c:152-174."""
    xyz = np.ascontiguousarray(xyz, np.float64)
    h, w, d = xyz.shape
    assert d == 3
    out = np.zeros((h, w), dtype='int32')
    P = ctypes.c_void_p
    c = _lib.context(device)
    with _lib._held(c):
        _lib.check(_lib.lib().s2p_hip_count_3d_neighbors_host(c, out.ctypes.data_as(P), xyz.ctypes.data_as(P), w, h,
                                                               ctypes.c_float(r), int(p)))
    return out


@broker.remote(inplace=("xyz",))
def remove_isolated_3d_points(xyz, r, p, n, q=1, device=None):
    """Discard (in place) isolated (groups of) points in a gridded set of 3D points (HIP);
    s2p/triangulation.py:304-328, c/disp_to_h.c:177-230.  `xyz` must be a C-contiguous float64 (h, w, 3) array
    (the reference silently filters a temporary copy otherwise, :328)."""
    h, w, d = xyz.shape
    assert d == 3, 'expecting a 3-channels image with shape (h, w, 3)'
    assert xyz.dtype == np.float64 and xyz.flags['C_CONTIGUOUS'], 'in-place filter needs a C-contiguous float64 array'
    c = _lib.context(device)
    with _lib._held(c):
        _lib.check(_lib.lib().s2p_hip_remove_isolated_3d_points_host(c, xyz.ctypes.data_as(ctypes.c_void_p), w, h,
                                                                     ctypes.c_float(r), int(p), int(n), int(q)))


def filter_xyz(xyz, r, n, img_gsd, device=None):
    """Discard (in place) points that have less than n points closer than r units; s2p/triangulation.py:331-343."""
    p = np.ceil(r / img_gsd).astype(int)
    remove_isolated_3d_points(xyz, r, p, n, device=device)


def write_to_ply(path_to_ply_file, xyz, colors=None, proj_com='', confidence=''):
    """
    Write a raster of 3D point coordinates as a point cloud in a .ply file: s2p.triangulation.write_to_ply
    (s2p/triangulation.py:392-429), same arguments and the same file, byte for byte (s2p_amd/ply.py).

    Args:
        path_to_ply_file (str): path to a .ply file
        xyz (array): (h, w, 3) x, y, z per pixel; pixels with a non-finite coordinate are dropped
        colors (np.array): (channels, h, w) colour image, optional
        proj_com (str): projection comment of the .ply header
        confidence (str): path to a confidence map image, optional
    """
    xyz_list = xyz.reshape(-1, 3)
    valid = np.all(np.isfinite(xyz_list), axis=1)
    colors_list = colors.transpose(1, 2, 0).reshape(-1, colors.shape[0])[valid] if colors is not None else None
    if confidence != '':
        from s2p_amd import io as rio
        extra_list = rio.read_image(confidence, np.float32).flatten()[valid].astype(np.float32)
        extra_names = ['confidence']
    else:
        extra_list, extra_names = None, None
    ply.write_3d_point_cloud_to_ply(path_to_ply_file, xyz_list[valid], colors=colors_list, extra_properties=extra_list,
                                    extra_properties_names=extra_names,
                                    comments=["created by S2P", "projection: {}".format(proj_com)])
