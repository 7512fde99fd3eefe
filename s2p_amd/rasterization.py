"""s2p_amd/rasterization.py -- drop-in for the `plyflatten` calls of s2p's DSM step.

s2p rasterises the point clouds of a tile and its neighbours with the external package plyflatten
(s2p/__init__.py:31 `from plyflatten import plyflatten_from_plyfiles_list`, :462-466 the call in plys_to_dsm;
tests/rasterization_test.py:13-28).  The package is a pip dependency (setup.py:52), absent from the reference tree;
this module mirrors its two public functions on top of libs2p_hip.so (s2p_hip_plyflatten_host: counting sort of the
points by cell, then the C code's running float32 mean per cell in input order -- same bits as the CPU code, pinned
on the reference's dsm_40cm.tiff).  Nothing here falls back to a CPU implementation.

    raster, profile = plyflatten_from_plyfiles_list(clouds_list, resolution, radius=0, roi=None, sigma=None)
    raster = plyflatten(cloud, xoff, yoff, resolution, xsize, ysize, radius, sigma)

`profile` carries what s2p hands to rasterio (common.rasterio_write): 'tiled', 'nodata', 'crs', 'transform'.  Without
rasterio / affine / pyproj in this image, 'crs' is the string of the PLY header comment ("epsg:32740", or the proj4
string of a "UTM 40S" comment) and 'transform' the 6 coefficients (a, b, c, d, e, f) of affine.Affine, in its order."""
import re

import numpy as np

from s2p_amd import _lib
from s2p_amd.ply import read_3d_point_cloud_from_ply  # noqa: F401  (re-exported: plyflatten's utils has one too)

def crs_from_ply_comments(comments):
    """The projection comment s2p writes into its clouds (s2p/__init__.py: "projection: CRS <crs>" or the older
    "projection: UTM <zone><N|S>"), as the string rasterio would be given."""
    for c in comments:
        m = re.match(r"\s*projection:\s*CRS\s+(.*\S)\s*$", c)
        if m:
            return m.group(1)
        m = re.match(r"\s*projection:\s*UTM\s+(\d+)([NS])\s*$", c)
        if m:
            return "+proj=utm +zone=%s%s +datum=WGS84 +units=m +no_defs" % (m.group(1), " +south" if m.group(2) == "S" else "")
    raise ValueError("no 'projection:' comment in the PLY header")


def plyflatten(cloud, xoff, yoff, resolution, xsize, ysize, radius, sigma, device=None):
    """plyflatten.plyflatten: rasterise an (n, 2 + nb) float64 cloud (x, y, then the values to average) on the grid
    with upper-left corner (xoff, yoff), square cells of `resolution`, xsize x ysize cells.  Returns (ysize, xsize,
    nb) float32, NaN in the cells no point contributed to."""
    return _lib.plyflatten(cloud, xoff, yoff, resolution, xsize, ysize, radius, sigma, device=device)


def plyflatten_from_plyfiles_list(clouds_list, resolution, radius=0, roi=None, sigma=None, device=None):
    """plyflatten.plyflatten_from_plyfiles_list, the call of s2p's plys_to_dsm (s2p/__init__.py:462-466).

    Args:
        clouds_list: list of PLY paths (x, y first, every further vertex property becomes a raster band)
        resolution: cell size, in the units of x and y
        radius: every point also contributes to the cells within `radius` cells of its own
        roi: (xoff, yoff, xsize, ysize); None = the extent of the clouds on the grid of multiples of `resolution`
        sigma: std-dev of the Gaussian weight on the distance to the cell centre; None = unweighted mean
    Returns:
        raster (ysize, xsize, nbands) float32, profile dict ('tiled', 'nodata', 'crs', 'transform')."""
    full, comments0 = [], None
    for path in clouds_list:
        data, comments = read_3d_point_cloud_from_ply(path)
        if comments0 is None:
            comments0 = comments
        full.append(np.asarray(data, np.float64))
    if not full:
        raise ValueError("plyflatten_from_plyfiles_list: empty list of clouds")
    cloud = np.concatenate(full)
    if roi is not None:
        xoff, yoff, xsize, ysize = roi
    else:
        xx, yy = cloud[:, 0], cloud[:, 1]
        xmin, xmax, ymin, ymax = np.amin(xx), np.amax(xx), np.amin(yy), np.amax(yy)
        xoff = np.floor(xmin / resolution) * resolution
        xsize = int(1 + np.floor((xmax - xoff) / resolution))
        yoff = np.ceil(ymax / resolution) * resolution
        ysize = int(1 - np.floor((ymin - yoff) / resolution))
    sigma = float("inf") if sigma is None else sigma
    raster = plyflatten(np.ascontiguousarray(cloud), xoff, yoff, resolution, int(xsize), int(ysize), radius, sigma, device=device)
    profile = {"tiled": True, "nodata": float("nan"), "crs": crs_from_ply_comments(comments0),
               "transform": (float(resolution), 0.0, float(xoff), 0.0, -float(resolution), float(yoff))}
    return raster, profile


def write_dsm(path, raster, profile):
    """Write one band of a rasterised DSM with its georeferencing, as plys_to_dsm does through
    common.rasterio_write(out_dsm, raster[:, :, 0], profile=profile) (s2p/__init__.py:468-469): with rasterio, through
    rasterio (profile['transform'] becomes an affine.Affine); without it, a float32 TIFF carrying the GeoTIFF tags of
    the reference's own dsm_40cm.tiff -- ModelPixelScale (33550), ModelTiepoint (33922), GDAL_NODATA (42113) and, for
    an "epsg:<code>" CRS, a GeoKeyDirectory (34735) naming the projected CRS."""
    a = np.ascontiguousarray(raster, np.float32)
    if a.ndim == 3:
        a = a[:, :, 0]
    t = profile["transform"]
    try:
        import rasterio
        from rasterio.transform import Affine
        prof = dict(profile, transform=Affine(*t), driver="GTiff", count=1, width=a.shape[1], height=a.shape[0], dtype="float32")
        with rasterio.Env():
            with rasterio.open(path, "w", **prof) as dst:
                dst.write(a[None, :, :])
        return
    except ImportError:
        pass
    from PIL import Image, TiffImagePlugin
    ifd = TiffImagePlugin.ImageFileDirectory_v2()
    ifd[33550] = (float(t[0]), float(-t[4]), 0.0)                                   # ModelPixelScaleTag
    ifd.tagtype[33550] = 12
    ifd[33922] = (0.0, 0.0, 0.0, float(t[2]), float(t[5]), 0.0)                     # ModelTiepointTag
    ifd.tagtype[33922] = 12
    nodata = profile.get("nodata")
    if nodata is not None:
        ifd[42113] = "nan" if nodata != nodata else repr(float(nodata))             # GDAL_NODATA
        ifd.tagtype[42113] = 2
    m = re.match(r"\s*epsg:(\d+)\s*$", str(profile.get("crs", "")), re.I)
    if m:                                                                           # GTModelType projected, pixel-is-area, the CRS
        ifd[34735] = (1, 1, 0, 3, 1024, 0, 1, 1, 1025, 0, 1, 1, 3072, 0, 1, int(m.group(1)))
        ifd.tagtype[34735] = 3
    Image.fromarray(a).save(path, format="TIFF", tiffinfo=ifd)

