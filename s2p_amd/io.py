"""s2p_amd/io.py -- raster file I/O of the shim: the data formats either side of the hot path
(float32 TIFF with NaN = invalid, 0/1 uint8 PNG masks).  The reference uses rasterio
(s2p/common.py:104-156); it is used here when installed, PIL otherwise."""
import os

import numpy as np

try:
    import rasterio
    import warnings
    warnings.filterwarnings("ignore", category=rasterio.errors.NotGeoreferencedWarning)
    HAVE_RASTERIO = True
except Exception:
    rasterio = None
    HAVE_RASTERIO = False


def image_size(path):
    """(width, height) without decoding the pixels (s2p/block_matching.py:63-64)."""
    if HAVE_RASTERIO:
        with rasterio.open(path, "r") as f:
            return f.width, f.height
    from PIL import Image
    with Image.open(path) as im:
        return im.size


def read_image(path, dtype=np.float32):
    """Single-band raster as a C-contiguous 2-D array; nodata -> NaN for float reads
    (s2p/common.py:104-122 rio_read_as_array_with_nans)."""
    if HAVE_RASTERIO:
        with rasterio.open(path, "r") as src:
            a = src.read(1)
            nodata = src.nodatavals[0] if src.nodatavals else None
        a = a.astype(dtype, copy=False)
        if nodata is not None and np.issubdtype(a.dtype, np.floating):
            a[a == nodata] = np.nan
        return np.ascontiguousarray(a)
    from PIL import Image
    Image.MAX_IMAGE_PIXELS = None
    with Image.open(path) as im:
        a = np.array(im)
    if a.ndim == 3:
        a = a[:, :, 0]
    return np.ascontiguousarray(a.astype(dtype, copy=False))


def read_window(path, x0, y0, x1, y1):
    """Pixels [y0:y1, x0:x1] of a single-band raster in their native sample type (uint8/uint16/float32
    feed the GPU resampler directly; anything else is converted to float32)."""
    if HAVE_RASTERIO:
        from rasterio.windows import Window
        with rasterio.open(path, "r") as src:
            a = src.read(1, window=Window(x0, y0, x1 - x0, y1 - y0))
            nodata = src.nodatavals[0] if src.nodatavals else None
        if nodata is not None:
            a = a.astype(np.float32)
            a[a == nodata] = np.nan
    else:
        from PIL import Image
        Image.MAX_IMAGE_PIXELS = None
        with Image.open(path) as im:
            a = np.array(im.crop((x0, y0, x1, y1)))
        if a.ndim == 3:
            a = a[:, :, 0]
    if a.dtype not in (np.uint8, np.uint16, np.float32):
        a = a.astype(np.float32)
    return np.ascontiguousarray(a)


def write_image(path, array):
    """float32 -> TIFF, uint8 -> PNG/TIFF by extension (s2p/common.py:125-156 rasterio_write)."""
    ext = os.path.splitext(path)[1].lower()
    if ext not in (".tif", ".tiff", ".png"):
        raise NotImplementedError("format {} not supported".format(ext))
    a = np.ascontiguousarray(array)
    if HAVE_RASTERIO:
        profile = dict(driver="GTiff" if ext != ".png" else "PNG", count=1, width=a.shape[1],
                       height=a.shape[0], dtype=a.dtype)
        with rasterio.Env():
            with rasterio.open(path, "w", **profile) as dst:
                dst.write(a[None, :, :])
        return
    from PIL import Image
    if a.dtype == np.float32:
        Image.fromarray(a).save(path)
    elif a.dtype == np.uint8:
        # 0/1 masks: zlib level 1 (the default level 6 spends 10x longer on a speckled mask for a few percent of size)
        Image.fromarray(a).save(path, compress_level=1) if ext == ".png" else Image.fromarray(a).save(path)
    elif a.dtype == np.uint16:
        Image.fromarray(a).save(path)
    else:
        raise NotImplementedError("dtype {} not supported".format(a.dtype))


def update_image(template, path, array):
    """Write `array` (float32) to `path` with the metadata of `template`: the reference copies an input file
    and rewrites its band in place (s2p/fusion.py:64-68).  Without rasterio the copy step has no metadata
    worth keeping (PIL drops geo tags on save): a plain float32 TIFF is written."""
    a = np.ascontiguousarray(array, np.float32)
    if HAVE_RASTERIO:
        import shutil
        shutil.copy(template, path)
        with rasterio.open(path, "r+") as f:
            f.write(a[None, :, :])
        return
    write_image(path, a)


def write_images(pairs):
    """Write several (path, array) outputs of one matcher call concurrently: the encoders (libtiff / zlib behind PIL or
    GDAL) release the GIL, so the disparity, confidence and mask files of a tile are compressed side by side."""
    pairs = list(pairs)
    if len(pairs) <= 1:
        for path, a in pairs:
            write_image(path, a)
        return
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=len(pairs)) as ex:
        list(ex.map(lambda pa: write_image(pa[0], pa[1]), pairs))


def read_images(paths, dtype=np.float32):
    """Decode several rasters concurrently (same reason)."""
    paths = list(paths)
    if len(paths) <= 1:
        return [read_image(p, dtype) for p in paths]
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=len(paths)) as ex:
        return list(ex.map(lambda p: read_image(p, dtype), paths))
