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


# ---- fast paths for the two formats the hot path's boundary actually carries: uncompressed single-band TIFF
# (float32 / uint16 / uint8: what `homography`, `mgm`, `sgbm` and rasterio's default GTiff profile write) and 8-bit grayscale PNG
# masks.  A 1024 x 1024 tile is 4 MB of raw samples: parsing the IFD and reading the strips directly costs 0.3 ms where a generic
# decoder takes 2-3; anything these few lines do not recognise (compression, tiles, several bands, nodata tags, BigTIFF) falls
# back to rasterio / PIL.
_TIFF_DTYPES = {(1, 8): np.uint8, (1, 16): np.uint16, (3, 32): np.float32, (2, 16): np.int16, (1, 32): np.uint32, (2, 32): np.int32, (3, 64): np.float64}


def _tiff_fast_read(path, alloc=None):
    import struct
    with open(path, "rb") as f:
        head = f.read(8)
        if len(head) < 8 or head[:2] not in (b"II", b"MM"):
            return None
        e = "<" if head[:2] == b"II" else ">"
        if struct.unpack(e + "H", head[2:4])[0] != 42:
            return None
        f.seek(struct.unpack(e + "I", head[4:8])[0])
        n = struct.unpack(e + "H", f.read(2))[0]
        raw = f.read(12 * n + 4)
        if struct.unpack(e + "I", raw[12 * n:])[0] != 0:
            return None                                    # more than one image in the file
        tags = {}
        sizes = {1: 1, 2: 1, 3: 2, 4: 4, 5: 8, 16: 8}
        codes = {1: "B", 3: "H", 4: "I", 16: "Q"}
        for i in range(n):
            tag, typ, cnt, val = struct.unpack(e + "HHI4s", raw[12 * i:12 * i + 12])
            if typ not in codes:
                if tag in (256, 257, 258, 259, 273, 277, 278, 279, 284, 339, 322, 323, 324, 325, 42113):
                    return None
                continue
            nbytes = sizes[typ] * cnt
            if nbytes <= 4:
                data = val[:nbytes]
            else:
                pos = f.tell()
                f.seek(struct.unpack(e + "I", val)[0])
                data = f.read(nbytes)
                f.seek(pos)
            tags[tag] = struct.unpack(e + codes[typ] * cnt, data)
        if 322 in tags or 323 in tags or 42113 in tags:    # tiled, or a nodata value to honour
            return None
        if tags.get(259, (1,))[0] != 1 or tags.get(277, (1,))[0] != 1 or tags.get(284, (1,))[0] != 1:
            return None
        if 256 not in tags or 257 not in tags or 273 not in tags:
            return None
        w, h = tags[256][0], tags[257][0]
        key = (tags.get(339, (1,))[0], tags.get(258, (1,))[0])
        if key not in _TIFF_DTYPES:
            return None
        dt = np.dtype(_TIFF_DTYPES[key]).newbyteorder(e)
        offs = tags[273]
        cnts = tags.get(279) or (w * h * dt.itemsize,)
        if len(offs) != len(cnts) or sum(cnts) != w * h * dt.itemsize:
            return None
        if all(offs[i] + cnts[i] == offs[i + 1] for i in range(len(offs) - 1)):
            f.seek(offs[0])
            if alloc is not None and dt.isnative:           # straight into the caller's kind of memory (page-locked: _lib.pinned_empty)
                a = alloc((w * h,), dt)
                if f.readinto(memoryview(a).cast("B")) != a.nbytes:
                    return None
            else:
                a = np.fromfile(f, dt, w * h)
        else:
            a = alloc((w * h,), dt) if (alloc is not None and dt.isnative) else np.empty(w * h, dt)   # the file's byte order stays with the array
            buf = a.view(np.uint8)
            o = 0
            for off, c in zip(offs, cnts):
                f.seek(off)
                buf[o:o + c] = np.frombuffer(f.read(c), np.uint8)
                o += c
        if a.size != w * h:
            return None
        return a.reshape(h, w)


def _tiff_fast_write(path, a):
    """Classic little-endian TIFF, one uncompressed strip, the tags every reader needs (and no others)."""
    import struct
    fmt = {np.dtype(np.float32): (3, 32), np.dtype(np.uint8): (1, 8), np.dtype(np.uint16): (1, 16)}[a.dtype]
    h, w = a.shape
    a = np.ascontiguousarray(a.astype(a.dtype.newbyteorder("<"), copy=False))
    entries = [(256, 4, w), (257, 4, h), (258, 3, fmt[1]), (259, 3, 1), (262, 3, 1), (273, 4, 8 + 2 + 12 * 10 + 4), (277, 3, 1),
               (278, 4, h), (279, 4, a.nbytes), (339, 3, fmt[0])]
    ifd = struct.pack("<H", len(entries)) + b"".join(struct.pack("<HHI", t, ty, 1) + (struct.pack("<H", v) + b"\0\0" if ty == 3 else struct.pack("<I", v))
                                                     for t, ty, v in entries) + struct.pack("<I", 0)
    with open(path, "wb") as f:
        f.write(b"II" + struct.pack("<HI", 42, 8) + ifd)
        f.write(memoryview(a).cast("B"))


_POOLS = {}


def _pool(which=0):
    """Thread pools for the encoders of this process (zlib and file writes release the GIL): pool 0 runs whole-file tasks
    (write_images, read_images), pool 1 the deflate pieces those tasks fan out -- a task never waits on its own pool.
    Keyed by pid: the orchestrator forks its workers (s2p/parallel.py), and a pool inherited through fork has no threads."""
    key = (os.getpid(), which)
    if key not in _POOLS:
        from concurrent.futures import ThreadPoolExecutor
        for k in [k for k in _POOLS if k[0] != key[0]]:
            del _POOLS[k]
        _POOLS[key] = ThreadPoolExecutor(max_workers=8)
    return _POOLS[key]


def _deflate_chunks(buf, parts=4, level=1):
    """zlib stream of `buf` built from `parts` raw-deflate pieces compressed side by side (each ends on a sync flush, the
    last one finishes the stream; adler32 of the whole runs beside them): the way pigz does it, any inflater reads it."""
    import struct
    import zlib
    n = len(buf)
    parts = max(1, min(parts, n // 65536))
    cuts = [n * i // parts for i in range(parts + 1)]

    def piece(i):
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        return c.compress(buf[cuts[i]:cuts[i + 1]]) + c.flush(zlib.Z_FINISH if i == parts - 1 else zlib.Z_SYNC_FLUSH)
    if parts == 1:
        return zlib.compress(buf, level)
    pool = _pool(1)                               # NOT the pool write_images runs write_image on: its workers would wait for tasks queued behind them
    futs = [pool.submit(piece, i) for i in range(1, parts)] + [pool.submit(zlib.adler32, buf)]
    first = piece(0)
    outs = [f.result() for f in futs]
    return b"\x78\x01" + first + b"".join(outs[:-1]) + struct.pack(">I", outs[-1] & 0xffffffff)


def _png_fast_write(path, a):
    """8-bit grayscale PNG, filter 0 on every row, zlib level 1: a 0 / 1 mask needs no better, and the generic encoder
    spends most of its time choosing filters."""
    import struct
    import zlib
    h, w = a.shape
    rows = np.zeros((h, w + 1), np.uint8)
    rows[:, 1:] = a

    def chunk(kind, data):
        return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data) & 0xffffffff)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0))
                + chunk(b"IDAT", _deflate_chunks(memoryview(rows).cast("B"))) + chunk(b"IEND", b""))


def image_size(path):
    """(width, height) without decoding the pixels (s2p/block_matching.py:63-64)."""
    if HAVE_RASTERIO:
        with rasterio.open(path, "r") as f:
            return f.width, f.height
    from PIL import Image
    with Image.open(path) as im:
        return im.size


def read_image(path, dtype=np.float32, alloc=None):
    """Single-band raster as a C-contiguous 2-D array; nodata -> NaN for float reads
    (s2p/common.py:104-122 rio_read_as_array_with_nans).  alloc(shape, dtype): allocator of the result for the files read
    in place (the shim passes _lib.pinned_empty: the samples land in page-locked memory, ready for DMA)."""
    if os.path.splitext(path)[1].lower() in (".tif", ".tiff"):
        try:
            a = _tiff_fast_read(path, alloc)
        except Exception:
            a = None
        if a is not None:
            return np.ascontiguousarray(a.astype(dtype, copy=False))
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
    if a.ndim == 2 and a.size > 0:
        if ext == ".png" and a.dtype == np.uint8:
            return _png_fast_write(path, a)
        if ext != ".png" and a.dtype in (np.dtype(np.float32), np.dtype(np.uint8), np.dtype(np.uint16)) and a.nbytes < 2 ** 32 - 4096:
            return _tiff_fast_write(path, a)      # classic TIFF: 32-bit offsets; larger rasters go to the generic writer (BigTIFF)
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
    """Write several (path, array) outputs of one matcher call concurrently: the encoders (zlib, libtiff / GDAL) and the
    file writes release the GIL, so the disparity, confidence and mask files of a tile are produced side by side."""
    pairs = list(pairs)
    if len(pairs) <= 1:
        for path, a in pairs:
            write_image(path, a)
        return
    pool = _pool()
    futs = [pool.submit(write_image, path, a) for path, a in pairs[:-1]]
    write_image(*pairs[-1])                      # the caller's thread takes the last one (the mask, whose encoder fans out itself)
    for f in futs:
        f.result()


def read_images(paths, dtype=np.float32, alloc=None):
    """Several rasters of one call.  Plain uncompressed TIFFs are read in place (0.1 ms each: a thread costs more than
    it hides); files that need a real decoder are decoded concurrently."""
    paths = list(paths)
    out = [None] * len(paths)
    slow = []

    def fast(i):
        p = paths[i]
        if os.path.splitext(p)[1].lower() in (".tif", ".tiff"):
            try:
                a = _tiff_fast_read(p, alloc)
            except Exception:
                a = None
            if a is not None:
                out[i] = np.ascontiguousarray(a.astype(dtype, copy=False))
                return True
        return False
    big = [i for i, p in enumerate(paths) if os.path.exists(p) and os.path.getsize(p) >= (1 << 20)]
    futs = {}
    if len(big) >= 2:                            # megabytes per file: the copies out of the page cache run side by side (readinto releases the GIL)
        futs = {i: _pool().submit(fast, i) for i in big[1:]}
    for i in range(len(paths)):
        ok = futs[i].result() if i in futs else fast(i)
        if not ok:
            slow.append(i)
    if len(slow) == 1:
        out[slow[0]] = read_image(paths[slow[0]], dtype)
    elif slow:
        for i, a in zip(slow, _pool().map(lambda p: read_image(p, dtype), [paths[i] for i in slow])):
            out[i] = a
    return out
