"""oracle/pyoracle.py -- TEST INFRASTRUCTURE: ctypes loaders for the CPU checkers.

* ``liboracle.so``          -- our own CPU restatement (oracle/*.c), always buildable (gcc).
* ``_ref/libsgbm_ref.so``   -- the real reference matcher built from /root/reference/3rdparty/sgbm
                               (oracle/Makefile `ref`); exists only when it was built in the
                               container that has /root/reference (the prebuilt .so then travels to
                               the GPU box).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liboracle.so")
REF_SO = os.path.join(HERE, "_ref", "libsgbm_ref.so")


class Dump(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in
                ("q1", "q2", "C", "S", "disp_raw", "disp_med", "disp_fin", "cost_raw")] + \
               [("geom", ctypes.c_int * 8), ("rminmax", ctypes.c_float * 2)]


def build(ref=None):
    """Compile liboracle.so (and the reference library when /root/reference is present)."""
    subprocess.run(["make", "-s", "-C", HERE, "oracle"], check=True)
    if ref is None:
        ref = os.path.isdir("/root/reference/3rdparty/sgbm")
    if ref:
        subprocess.run(["make", "-s", "-j8", "-C", HERE, "ref"], check=True)
        if os.path.isdir("/root/reference/c"):
            subprocess.run(["make", "-s", "-C", HERE, "ref_tri"], check=True)


_libs = {}


def _load(path):
    if path not in _libs:
        _libs[path] = ctypes.CDLL(path)
    return _libs[path]


def have_ref():
    return os.path.exists(REF_SO)


def oracle_lib():
    if not os.path.exists(ORACLE_SO):
        build(ref=False)
    return _load(ORACLE_SO)


def ref_lib():
    return _load(REF_SO)


def sgbm_geometry(w, dmin, dmax):
    """Canvas geometry of the sgbm driver (sgbm.cpp:166-207, stereosgbm.cpp:328-337)."""
    maxdisp, mindisp = -dmin, -dmax
    ndisp = 16 * int(np.ceil((maxdisp - mindisp) / 16.0))
    x0 = max(maxdisp, 0)
    Wc = w + max(-mindisp, 0) + max(maxdisp, 0)
    minD, maxD = mindisp, mindisp + ndisp
    minX1, maxX1 = max(-maxD, 0), Wc + min(minD, 0)
    return dict(Wc=Wc, width1=maxX1 - minX1, D=ndisp, minD=minD, x0=x0, minX1=minX1, maxX1=maxX1,
                invalid=(minD - 1) * 16)


def _run(fn, im1, im2, dmin, dmax, win, P1, P2, lr, dump):
    im1 = np.ascontiguousarray(im1, np.float32)
    im2 = np.ascontiguousarray(im2, np.float32)
    h, w = im1.shape
    od = np.empty((h, w), np.float32)
    oc = np.empty((h, w), np.float32)
    out = {}
    dptr = None
    if dump:
        g = sgbm_geometry(w, dmin, dmax)
        d = Dump()
        arrs = dict(q1=np.zeros((h, w), np.uint8), q2=np.zeros((h, w), np.uint8),
                    disp_raw=np.zeros((h, g["Wc"]), np.int16), disp_med=np.zeros((h, g["Wc"]), np.int16),
                    disp_fin=np.zeros((h, g["Wc"]), np.int16), cost_raw=np.zeros((h, g["Wc"]), np.int16))
        if dump == "full" and g["width1"] > 0:
            arrs["C"] = np.zeros((h, g["width1"], g["D"]), np.int16)
            arrs["S"] = np.zeros((h, g["width1"], g["D"]), np.int16)
        for k, a in arrs.items():
            setattr(d, k, a.ctypes.data)
        dptr = ctypes.byref(d)
        out.update(arrs)
    fn.restype = ctypes.c_int
    rc = fn(im1.ctypes.data_as(ctypes.c_void_p), im2.ctypes.data_as(ctypes.c_void_p),
            ctypes.c_int(w), ctypes.c_int(h), ctypes.c_int(dmin), ctypes.c_int(dmax),
            ctypes.c_int(win), ctypes.c_int(P1), ctypes.c_int(P2), ctypes.c_int(lr),
            od.ctypes.data_as(ctypes.c_void_p), oc.ctypes.data_as(ctypes.c_void_p), dptr)
    out.update(rc=rc, disp=od, cost=oc)
    if dump:
        out["geom"] = list(d.geom)
        out["rminmax"] = list(d.rminmax)
    return out


def oracle_sgbm(im1, im2, dmin, dmax, win=3, P1=8, P2=32, lr=1, dump=False):
    """Our CPU restatement; dump in (False, True, 'full')."""
    return _run(oracle_lib().s2p_oracle_sgbm, im1, im2, dmin, dmax, win, P1, P2, lr, dump)


def ref_sgbm(im1, im2, dmin, dmax, win=3, P1=8, P2=32, lr=1, dump=False):
    """The real reference (only where oracle/_ref/libsgbm_ref.so exists)."""
    return _run(ref_lib().s2p_ref_sgbm, im1, im2, dmin, dmax, win, P1, P2, lr, dump)


def oracle_rejection_mask(disp, im1, im2):
    disp = np.ascontiguousarray(disp, np.float32)
    im1 = np.ascontiguousarray(im1, np.float32)
    im2 = np.ascontiguousarray(im2, np.float32)
    h, w = disp.shape
    m = np.zeros((h, w), np.uint8)
    oracle_lib().s2p_oracle_rejection_mask(
        disp.ctypes.data_as(ctypes.c_void_p), im1.ctypes.data_as(ctypes.c_void_p),
        im2.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(w), ctypes.c_int(h),
        m.ctypes.data_as(ctypes.c_void_p))
    return m


def set_alias_oob(flag):
    """1 (default): reproduce the reference's out-of-bounds disp2 aliasing bit-for-bit;
    0: 'padded' semantics (the out-of-bounds store has no side effect). See sgbm_oracle.c."""
    ctypes.c_int.in_dll(oracle_lib(), "s2p_oracle_alias_oob").value = int(flag)


def oob_count():
    return ctypes.c_long.in_dll(oracle_lib(), "s2p_oracle_oob_count").value


class CensusParams(ctypes.Structure):
    _fields_ = [("census_win", ctypes.c_int), ("P1", ctypes.c_int), ("P2", ctypes.c_int), ("nb_dir", ctypes.c_int),
                ("lr_check", ctypes.c_int), ("lr_tau", ctypes.c_float), ("mindiff", ctypes.c_int),
                ("median", ctypes.c_int), ("remove_small_cc", ctypes.c_int), ("fix_overcount", ctypes.c_int),
                ("recursion", ctypes.c_int), ("scales", ctypes.c_int), ("subpix", ctypes.c_int), ("cost", ctypes.c_int),
                ("subpix_model", ctypes.c_int)]


class CensusDump(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("C", "S", "disp_raw", "disp_med")] + [("dmin0", ctypes.c_int), ("D0", ctypes.c_int)]


def census_params(**kw):
    p = CensusParams(5, 8, 32, 8, 1, 1.0, -1, 1, 0, 1, 0, 1, 1, 0, 0)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def oracle_census_sgm(im1, im2, dmin, dmax, params=None, dump=False):
    """CPU statement of the census / 8-path SGM matcher ('mgm' stand-in); range inclusive."""
    im1 = np.ascontiguousarray(im1, np.float32)
    im2 = np.ascontiguousarray(im2, np.float32)
    h, w = im1.shape
    p = params or census_params()
    D = ((2 if p.subpix == 2 else 1) * (dmax - dmin) + 1 + 15) // 16 * 16
    out = dict(disp=np.empty((h, w), np.float32), conf=np.empty((h, w), np.float32), mask=np.empty((h, w), np.uint8))
    dptr = None
    if dump:
        d = CensusDump()
        arrs = dict(disp_raw=np.zeros((h, w), np.float32), disp_med=np.zeros((h, w), np.float32))
        if dump == "full":
            arrs["C"] = np.zeros((h, w, D), np.uint8)
            arrs["S"] = np.zeros((h, w, D), np.uint16)
        for k, a in arrs.items():
            setattr(d, k, a.ctypes.data)
        dptr = ctypes.byref(d)
        out.update(arrs)
    fn = oracle_lib().s2p_oracle_census_sgm
    fn.restype = ctypes.c_int
    out["rc"] = fn(im1.ctypes.data_as(ctypes.c_void_p), im2.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(w),
                   ctypes.c_int(h), ctypes.c_int(dmin), ctypes.c_int(dmax), ctypes.byref(p),
                   out["disp"].ctypes.data_as(ctypes.c_void_p), out["conf"].ctypes.data_as(ctypes.c_void_p),
                   out["mask"].ctypes.data_as(ctypes.c_void_p), dptr)
    if dump:                                   # C / S are laid out [h][w][D0] from dmin0 (the finest level of a multi-scale call is narrowed)
        out["dmin0"], out["D0"] = int(d.dmin0), int(d.D0)
        for k in ("C", "S"):
            if k in out and d.D0 > 0 and d.D0 != D:
                out[k] = out[k].reshape(-1)[:h * w * d.D0].reshape(h, w, d.D0)
    return out


def oracle_warp(src, H, w, h):
    """CPU statement of `homography src -h H out w h` (quintic B-spline); src any real dtype."""
    src = np.ascontiguousarray(src, np.float32)
    H = np.ascontiguousarray(H, np.float64).reshape(9)
    sh, sw = src.shape
    out = np.empty((h, w), np.float32)
    fn = oracle_lib().s2p_oracle_warp_homography
    fn.restype = ctypes.c_int
    rc = fn(src.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(sw), ctypes.c_int(sh), H.ctypes.data_as(ctypes.c_void_p),
            out.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(w), ctypes.c_int(h))
    if rc:
        raise ValueError("singular homography")
    return out


def oracle_erode(mask, radius):
    mask = np.ascontiguousarray(mask, np.uint8)
    h, w = mask.shape
    out = np.empty_like(mask)
    oracle_lib().s2p_oracle_erode_disk(mask.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(w), ctypes.c_int(h),
                                       ctypes.c_int(int(radius)), out.ctypes.data_as(ctypes.c_void_p))
    return out


# ---- triangulation (c/disp_to_h.c, c/rpc.c) ---------------------------------------------------------------
REF_TRI_SO = os.path.join(HERE, "_ref", "libdisp_to_h_ref.so")


class RPC(ctypes.Structure):
    """struct rpc of c/rpc.h:13-31 (same layout as s2p/triangulation.py:23-45 RPCStruct)."""
    _fields_ = [("numx", ctypes.c_double * 20), ("denx", ctypes.c_double * 20), ("numy", ctypes.c_double * 20),
                ("deny", ctypes.c_double * 20), ("scale", ctypes.c_double * 3), ("offset", ctypes.c_double * 3),
                ("inumx", ctypes.c_double * 20), ("idenx", ctypes.c_double * 20), ("inumy", ctypes.c_double * 20),
                ("ideny", ctypes.c_double * 20), ("iscale", ctypes.c_double * 3), ("ioffset", ctypes.c_double * 3),
                ("dmval", ctypes.c_double * 4), ("imval", ctypes.c_double * 4), ("delta", ctypes.c_double)]


def rpc_from_geotiff_tag(v, delta=1.0):
    """RPCCoefficientTag (50844, 92 doubles: ERR_BIAS, ERR_RAND, LINE_OFF, SAMP_OFF, LAT_OFF, LONG_OFF, HEIGHT_OFF,
    LINE_SCALE, SAMP_SCALE, LAT_SCALE, LONG_SCALE, HEIGHT_SCALE, LINE_NUM[20], LINE_DEN[20], SAMP_NUM[20],
    SAMP_DEN[20]) -> the C struct, filled the way s2p/triangulation.py:47-82 does from an rpcm model that
    has no direct (lat/lon) polynomials: numx.. = NaN, so the C code localises iteratively."""
    v = [float(x) for x in v]
    r = RPC()
    line_off, samp_off, lat_off, lon_off, h_off, line_sc, samp_sc, lat_sc, lon_sc, h_sc = v[2:12]
    r.offset[:] = [samp_off, line_off, h_off]
    r.ioffset[:] = [lon_off, lat_off, h_off]
    r.scale[:] = [samp_sc, line_sc, h_sc]
    r.iscale[:] = [lon_sc, lat_sc, h_sc]
    r.inumy[:] = v[12:32]; r.ideny[:] = v[32:52]; r.inumx[:] = v[52:72]; r.idenx[:] = v[72:92]
    for a in (r.numx, r.denx, r.numy, r.deny):
        a[:] = [float("nan")] * 20
    r.delta = delta
    return r


def have_ref_tri():
    return os.path.exists(REF_TRI_SO)


def _tri(fn, rpc1, rpc2, H1, H2, disp, mask_rect, img_bbx, mask_orig):
    disp = np.ascontiguousarray(disp, np.float32)
    h, w = disp.shape
    dispy = np.zeros((h, w), np.float32)
    msk = np.ascontiguousarray(mask_rect, np.float32)
    mo = np.ascontiguousarray(mask_orig, np.float32)
    hh, ww = mo.shape
    lla = np.zeros((h, w, 3), np.float64)
    err = np.zeros((h, w), np.float32)
    Ha = np.ascontiguousarray(np.asarray(H1, np.float64).reshape(9))
    Hb = np.ascontiguousarray(np.asarray(H2, np.float64).reshape(9))
    bb = np.asarray(img_bbx, np.float32)
    P = ctypes.c_void_p
    fn.restype = None
    fn(lla.ctypes.data_as(P), err.ctypes.data_as(P), disp.ctypes.data_as(P), dispy.ctypes.data_as(P),
       msk.ctypes.data_as(P), ctypes.c_int(w), ctypes.c_int(h), mo.ctypes.data_as(P), ctypes.c_int(ww), ctypes.c_int(hh),
       Ha.ctypes.data_as(P), Hb.ctypes.data_as(P), ctypes.byref(rpc1), ctypes.byref(rpc2), bb.ctypes.data_as(P))
    return lla, err


def ref_disp_to_lonlatalt(rpc1, rpc2, H1, H2, disp, mask_rect, img_bbx, mask_orig):
    """The reference's own disp_to_lonlatalt (oracle/Makefile ref_tri), c/disp_to_h.c:70-140."""
    return _tri(_load(REF_TRI_SO).disp_to_lonlatalt, rpc1, rpc2, H1, H2, disp, mask_rect, img_bbx, mask_orig)


def oracle_disp_to_lonlatalt(rpc1, rpc2, H1, H2, disp, mask_rect, img_bbx, mask_orig):
    """Our C restatement (oracle/triangulation_oracle.c)."""
    return _tri(oracle_lib().s2p_oracle_disp_to_lonlatalt, rpc1, rpc2, H1, H2, disp, mask_rect, img_bbx, mask_orig)


def _corresp(fn, rpc1, rpc2, pts1, pts2):
    a = np.ascontiguousarray(pts1, np.float32)
    b = np.ascontiguousarray(pts2, np.float32)
    n = len(a)
    lla = np.zeros((n, 3), np.float64)
    err = np.zeros(n, np.float32)
    P = ctypes.c_void_p
    fn.restype = None
    fn(lla.ctypes.data_as(P), err.ctypes.data_as(P), a.ctypes.data_as(P), b.ctypes.data_as(P), ctypes.c_int(n),
       ctypes.byref(rpc1), ctypes.byref(rpc2))
    return lla, err


def ref_stereo_corresp_to_lonlatalt(rpc1, rpc2, pts1, pts2):
    """The reference's own stereo_corresp_to_lonlatalt (c/disp_to_h.c:43-67, oracle/Makefile ref_tri)."""
    return _corresp(_load(REF_TRI_SO).stereo_corresp_to_lonlatalt, rpc1, rpc2, pts1, pts2)


def oracle_stereo_corresp_to_lonlatalt(rpc1, rpc2, pts1, pts2):
    return _corresp(oracle_lib().s2p_oracle_stereo_corresp_to_lonlatalt, rpc1, rpc2, pts1, pts2)


def _count3d(fn, xyz, r, p):
    xyz = np.ascontiguousarray(xyz, np.float64)
    h, w, _ = xyz.shape
    out = np.zeros((h, w), np.int32)
    fn.restype = None
    fn(out.ctypes.data_as(ctypes.c_void_p), xyz.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(w), ctypes.c_int(h),
       ctypes.c_float(r), ctypes.c_int(int(p)))
    return out


def _remove3d(fn, xyz, r, p, n, q):
    out = np.array(xyz, np.float64, order="C", copy=True)
    h, w, _ = out.shape
    fn.restype = None
    fn(out.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(w), ctypes.c_int(h), ctypes.c_float(r), ctypes.c_int(int(p)),
       ctypes.c_int(int(n)), ctypes.c_int(int(q)))
    return out


def ref_count_3d_neighbors(xyz, r, p):
    """The reference's own count_3d_neighbors (c/disp_to_h.c:152-174)."""
    return _count3d(_load(REF_TRI_SO).count_3d_neighbors, xyz, r, p)


def oracle_count_3d_neighbors(xyz, r, p):
    return _count3d(oracle_lib().s2p_oracle_count_3d_neighbors, xyz, r, p)


def ref_remove_isolated_3d_points(xyz, r, p, n, q=1):
    """The reference's own remove_isolated_3d_points (c/disp_to_h.c:177-230); returns a filtered copy."""
    return _remove3d(_load(REF_TRI_SO).remove_isolated_3d_points, xyz, r, p, n, q)


def oracle_remove_isolated_3d_points(xyz, r, p, n, q=1):
    return _remove3d(oracle_lib().s2p_oracle_remove_isolated_3d_points, xyz, r, p, n, q)


# ---- fusion.merge_n (s2p/fusion.py:16-68) -------------------------------------------------------------------
def average_if_close(x, threshold):
    """s2p/fusion.py:16-23, restated."""
    if np.nanmax(x) - np.nanmin(x) > threshold:
        return np.nan
    return np.nanmedian(x)


def oracle_merge_n(images, offsets, averaging="average_if_close", threshold=1, fn=None):
    """The array core of fusion.merge_n (s2p/fusion.py:46-68) with numpy itself, as the reference runs it:
    float64 (h, w, n) stack of image - offset (:46-49), np.apply_along_axis of the operator over the last
    axis (:54-58), + mean(offsets) (:61), cast to float32 (:68).  `fn` substitutes the per-pixel function
    (tests/golden/make_golden.py passes the reference's own average_if_close)."""
    import warnings
    h, w = images[0].shape
    x = np.empty((h, w, len(images)))
    for i, img in enumerate(images):
        # the call site passes np.loadtxt scalars = 0-d float64 arrays (s2p/__init__.py:372-381): under numpy >= 2
        # promotion rules float32 image - float64 0-d array is a float64 subtraction (numpy 1.x kept float32)
        x[:, :, i] = np.asarray(img, np.float32) - np.asarray(offsets[i], np.float64).reshape(())
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")          # all-NaN slices: numpy warns and returns NaN, as in the reference
        if averaging.startswith(("np.", "numpy.")):
            avg = np.apply_along_axis(getattr(np, averaging.split(".")[1]), axis=2, arr=x)
        elif averaging == "average_if_close":
            avg = np.apply_along_axis(fn or average_if_close, 2, x, threshold)
        else:
            raise ValueError(averaging)
    avg = avg + np.mean(offsets)
    return avg.astype("float32")


def oracle_plyflatten(cloud, xoff, yoff, resolution, xsize, ysize, radius=0, sigma=float("inf")):
    """`plyflatten.plyflatten` (the array-level call behind s2p/__init__.py:462-466), restated in rasterize_oracle.c.
    cloud: (n, 2 + nb) float64 rows x, y, values; returns (ysize, xsize, nb) float32."""
    c = np.ascontiguousarray(cloud, np.float64)
    nb = c.shape[1] - 2
    out = np.empty((ysize, xsize, nb), np.float32)
    fn = oracle_lib().s2p_oracle_plyflatten
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                   ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_void_p]
    rc = fn(c.ctypes.data, c.shape[0], nb, xoff, yoff, resolution, xsize, ysize, int(radius), float(sigma), out.ctypes.data)
    if rc:
        raise ValueError("s2p_oracle_plyflatten: status %d" % rc)
    return out


def oracle_height_transfer(height_map, H, w, h):
    """The resampling half of triangulation.height_map, restated with the reference's own scipy calls
    (s2p/triangulation.py:376-389): order-1 affine_transform of nan_to_num(height_map).T, order-0 transform of the NaN
    mask, 3x3 binary dilation, NaN where set.  scipy (a dependency of the reference) is importable beside the tests."""
    from scipy import ndimage
    height_map = np.asarray(height_map, np.float64)
    out = ndimage.affine_transform(np.nan_to_num(height_map).T, H, output_shape=(w, h), order=1).T
    if np.isnan(height_map).any():
        i = ndimage.affine_transform(np.isnan(height_map).T, H, output_shape=(w, h), order=0).T
        i = ndimage.binary_dilation(i, structure=np.ones((3, 3)))
        out[i] = np.nan
    return out


def oracle_height_map_to_lonlatalt(rpc, heights, off_x=0, off_y=0):
    """The localisation inside triangulation.height_map_to_xyz (s2p/triangulation.py:165-219) on the RPC code of
    c/rpc.c as restated in triangulation_oracle.c: (h, w) float32 heights -> (h, w, 3) float64 lon, lat, alt."""
    a = np.ascontiguousarray(heights, np.float32)
    h, w = a.shape
    out = np.empty((h, w, 3), np.float64)
    fn = oracle_lib().s2p_oracle_height_map_to_lonlatalt
    fn.restype = None
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    fn(ctypes.addressof(rpc), a.ctypes.data, w, h, int(off_x), int(off_y), out.ctypes.data)
    return out


def oracle_cargarse_basura(height_map):
    """common.cargarse_basura (s2p/common.py:224-235) on an array: cleanup_oracle.c."""
    a = np.ascontiguousarray(height_map, np.float32)
    h, w = a.shape
    out = np.empty_like(a)
    fn = oracle_lib().s2p_oracle_cargarse_basura
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    if fn(a.ctypes.data, w, h, out.ctypes.data):
        raise MemoryError("s2p_oracle_cargarse_basura")
    return out
