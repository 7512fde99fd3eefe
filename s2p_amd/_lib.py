"""s2p_amd/_lib.py -- ctypes binding of libs2p_hip.so (include/s2p_hip.h).

Mirrors how the reference binds its own C libraries (s2p/triangulation.py:18-20,117-145):
ctypes.CDLL on <pkg>/lib/<name>.so, numpy ndpointer argtypes, caller-allocated outputs.
There is NO CPU fallback: if the library or a GPU is missing, calls raise.
Nothing touches the HIP runtime at import time (fork-safety, s2p/parallel.py:80): the library is
dlopen'ed on first use and the context is created per process, lazily.
"""
import ctypes
import os
import sys
import threading

import numpy as np

from s2p_amd import broker

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("S2P_HIP_LIB") or os.path.join(HERE, "lib", "libs2p_hip.so")   # S2P_HIP_LIB: a probe build (tools/build_variants.sh)

OK, EMPTY_RANGE, TIMEOUT, RUNTIME_ERROR, UNSUPPORTED, BAD_ARGUMENT = range(6)


class HipError(RuntimeError):
    """libs2p_hip.so reported a failure (no silent fallback exists)."""

    def __init__(self, code, msg):
        super().__init__("libs2p_hip: status %d: %s" % (code, msg))
        self.code = code
        self.msg = msg

    def __reduce__(self):
        # An exception travels from a Pool worker to its parent as a pickle, and the default reduction of an exception re-calls
        # __init__ with `args` alone -- (text,) here, one argument short: the parent's result-handler thread died on it, every later
        # result of that Pool was lost with it, and r.get(timeout) answered TimeoutError.  That, not a stuck GPU, is the "worker that
        # never came back" of round 4's 16-process direct-mode Pools (profiles/r04/pool_direct_sweep_run2...json): one worker's
        # HipError -- a hand-off wait that timed out under 16 time-sliced processes -- turned into a silent loss of the whole Pool.
        return (HipError, (self.code, self.msg))


class SgbmParams(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in
                ("win", "P1", "P2", "lr", "prefilter_cap", "uniqueness_ratio", "speckle_window", "speckle_range")]


class SgbmDump(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in
                ("q1", "q2", "C", "S", "disp_raw", "disp_med", "disp_fin", "cost_raw")] + \
               [("geom", ctypes.c_int * 8), ("rminmax", ctypes.c_float * 2)]


class CensusParams(ctypes.Structure):
    _fields_ = [("census_win", ctypes.c_int), ("P1", ctypes.c_int), ("P2", ctypes.c_int), ("nb_dir", ctypes.c_int),
                ("lr_check", ctypes.c_int), ("lr_tau", ctypes.c_float), ("mindiff", ctypes.c_int),
                ("median", ctypes.c_int), ("remove_small_cc", ctypes.c_int), ("fix_overcount", ctypes.c_int),
                ("recursion", ctypes.c_int), ("scales", ctypes.c_int), ("subpix", ctypes.c_int), ("cost", ctypes.c_int)]


class CensusDump(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("C", "S", "disp_raw", "disp_med")] + [("dmin0", ctypes.c_int), ("D0", ctypes.c_int)]


_lib = None
_lock = threading.Lock()
_ctx = {}          # (pid, device, stream) -> ctx pointer
_ctx_locks = {}    # id of a shared context -> lock: a context owns ONE workspace, so calls on it are serialised
                   # (the reference's workers are processes; threads that want overlap use one context each, tiles.py)


class RpcStruct(ctypes.Structure):
    """s2p_rpc of include/s2p_hip.h = struct rpc of c/rpc.h:13-31 (s2p_amd.triangulation.RPCStruct subclasses it)."""
    _fields_ = [("numx", ctypes.c_double * 20), ("denx", ctypes.c_double * 20),
                ("numy", ctypes.c_double * 20), ("deny", ctypes.c_double * 20),
                ("scale", ctypes.c_double * 3), ("offset", ctypes.c_double * 3),
                ("inumx", ctypes.c_double * 20), ("idenx", ctypes.c_double * 20),
                ("inumy", ctypes.c_double * 20), ("ideny", ctypes.c_double * 20),
                ("iscale", ctypes.c_double * 3), ("ioffset", ctypes.c_double * 3),
                ("dmval", ctypes.c_double * 4), ("imval", ctypes.c_double * 4),
                ("delta", ctypes.c_double)]


class TileDesc(ctypes.Structure):
    """s2p_tile of include/s2p_hip.h."""
    _fields_ = [("src1", ctypes.c_void_p), ("src1_dtype", ctypes.c_int), ("sw1", ctypes.c_int), ("sh1", ctypes.c_int),
                ("H1", ctypes.c_double * 9),
                ("src2", ctypes.c_void_p), ("src2_dtype", ctypes.c_int), ("sw2", ctypes.c_int), ("sh2", ctypes.c_int),
                ("H2", ctypes.c_double * 9),
                ("w", ctypes.c_int), ("h", ctypes.c_int), ("dmin", ctypes.c_int), ("dmax", ctypes.c_int),
                ("algo", ctypes.c_int),
                ("sgbm", ctypes.POINTER(SgbmParams)), ("census", ctypes.POINTER(CensusParams)),
                ("erosion", ctypes.c_int),
                ("rpca", ctypes.c_void_p), ("rpcb", ctypes.c_void_p),
                ("ha", ctypes.c_double * 9), ("hb", ctypes.c_double * 9),
                ("msk_orig", ctypes.c_void_p), ("ow", ctypes.c_int), ("oh", ctypes.c_int),
                ("bbox", ctypes.c_float * 4)]


class TileOut(ctypes.Structure):
    """s2p_tile_out of include/s2p_hip.h."""
    _fields_ = [(n, ctypes.c_void_p) for n in ("rect1", "rect2", "disp", "mask", "lonlatalt", "err")]


class _held:
    """Serialise host-level calls that share a context."""

    def __init__(self, ctx):
        key = ctx.value if hasattr(ctx, "value") else int(ctx)
        with _lock:
            self.lock = _ctx_locks.setdefault(key, threading.Lock())

    def __enter__(self):
        self.lock.acquire()

    def __exit__(self, *a):
        self.lock.release()


def lib():
    """dlopen libs2p_hip.so (does not initialise HIP)."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    # in-tree build on first use (hipcc cross-compiles; there is no other code path to fall back to)
                    try:
                        from s2p_amd import build as _build
                        _build.build()
                    except Exception as e:
                        raise HipError(RUNTIME_ERROR, "%s missing and could not be built (%s); run python -m s2p_amd.build"
                                       % (LIB_PATH, e))
                L = ctypes.CDLL(LIB_PATH)
                try:
                    L.s2p_hip_build_info
                except AttributeError:
                    # a library of an earlier round left in place by an in-tree update (the .so is untracked and only built when it is
                    # missing), or an old variant selected through S2P_HIP_LIB: rebuild the shipped path once, else say what to do
                    L = None
                    if "S2P_HIP_LIB" not in os.environ:
                        try:
                            from s2p_amd import build as _build
                            _build.build(force=True)
                            L = ctypes.CDLL(LIB_PATH)
                            L.s2p_hip_build_info
                        except Exception:
                            L = None
                    if L is None:
                        raise HipError(RUNTIME_ERROR, "%s is a stale library (no s2p_hip_build_info): run python -m s2p_amd.build --force" % LIB_PATH)
                L.s2p_hip_last_error.restype = ctypes.c_char_p
                L.s2p_hip_build_info.restype = ctypes.c_char_p
                info = L.s2p_hip_build_info().decode("utf-8", "replace")
                if "PROBE BUILD" in info:                    # measurement switches on: results may be invalid (csrc/probe_guard.hpp)
                    if "S2P_HIP_LIB" not in os.environ:
                        raise HipError(RUNTIME_ERROR, "%s is a probe build (%s): the shipped path only ever holds the shipped configuration; "
                                       "rebuild with python -m s2p_amd.build --force" % (LIB_PATH, info))
                    sys.stderr.write("s2p_amd: %s\n" % info)
                L.s2p_hip_ctx_create.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
                L.s2p_hip_ctx_destroy.argtypes = [ctypes.c_void_p]
                L.s2p_hip_ctx_destroy.restype = None
                L.s2p_hip_ctx_sync.argtypes = [ctypes.c_void_p]
                L.s2p_hip_ctx_use_graphs.argtypes = [ctypes.c_void_p, ctypes.c_int]
                L.s2p_hip_sgbm_default_params.argtypes = [ctypes.POINTER(SgbmParams)]
                L.s2p_hip_sgbm_default_params.restype = None
                fp = ctypes.c_void_p
                L.s2p_hip_sgbm_host.argtypes = [ctypes.c_void_p, fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                ctypes.c_int, ctypes.POINTER(SgbmParams), fp, fp, fp, ctypes.c_double]
                L.s2p_hip_sgbm_debug.argtypes = [ctypes.c_void_p, fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                 ctypes.c_int, ctypes.POINTER(SgbmParams), fp, fp, fp,
                                                 ctypes.POINTER(SgbmDump)]
                L.s2p_hip_sgbm_dev.argtypes = [ctypes.c_void_p, fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                               ctypes.c_int, ctypes.POINTER(SgbmParams), fp, fp, fp]
                L.s2p_hip_sgbm_geometry.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int * 8]
                L.s2p_hip_census_default_params.argtypes = [ctypes.POINTER(CensusParams)]
                L.s2p_hip_census_default_params.restype = None
                L.s2p_hip_census_sgm_host.argtypes = [ctypes.c_void_p, fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                      ctypes.c_int, ctypes.POINTER(CensusParams), fp, fp, fp, ctypes.c_double]
                L.s2p_hip_census_sgm_debug.argtypes = [ctypes.c_void_p, fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                       ctypes.c_int, ctypes.POINTER(CensusParams), fp, fp, fp,
                                                       ctypes.POINTER(CensusDump)]
                L.s2p_hip_census_sgm_dev.argtypes = [ctypes.c_void_p, fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                     ctypes.c_int, ctypes.POINTER(CensusParams), fp, fp, fp]
                L.s2p_hip_census_sgm_dev_batch.argtypes = [ctypes.c_void_p, ctypes.c_int, fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                           ctypes.c_int, ctypes.POINTER(CensusParams), fp, fp, fp]
                L.s2p_hip_census_sgm_host_batch.argtypes = [ctypes.c_void_p, ctypes.c_int, fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                            ctypes.c_int, ctypes.POINTER(CensusParams), fp, fp, fp, ctypes.c_double]
                L.s2p_hip_census_sgm_host_batch_v.argtypes = [ctypes.c_void_p, ctypes.c_int, fp, fp, fp, fp, fp, fp, ctypes.POINTER(CensusParams),
                                                              fp, fp, fp, ctypes.c_double]
                L.s2p_hip_census_sgm_host_batch_reserve.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                                    ctypes.POINTER(CensusParams)]
                L.s2p_hip_host_register.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
                L.s2p_hip_host_unregister.argtypes = [ctypes.c_void_p]
                L.s2p_hip_host_unregister.restype = None
                L.s2p_hip_warp_host.argtypes = [ctypes.c_void_p, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                ctypes.POINTER(ctypes.c_double), fp, ctypes.c_int, ctypes.c_int]
                L.s2p_hip_warp_dev.argtypes = L.s2p_hip_warp_host.argtypes
                L.s2p_hip_erode_mask_host.argtypes = [ctypes.c_void_p, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, fp]
                L.s2p_hip_disp_to_lonlatalt_host.argtypes = [ctypes.c_void_p, fp, fp, fp, fp, fp, ctypes.c_int, ctypes.c_int,
                                                             fp, ctypes.c_int, ctypes.c_int, fp, fp, fp, fp, fp]
                L.s2p_hip_stereo_corresp_to_lonlatalt_host.argtypes = [ctypes.c_void_p, fp, fp, fp, fp, ctypes.c_int, fp, fp]
                L.s2p_hip_count_3d_neighbors_host.argtypes = [ctypes.c_void_p, fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int]
                L.s2p_hip_remove_isolated_3d_points_host.argtypes = [ctypes.c_void_p, fp, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                                                     ctypes.c_int, ctypes.c_int, ctypes.c_int]
                L.s2p_hip_rejection_mask_host.argtypes = [ctypes.c_void_p, fp, fp, fp, ctypes.c_int, ctypes.c_int, fp]
                L.s2p_hip_merge_n_host.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), fp, ctypes.c_int, ctypes.c_int,
                                                   ctypes.c_int, ctypes.c_int, ctypes.c_double, fp]
                L.s2p_hip_height_transfer_host.argtypes = [ctypes.c_void_p, fp, ctypes.c_int, ctypes.c_int, fp, ctypes.c_int, ctypes.c_int, fp]
                L.s2p_hip_plyflatten_host.argtypes = [ctypes.c_void_p, fp, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double,
                                                      ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, fp]
                L.s2p_hip_height_map_to_lonlatalt_host.argtypes = [ctypes.c_void_p, fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, fp]
                L.s2p_hip_cargarse_basura_host.argtypes = [ctypes.c_void_p, fp, ctypes.c_int, ctypes.c_int, fp]
                L.s2p_hip_pinned_alloc.argtypes = [ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]
                L.s2p_hip_pinned_free.argtypes = [ctypes.c_void_p]
                L.s2p_hip_pinned_free.restype = None
                L.s2p_hip_tile_host.argtypes = [ctypes.c_void_p, ctypes.POINTER(TileDesc), ctypes.POINTER(TileOut), ctypes.c_double]
                L.s2p_hip_tile_host_batch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(TileDesc), ctypes.POINTER(TileOut), ctypes.c_double]
                L.s2p_hip_timing_enable.argtypes = [ctypes.c_void_p, ctypes.c_int]
                L.s2p_hip_timing_reset.argtypes = [ctypes.c_void_p]
                L.s2p_hip_timing_get.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_double),
                                                 ctypes.POINTER(ctypes.c_int)]
                _lib = L
    return _lib


def last_error():
    return lib().s2p_hip_last_error().decode("utf-8", "replace")


def check(code):
    if code != OK:
        raise HipError(code, last_error())


def device_count():
    n = lib().s2p_hip_device_count()
    if n < 0:                      # forked from a process that had already initialised HIP (see s2p_hip_device_count)
        raise HipError(RUNTIME_ERROR, last_error())
    return n


def default_device():
    """GPU of this worker: S2P_HIP_DEVICE, else LOCAL_RANK, else (pid mod device count) so that the
    reference's forked Pool workers (s2p/parallel.py:76-98) spread over the node's GPUs."""
    for k in ("S2P_HIP_DEVICE", "LOCAL_RANK"):
        if k in os.environ:
            return int(os.environ[k])
    n = device_count()
    if n <= 0:
        raise HipError(RUNTIME_ERROR, "no HIP device visible; the s2p_amd hot path has no CPU fallback")
    return os.getpid() % n


_tls = threading.local()


class thread_context:
    """Within the block, context() of THIS thread answers `ctx` (the GPU broker runs its workers' array-level calls on a few contexts
    side by side instead of serialising them on the process's one)."""

    def __init__(self, ctx):
        self.ctx = ctx

    def __enter__(self):
        self.old = getattr(_tls, "ctx", None)
        _tls.ctx = self.ctx

    def __exit__(self, *a):
        _tls.ctx = self.old


def context(device=None, stream=None):
    """Per-(process, device) context, created lazily (first HIP call of the process)."""
    if stream is None and getattr(_tls, "ctx", None) is not None:
        return _tls.ctx
    if device is None:
        device = default_device()
    key = (os.getpid(), device, stream)
    c = _ctx.get(key)
    if c is None:
        p = ctypes.c_void_p()
        check(lib().s2p_hip_ctx_create(device, stream, ctypes.byref(p)))
        c = _ctx[key] = p
    return c


def default_sgbm_params(**kw):
    p = SgbmParams()
    lib().s2p_hip_sgbm_default_params(ctypes.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def sgbm_geometry(w, dmin, dmax):
    g = (ctypes.c_int * 8)()
    check(lib().s2p_hip_sgbm_geometry(w, dmin, dmax, g))
    return dict(zip(("Wc", "width1", "D", "minD", "x0", "minX1", "maxX1", "invalid"), list(g)))


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def sgbm(im1, im2, dmin, dmax, params=None, timeout=-1.0, want_cost=True, want_mask=True, device=None, dump=False, ctx=None, pinned=False, out=None):
    """Run the sgbm matcher on two float32 arrays; returns dict(disp, cost, mask[, stage dumps]).
    `out`: dict of caller-owned C-contiguous arrays (disp[, cost][, mask]) that receive the results instead of fresh ones."""
    im1 = np.ascontiguousarray(im1, np.float32)
    im2 = np.ascontiguousarray(im2, np.float32)
    assert im1.shape == im2.shape and im1.ndim == 2
    h, w = im1.shape
    p = params or default_sgbm_params()
    new = pinned_empty if pinned else np.empty              # pinned: the results come back by DMA into page-locked arrays
    disp = new((h, w), np.float32)
    cost = new((h, w), np.float32) if want_cost else None
    mask = new((h, w), np.uint8) if want_mask else None
    if out is not None:
        disp, cost, mask = out["disp"], out.get("cost"), out.get("mask")
        assert disp.shape == (h, w) and disp.dtype == np.float32 and disp.flags.c_contiguous
    ctx = ctx or context(device)
    out = dict(disp=disp, cost=cost, mask=mask)
    if not dump:
        with _held(ctx):
            check(lib().s2p_hip_sgbm_host(ctx, _ptr(im1), _ptr(im2), w, h, int(dmin), int(dmax), ctypes.byref(p),
                                          _ptr(disp), _ptr(cost), _ptr(mask), float(timeout)))
        return out
    g = sgbm_geometry(w, int(dmin), int(dmax))
    d = SgbmDump()
    arrs = dict(q1=np.zeros((h, w), np.uint8), q2=np.zeros((h, w), np.uint8),
                disp_raw=np.zeros((h, g["Wc"]), np.int16), disp_med=np.zeros((h, g["Wc"]), np.int16),
                disp_fin=np.zeros((h, g["Wc"]), np.int16), cost_raw=np.zeros((h, g["Wc"]), np.int16))
    if dump == "full" and g["width1"] > 0:
        arrs["C"] = np.zeros((h, g["width1"], g["D"]), np.int16)
        arrs["S"] = np.zeros((h, g["width1"], g["D"]), np.int16)
    for k, a in arrs.items():
        setattr(d, k, a.ctypes.data)
    with _held(ctx):
        check(lib().s2p_hip_sgbm_debug(ctx, _ptr(im1), _ptr(im2), w, h, int(dmin), int(dmax), ctypes.byref(p),
                                       _ptr(disp), _ptr(cost), _ptr(mask), ctypes.byref(d)))
    out.update(arrs)
    out["geom"] = list(d.geom)
    out["rminmax"] = list(d.rminmax)
    return out


def default_census_params(**kw):
    p = CensusParams()
    lib().s2p_hip_census_default_params(ctypes.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def census_sgm(im1, im2, dmin, dmax, params=None, timeout=-1.0, want_conf=True, want_mask=True, device=None, dump=False, ctx=None, pinned=False, out=None):
    """Census / 8-path SGM matcher ('mgm' family stand-in); [dmin, dmax] inclusive.
    Returns dict(disp, conf, mask[, stage dumps]).  `out`: dict of caller-owned C-contiguous arrays (disp[, conf][, mask])."""
    im1 = np.ascontiguousarray(im1, np.float32)
    im2 = np.ascontiguousarray(im2, np.float32)
    assert im1.shape == im2.shape and im1.ndim == 2
    h, w = im1.shape
    p = params or default_census_params()
    new = pinned_empty if pinned else np.empty
    disp = new((h, w), np.float32)
    conf = new((h, w), np.float32) if want_conf else None
    mask = new((h, w), np.uint8) if want_mask else None
    if out is not None:
        disp, conf, mask = out["disp"], out.get("conf"), out.get("mask")
        assert disp.shape == (h, w) and disp.dtype == np.float32 and disp.flags.c_contiguous
    ctx = ctx or context(device)
    out = dict(disp=disp, conf=conf, mask=mask)
    if not dump:
        with _held(ctx):
            check(lib().s2p_hip_census_sgm_host(ctx, _ptr(im1), _ptr(im2), w, h, int(dmin), int(dmax), ctypes.byref(p),
                                                _ptr(disp), _ptr(conf), _ptr(mask), float(timeout)))
        return out
    D = ((2 if p.subpix == 2 else 1) * (int(dmax) - int(dmin)) + 1 + 15) // 16 * 16
    d = CensusDump()
    arrs = dict(disp_raw=np.zeros((h, w), np.float32), disp_med=np.zeros((h, w), np.float32))
    if dump == "full":
        arrs["C"] = np.zeros((h, w, D), np.uint8)
        arrs["S"] = np.zeros((h, w, D), np.uint16)
    for k, a in arrs.items():
        setattr(d, k, a.ctypes.data)
    with _held(ctx):
        check(lib().s2p_hip_census_sgm_debug(ctx, _ptr(im1), _ptr(im2), w, h, int(dmin), int(dmax), ctypes.byref(p),
                                             _ptr(disp), _ptr(conf), _ptr(mask), ctypes.byref(d)))
    out.update(arrs)
    out["dmin0"], out["D0"] = int(d.dmin0), int(d.D0)   # C / S are laid out [h][w][D0] from dmin0 (narrowed at the finest level of a multi-scale call)
    for k in ("C", "S"):
        if k in out and 0 < d.D0 != D:
            out[k] = out[k].reshape(-1)[:h * w * d.D0].reshape(h, w, d.D0)
    return out


def census_sgm_host_batch(ctx, im1, im2, w, h, dmin, dmax, params, disp, conf, mask, timeout=-1.0):
    """s2p_hip_census_sgm_host_batch on raw host ADDRESSES (lists of n ints; entries of conf / mask may be 0 = not wanted): n
    equal-shape pairs through one batched launch sequence, results written straight to the given addresses.  The caller owns
    the memory and `ctx` (the GPU broker: one context per lane, the addresses point into its clients' shared arenas)."""
    n = len(im1)
    P = ctypes.c_void_p * n
    with _held(ctx):
        check(lib().s2p_hip_census_sgm_host_batch(ctx, n, P(*im1), P(*im2), int(w), int(h), int(dmin), int(dmax), ctypes.byref(params),
                                                  P(*disp), P(*[c or None for c in conf]), P(*[m or None for m in mask]), float(timeout)))


def census_sgm_host_batch_v(ctx, im1, im2, w, h, dmin, dmax, params, disp, conf, mask, timeout=-1.0):
    """s2p_hip_census_sgm_host_batch_v: n tiles of different sizes / ranges (lists w, h, dmin, dmax) on raw host addresses, one call."""
    n = len(im1)
    P, I = ctypes.c_void_p * n, ctypes.c_int * n
    with _held(ctx):
        check(lib().s2p_hip_census_sgm_host_batch_v(ctx, n, P(*im1), P(*im2), I(*[int(v) for v in w]), I(*[int(v) for v in h]),
                                                    I(*[int(v) for v in dmin]), I(*[int(v) for v in dmax]), ctypes.byref(params),
                                                    P(*disp), P(*[c or None for c in conf]), P(*[m or None for m in mask]), float(timeout)))


def census_sgm_host_batch_reserve(ctx, n, w, h, dmin, dmax, params):
    """Size the context's workspace for host batches of up to n such tiles ahead of the first one."""
    with _held(ctx):
        check(lib().s2p_hip_census_sgm_host_batch_reserve(ctx, int(n), int(w), int(h), int(dmin), int(dmax), ctypes.byref(params)))


def host_register(addr, nbytes):
    """Page-lock a range this process already maps (hipHostRegister); raises HipError when the driver refuses."""
    check(lib().s2p_hip_host_register(ctypes.c_void_p(addr), int(nbytes)))


def host_unregister(addr):
    lib().s2p_hip_host_unregister(ctypes.c_void_p(addr))


@broker.remote()
def rejection_mask(disp, im1, im2, device=None):
    """create_rejection_mask (s2p/block_matching.py:18-32) on arrays."""
    disp = np.ascontiguousarray(disp, np.float32)
    im1 = np.ascontiguousarray(im1, np.float32)
    im2 = np.ascontiguousarray(im2, np.float32)
    h, w = disp.shape
    m = np.empty((h, w), np.uint8)
    c = context(device)
    with _held(c):
        check(lib().s2p_hip_rejection_mask_host(c, _ptr(disp), _ptr(im1), _ptr(im2), w, h, _ptr(m)))
    return m


_WARP_DTYPES = {np.dtype(np.float32): 0, np.dtype(np.uint16): 1, np.dtype(np.uint8): 2}


@broker.remote(out=lambda a: ((int(a["h"]), int(a["w"])), np.float32))
def warp(src, H, w, h, device=None, out=None):
    """`homography src -h H out w h` on arrays: dst(x) = src(H^-1 x), quintic B-spline, float32 out (`out`: a C-contiguous (h, w)
    float32 array to fill and return)."""
    src = np.ascontiguousarray(src)
    if src.dtype not in _WARP_DTYPES:
        src = src.astype(np.float32)
    Hm = np.ascontiguousarray(np.asarray(H, np.float64).reshape(9))
    sh, sw = src.shape
    if out is None:
        out = np.empty((int(h), int(w)), np.float32)
    assert out.shape == (int(h), int(w)) and out.dtype == np.float32 and out.flags.c_contiguous
    c = context(device)
    with _held(c):
        check(lib().s2p_hip_warp_host(c, _ptr(src), _WARP_DTYPES[src.dtype], sw, sh,
                                      Hm.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), _ptr(out), int(w), int(h)))
    return out


def _tile_desc(src1, H1, src2, H2, w, h, dmin, dmax, algo="census", params=None, erosion=0, tri=None,
               want_rect=True, out=None, pinned=False):
    """(s2p_tile, s2p_tile_out, result dict, objects the descriptors point into) of one tile() call."""
    srcs = []
    for s_ in (src1, src2):
        a = np.ascontiguousarray(s_)
        srcs.append(a if a.dtype in _WARP_DTYPES else a.astype(np.float32))
    w, h = int(w), int(h)
    t = TileDesc()
    t.src1, t.src1_dtype, t.sh1, t.sw1 = srcs[0].ctypes.data, _WARP_DTYPES[srcs[0].dtype], srcs[0].shape[0], srcs[0].shape[1]
    t.src2, t.src2_dtype, t.sh2, t.sw2 = srcs[1].ctypes.data, _WARP_DTYPES[srcs[1].dtype], srcs[1].shape[0], srcs[1].shape[1]
    t.H1[:] = list(np.asarray(H1, np.float64).reshape(9))
    t.H2[:] = list(np.asarray(H2, np.float64).reshape(9))
    t.w, t.h, t.dmin, t.dmax = w, h, int(dmin), int(dmax)
    t.algo = {"sgbm": 0, "census": 1}[algo]
    if params is not None:
        if algo == "sgbm":
            t.sgbm = ctypes.pointer(params)
        else:
            t.census = ctypes.pointer(params)
    t.erosion = int(erosion)
    want = {"disp": ((h, w), np.float32), "mask": ((h, w), np.uint8)}
    if want_rect:
        want["rect1"] = want["rect2"] = ((h, w), np.float32)
    if tri is not None:
        want["lonlatalt"], want["err"] = ((h, w, 3), np.float64), ((h, w), np.float32)
    if out is not None and sorted(out) == sorted(want) and all(out[k].shape == v[0] and out[k].dtype == v[1] for k, v in want.items()):
        pass                                                  # recycle the caller's buffers
    else:
        out = {k: (pinned_empty if pinned else np.empty)(v[0], v[1]) for k, v in want.items()}
    keep = [srcs, params]
    if tri is not None:
        mo = np.ascontiguousarray(tri["msk_orig"], np.float32)
        keep += [mo, tri["rpca"], tri["rpcb"]]
        t.rpca, t.rpcb = ctypes.addressof(tri["rpca"]), ctypes.addressof(tri["rpcb"])
        t.ha[:] = list(np.asarray(tri["ha"], np.float64).reshape(9))
        t.hb[:] = list(np.asarray(tri["hb"], np.float64).reshape(9))
        t.msk_orig, t.oh, t.ow = mo.ctypes.data, mo.shape[0], mo.shape[1]
        t.bbox[:] = [float(v) for v in tri["bbox"]]
    o = TileOut()
    for k, a in out.items():
        setattr(o, k, a.ctypes.data)
    return t, o, out, keep


def tile_batch(tiles, timeout=-1.0, device=None, ctx=None):
    """Several tiles of ONE shape (same w, h, dmin, dmax, census parameters) through one library call
    (s2p_hip_tile_host_batch): the pairs are matched by one batched launch sequence.  `tiles`: a list of dicts with the
    arguments of tile() (src1, H1, src2, H2, w, h, dmin, dmax[, params, erosion, tri, want_rect, out, pinned]);
    returns the list of result dicts, each byte-identical to what tile() returns for that tile."""
    if any(kw.get("algo", "census") != "census" for kw in tiles):
        raise ValueError("tile_batch: census / SGM tiles only (the sgbm matcher has no batched launch)")
    descs = [_tile_desc(**kw) for kw in tiles]
    n = len(descs)
    if n == 0:
        return []
    T, O = (TileDesc * n)(), (TileOut * n)()
    for i, d in enumerate(descs):
        T[i], O[i] = d[0], d[1]
    c = ctx if ctx is not None else context(device)
    with _held(c):
        check(lib().s2p_hip_tile_host_batch(c, n, T, O, float(timeout)))
    return [d[2] for d in descs]


def tile(src1, H1, src2, H2, w, h, dmin, dmax, algo="census", params=None, erosion=0, tri=None,
         want_rect=True, timeout=-1.0, device=None, ctx=None, out=None, pinned=False):
    """One tile through rectify -> match -> rejection mask (+ erosion) -> triangulation in ONE library call
    (s2p_hip_tile_host): the tile stays in HBM between the steps.

    src1, src2: source windows (float32 / uint16 / uint8 arrays); H1, H2: 3x3 maps from window to rectified
    coordinates; algo: 'sgbm' or 'census'; tri: None, or dict(rpca, rpcb (RpcStruct), ha, hb (3x3),
    msk_orig (2-D), bbox (4 floats)).  Returns dict(rect1, rect2, disp, mask[, lonlatalt, err]).
    `out`: a dict returned by an earlier call with the same shapes, whose arrays are overwritten and returned
    again (a scheduler that streams tiles avoids faulting in ~7 MB of fresh pages per tile that way).
    `pinned`: fresh result arrays come from pinned_empty (page-locked: the downloads are DMAs that overlap other tiles)."""
    t, o, out, keep = _tile_desc(src1, H1, src2, H2, w, h, dmin, dmax, algo, params, erosion, tri, want_rect, out, pinned)
    c = ctx if ctx is not None else context(device)
    with _held(c):
        check(lib().s2p_hip_tile_host(c, ctypes.byref(t), ctypes.byref(o), float(timeout)))
    return out


MERGE_OPS = {"average_if_close": 0, "np.nanmedian": 1, "np.median": 2, "np.nanmean": 3, "np.mean": 4,
             "np.nanmin": 5, "np.nanmax": 6, "np.min": 7, "np.max": 8}


@broker.remote()
def merge_n(images, offsets, averaging="average_if_close", threshold=1, device=None):
    """fusion.merge_n on arrays (s2p/fusion.py:26-68): pixelwise merge of n equal-size float32 maps after
    subtracting `offsets`; returns the float32 map (mean offset added back)."""
    name = averaging.replace("numpy.", "np.")
    if name not in MERGE_OPS:
        raise ValueError("merge_n: unsupported averaging %r (supported: %s)" % (averaging, ", ".join(MERGE_OPS)))
    imgs = [np.ascontiguousarray(a, np.float32) for a in images]
    assert len(imgs) == len(offsets) and len(imgs) > 0 and all(a.shape == imgs[0].shape and a.ndim == 2 for a in imgs)
    h, w = imgs[0].shape
    ptrs = (ctypes.c_void_p * len(imgs))(*[a.ctypes.data for a in imgs])
    off = np.ascontiguousarray(offsets, np.float64)
    out = np.empty((h, w), np.float32)
    c = context(device)
    with _held(c):
        check(lib().s2p_hip_merge_n_host(c, ptrs, _ptr(off), len(imgs), w, h, MERGE_OPS[name], float(threshold), _ptr(out)))
    return out


@broker.remote()
def height_transfer(heights, H, w, h, device=None):
    """The resampling half of triangulation.height_map (s2p/triangulation.py:376-389): `heights` (hr, wr) float64 on
    the rectified grid -> (h, w) float64 on the original grid through the affine map H (3x3, bottom row [0, 0, 1]),
    scipy's order-1 affine_transform plus its NaN handling, bit for bit."""
    a = np.ascontiguousarray(heights, np.float64)
    Hm = np.ascontiguousarray(np.asarray(H, np.float64).reshape(9))
    out = np.empty((int(h), int(w)), np.float64)
    c = context(device)
    with _held(c):
        check(lib().s2p_hip_height_transfer_host(c, _ptr(a), a.shape[1], a.shape[0], _ptr(Hm), int(w), int(h), _ptr(out)))
    return out


@broker.remote()
def plyflatten(cloud, xoff, yoff, resolution, xsize, ysize, radius=0, sigma=float("inf"), device=None):
    """`plyflatten.plyflatten` on an array (the C entry `rasterize_cloud` behind s2p/__init__.py:462-466): cloud is
    (n, 2 + nb) float64 rows x, y, values; returns the (ysize, xsize, nb) float32 raster, NaN where no point fell."""
    c_ = np.ascontiguousarray(cloud, np.float64)
    if c_.ndim != 2 or c_.shape[1] < 3:
        raise ValueError("plyflatten: cloud must be (n, 2 + nb) with nb >= 1")
    nb = c_.shape[1] - 2
    out = np.empty((int(ysize), int(xsize), nb), np.float32)
    c = context(device)
    with _held(c):
        check(lib().s2p_hip_plyflatten_host(c, _ptr(c_) if len(c_) else None, c_.shape[0], nb, float(xoff), float(yoff),
                                            float(resolution), int(xsize), int(ysize), int(radius), float(sigma), _ptr(out)))
    return out


@broker.remote()
def erode_mask(mask, radius, device=None):
    """masking.erosion on an array (s2p/masking.py:87-97)."""
    mask = np.ascontiguousarray(mask, np.uint8)
    h, w = mask.shape
    out = np.empty_like(mask)
    c = context(device)
    with _held(c):
        check(lib().s2p_hip_erode_mask_host(c, _ptr(mask), w, h, int(radius), _ptr(out)))
    return out


@broker.remote()
def height_map_to_lonlatalt(rpc, heights, off_x=0, off_y=0, device=None):
    """The localisation of triangulation.height_map_to_xyz (s2p/triangulation.py:165-219): (h, w) float32 heights on the
    grid of the reference image starting at (off_x, off_y) -> (h, w, 3) float64 lon, lat, alt (NaN where the height is)."""
    a = np.ascontiguousarray(heights, np.float32)
    if a.ndim != 2:
        raise ValueError("height_map_to_lonlatalt: heights must be 2-D")
    out = np.empty(a.shape + (3,), np.float64)
    c = context(device)
    with _held(c):
        check(lib().s2p_hip_height_map_to_lonlatalt_host(c, ctypes.addressof(rpc), _ptr(a), a.shape[1], a.shape[0], int(off_x), int(off_y), _ptr(out)))
    return out


@broker.remote()
def cargarse_basura(height_map, device=None):
    """common.cargarse_basura on an array (s2p/common.py:224-235): 5 x 5 range filter + small-component removal."""
    a = np.ascontiguousarray(height_map, np.float32)
    if a.ndim != 2:
        raise ValueError("cargarse_basura: a 2-D map is expected")
    out = np.empty_like(a)
    c = context(device)
    with _held(c):
        check(lib().s2p_hip_cargarse_basura_host(c, _ptr(a), a.shape[1], a.shape[0], _ptr(out)))
    return out


# ---- page-locked host arrays (include/s2p_hip.h: s2p_hip_pinned_alloc) ----------------------------------------------------
_PIN_CLASS = 1 << 20                        # blocks are rounded up to 1 MiB classes and recycled through per-class free lists
_pin_free = {}                              # (pid, nbytes) -> [address, ...]
_pin_lock = threading.RLock()               # re-entrant: a cyclic-GC pass inside a locked region may run another block's finalizer on this thread
_PIN_CACHE_BYTES = 1 << 30                  # at most 1 GiB of idle pinned blocks is kept per process
_PIN_LIVE_BYTES = int(os.environ.get("S2P_HIP_PINNED_MAX_MB", "8192")) << 20   # ... and at most this much is page-locked at a time:
_pin_idle = [0]                             # a job that keeps every tile's results (no sink) gets pageable arrays beyond it
_pin_live = [0]


def _pin_release(addr, nbytes, pid):
    try:
        if os.getpid() != pid:
            return                          # a forked child: the block belongs to the parent's runtime
        with _pin_lock:
            _pin_live[0] -= nbytes
            if _pin_idle[0] + nbytes <= _PIN_CACHE_BYTES:
                _pin_free.setdefault((pid, nbytes), []).append(addr)
                _pin_idle[0] += nbytes
                return
        lib().s2p_hip_pinned_free(addr)
    except Exception:
        pass                                # interpreter shutdown


def pinned_empty(shape, dtype=np.float32):
    """np.empty in page-locked host memory: transfers to / from such an array are DMAs that overlap kernels and other
    transfers (a pageable array is staged by the runtime on the calling thread).  Blocks are recycled per size class; a
    block returns to its free list when the last array (or view) on it dies.  Page-locking is an optimisation of the
    transfer, not a requirement of any entry point: when the driver refuses the allocation, or the process already holds
    S2P_HIP_PINNED_MAX_MB (default 8192) of live page-locked arrays, a plain np.empty comes back instead."""
    import weakref
    dt = np.dtype(dtype)
    count = int(np.prod(shape))
    nbytes = max(_PIN_CLASS, (count * dt.itemsize + _PIN_CLASS - 1) // _PIN_CLASS * _PIN_CLASS)
    pid = os.getpid()
    with _pin_lock:
        if _pin_live[0] + nbytes > _PIN_LIVE_BYTES:
            return np.empty(shape, dt)
        lst = _pin_free.get((pid, nbytes))
        addr = lst.pop() if lst else None
        if addr is not None:
            _pin_idle[0] -= nbytes
        _pin_live[0] += nbytes
    if addr is None:
        p = ctypes.c_void_p()
        if lib().s2p_hip_pinned_alloc(nbytes, ctypes.byref(p)) != OK or not p.value:
            with _pin_lock:
                _pin_live[0] -= nbytes
            return np.empty(shape, dt)      # (a process whose HIP runtime is unusable hears about it from the library call that follows)
        addr = p.value
    buf = (ctypes.c_char * nbytes).from_address(addr)      # numpy arrays on it keep `buf` alive through .base
    weakref.finalize(buf, _pin_release, addr, nbytes, pid)
    return np.frombuffer(buf, dtype=dt, count=count).reshape(shape)


def is_pinned(a):
    """True for arrays handed out by pinned_empty / pinned_copy (and their views)."""
    b = a
    while isinstance(b, np.ndarray) and b.base is not None:
        b = b.base
    return isinstance(b, ctypes.Array) or (isinstance(getattr(b, "obj", None), ctypes.Array))


def pinned_copy(a):
    """A page-locked copy of an array (same shape / dtype, C order)."""
    a = np.asarray(a)
    out = pinned_empty(a.shape, a.dtype)
    np.copyto(out, a)
    return out
