"""Randomised GPU-vs-oracle sweep over shapes, ranges and NaN patterns (seeded: reproducible).
Small images make tile-border, strip-border, lane-padding and tiny-range corner cases frequent."""
import os

import numpy as np
import pytest

from helpers import same

pytestmark = pytest.mark.gpu
SEED = int(os.environ.get('S2P_FUZZ_SEED', '0'))      # extra sweeps: S2P_FUZZ_SEED=k python -m pytest tests/test_gpu_fuzz.py -m gpu


@pytest.fixture(scope="module")
def hip():
    from s2p_amd import _lib
    assert _lib.device_count() > 0
    return _lib


def rand_pair(rng, h, w, nan_frac):
    base = rng.uniform(0, 1000, (h, w + 64)).astype(np.float32)
    for _ in range(2):                                           # cheap blur
        base[:, 1:-1] = (base[:, :-2] + 2 * base[:, 1:-1] + base[:, 2:]) / 4
        if h > 2:
            base[1:-1, :] = (base[:-2, :] + 2 * base[1:-1, :] + base[2:, :]) / 4
    s = int(rng.integers(0, 24))
    im1 = np.ascontiguousarray(base[:, 32:32 + w])
    im2 = np.ascontiguousarray(base[:, 32 - s + 12:32 - s + 12 + w] * np.float32(rng.uniform(0.8, 1.2)))
    if nan_frac > 0:
        im1[rng.uniform(size=im1.shape) < nan_frac] = np.nan
        im2[rng.uniform(size=im2.shape) < nan_frac] = np.nan
    return im1, im2


def rand_geometry(rng):
    h = int(rng.choice([1, 2, 3, 5, 17, 31, 32, 33, 64, 97]))
    w = int(rng.choice([3, 8, 15, 16, 31, 33, 64, 65, 100, 129]))
    lo = int(rng.integers(-40, 30))
    span = int(rng.choice([1, 2, 7, 15, 16, 17, 31, 33, 64, 100]))
    return h, w, lo, lo + span


@pytest.mark.parametrize("chunk", range(4))
def test_sgbm_fuzz(hip, oracle, chunk):
    rng = np.random.default_rng(1000 + chunk + 10 * SEED)
    for _ in range(40):
        h, w, dmin, dmax = rand_geometry(rng)
        im1, im2 = rand_pair(rng, h, w, float(rng.choice([0, 0, 0.02, 0.3])))
        if hip.sgbm_geometry(w, dmin, dmax)["width1"] == 1:          # undefined in the reference: both sides refuse
            with pytest.raises(hip.HipError) as e:
                hip.sgbm(im1, im2, dmin, dmax)
            assert e.value.code == hip.UNSUPPORTED and oracle.oracle_sgbm(im1, im2, dmin, dmax)["rc"] == 4
            continue
        r = hip.sgbm(im1, im2, dmin, dmax, dump="full")
        oracle.set_alias_oob(0)
        o = oracle.oracle_sgbm(im1, im2, dmin, dmax, dump="full")
        oracle.set_alias_oob(1)
        tag = "h=%d w=%d d=[%d,%d]" % (h, w, dmin, dmax)
        assert r["geom"] == o["geom"], tag
        for k in ("q1", "q2", "C", "S", "disp_raw", "cost_raw", "disp_med", "disp_fin", "disp", "cost"):
            if k in o and k in r:
                assert same(o[k], r[k]), "%s stage %s" % (tag, k)
        assert same(oracle.oracle_rejection_mask(o["disp"], im1, im2), r["mask"]), tag


@pytest.mark.parametrize("chunk", range(4))
def test_census_fuzz(hip, oracle, chunk):
    rng = np.random.default_rng(2000 + chunk + 10 * SEED)
    for _ in range(40):
        h, w, dmin, dmax = rand_geometry(rng)
        im1, im2 = rand_pair(rng, h, w, float(rng.choice([0, 0, 0.02, 0.3])))
        kw = dict(census_win=int(rng.choice([3, 5])), median=int(rng.integers(0, 2)), lr_check=int(rng.integers(0, 2)),
                  remove_small_cc=int(rng.choice([0, 5, 25])), P1=int(rng.choice([4, 8])), P2=int(rng.choice([16, 32, 100])),
                  fix_overcount=int(rng.integers(0, 2)), recursion=int(rng.integers(0, 3)), nb_dir=int(rng.choice([8, 8, 4])))
        r = hip.census_sgm(im1, im2, dmin, dmax, params=hip.default_census_params(**{"recursion": 0, **kw}), dump="full")
        o = oracle.oracle_census_sgm(im1, im2, dmin, dmax, params=oracle.census_params(**kw), dump="full")
        tag = "h=%d w=%d d=[%d,%d] %s" % (h, w, dmin, dmax, kw)
        for k in ("C", "S", "disp_raw", "disp_med", "disp", "conf", "mask"):
            assert same(o[k], r[k]), "%s stage %s" % (tag, k)
        # the default call has no confidence image and runs the packed WTA kernel (k_wta_census_pk)
        q = hip.census_sgm(im1, im2, dmin, dmax, params=hip.default_census_params(**{"recursion": 0, **kw}), want_conf=False)
        assert same(o["disp"], q["disp"]) and same(o["mask"], q["mask"]), "%s packed WTA" % tag


@pytest.mark.parametrize("chunk", range(2))
def test_mgm_multi_fuzz(hip, oracle, chunk):
    """Coarse-to-fine levels and half-pixel candidates at random sizes around the 128-pixel pyramid threshold."""
    rng = np.random.default_rng(4000 + chunk + 10 * SEED)
    for _ in range(8):
        h = int(rng.choice([64, 127, 128, 255, 256, 257, 300, 511, 513]))
        w = int(rng.choice([130, 255, 256, 259, 300, 512, 515]))
        lo = int(rng.integers(-40, 10))
        dmin, dmax = lo, lo + int(rng.choice([3, 16, 33, 64]))
        im1, im2 = rand_pair(rng, h, w, float(rng.choice([0, 0, 0.02, 0.2])))
        kw = dict(scales=int(rng.choice([1, 2, 6])), subpix=int(rng.choice([1, 2])), median=int(rng.integers(0, 2)),
                  lr_check=int(rng.integers(0, 3)), remove_small_cc=int(rng.choice([0, 25])), recursion=int(rng.integers(0, 3)),
                  census_win=int(rng.choice([3, 5])))
        r = hip.census_sgm(im1, im2, dmin, dmax, params=hip.default_census_params(**{"recursion": 0, **kw}), dump="full")
        o = oracle.oracle_census_sgm(im1, im2, dmin, dmax, params=oracle.census_params(**kw), dump="full")
        tag = "h=%d w=%d d=[%d,%d] %s" % (h, w, dmin, dmax, kw)
        for k in ("disp_raw", "disp_med", "disp", "conf", "mask"):
            assert same(o[k], r[k]), "%s stage %s" % (tag, k)
        q = hip.census_sgm(im1, im2, dmin, dmax, params=hip.default_census_params(**{"recursion": 0, **kw}), want_conf=False)
        assert same(o["disp"], q["disp"]) and same(o["mask"], q["mask"]), "%s packed WTA" % tag


def test_warp_fuzz(hip, oracle):
    rng = np.random.default_rng(3000 + SEED)
    for _ in range(60):
        sh, sw = int(rng.integers(1, 90)), int(rng.integers(1, 90))
        src = rng.uniform(0, 500, (sh, sw)).astype(np.float32)
        if rng.uniform() < 0.4 and sh * sw > 4:
            src[rng.uniform(size=src.shape) < 0.05] = np.nan
        a = rng.uniform(-0.3, 0.3)
        H = np.array([[np.cos(a) * rng.uniform(0.8, 1.3), -np.sin(a), rng.uniform(-10, 10)],
                      [np.sin(a), np.cos(a) * rng.uniform(0.8, 1.3), rng.uniform(-10, 10)],
                      [rng.uniform(-1e-4, 1e-4), rng.uniform(-1e-4, 1e-4), 1.0]])
        w, h = int(rng.integers(1, 100)), int(rng.integers(1, 100))
        out, ref = hip.warp(src, H, w, h), oracle.oracle_warp(src, H, w, h)
        assert same(out, ref), (sh, sw, w, h)
