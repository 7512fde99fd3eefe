"""CPU tests on the one tile of the reference's test data that pins the two un-vendored binaries
(`homography`, `mgm`): tests/golden/warp_tile.npz and mgm_tile.npz.
These pin the ORACLES (oracle/resample_oracle.c, oracle/census_oracle.c) empirically/statistically:
the binaries' sources are not in the reference tree."""
import numpy as np

from helpers import load_golden

# bars (measured values in the comments; the fixture's own noise floor for the resampler is the
# %12.6f rounding of H_ref.txt: mean 0.006 on values 100..700)
WARP_MEAN, WARP_MAX = 0.02, 0.15
MGM_HALF_PX, MGM_ONE_PX, MGM_VALID_GAP = 0.985, 0.995, 0.01


def interior(a, b=12):
    return a[b:-b, b:-b]


def test_resampler_reproduces_rectified_ref(oracle):
    g = load_golden("warp_tile")
    w, h = (int(v) for v in g["size"])
    out = oracle.oracle_warp(g["src"], g["H"], w, h)
    e = np.abs(out - g["expected"])
    assert np.isfinite(out).all()
    assert interior(e).mean() <= WARP_MEAN          # measured 0.0082
    assert interior(e).max() <= WARP_MAX            # measured 0.087
    assert e.mean() <= WARP_MEAN                    # borders included: 0.0088


def test_lower_spline_orders_do_not_fit(oracle):
    """Guards the empirical claim 'quintic': bilinear sampling of the same crop misses the fixture by
    two orders of magnitude more."""
    g = load_golden("warp_tile")
    w, h = (int(v) for v in g["size"])
    Hi = np.linalg.inv(g["H"])
    xx, yy = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    X, Y, Z = (Hi[i, 0] * xx + Hi[i, 1] * yy + Hi[i, 2] for i in range(3))
    u, v = X / Z, Y / Z
    x0, y0 = np.floor(u).astype(int), np.floor(v).astype(int)
    fx, fy = u - x0, v - y0
    s = g["src"].astype(np.float64)
    bil = (1 - fy) * ((1 - fx) * s[y0, x0] + fx * s[y0, x0 + 1]) + fy * ((1 - fx) * s[y0 + 1, x0] + fx * s[y0 + 1, x0 + 1])
    assert interior(np.abs(bil - g["expected"])).mean() > 1.0


def test_census_matcher_agrees_with_stored_mgm_tile(oracle):
    g = load_golden("mgm_tile")
    w, h = (int(v) for v in g["size"])
    sec = oracle.oracle_warp(g["src"], g["H"], w, h)
    d_ref = g["disp"]
    dmin = int(np.floor(np.nanmin(d_ref))) - 4
    dmax = int(np.ceil(np.nanmax(d_ref))) + 4
    r = oracle.oracle_census_sgm(g["ref"], sec, dmin, dmax)          # the 'mgm' call's parameters
    d = r["disp"]
    both = np.isfinite(d) & np.isfinite(d_ref)
    e = np.abs(d[both] - d_ref[both])
    assert (e <= 0.5).mean() >= MGM_HALF_PX         # measured 0.989 (0.977 without the overcount fix)
    assert (e <= 1.0).mean() >= MGM_ONE_PX          # measured 0.997
    assert abs(np.isfinite(d).mean() - np.isfinite(d_ref).mean()) <= MGM_VALID_GAP   # 0.930 vs 0.950
    # MGM recursion (recursion = 1): the published two-predecessor form, integer mean of the messages
    rm = oracle.oracle_census_sgm(g["ref"], sec, dmin, dmax, params=oracle.census_params(recursion=1))["disp"]
    bm = np.isfinite(rm) & np.isfinite(d_ref)
    em = np.abs(rm[bm] - d_ref[bm])
    assert (em <= 0.5).mean() >= 0.99 and (em <= 1.0).mean() >= 0.997      # measured 0.9953 / 0.9983
    assert abs(np.isfinite(rm).mean() - np.isfinite(d_ref).mean()) <= 0.01   # 0.953 vs 0.950
    # mask convention of the fixture: mask == isfinite(disp) (values 0/1)
    assert set(np.unique(g["mask"])) <= {0, 1}
    assert np.array_equal(g["mask"] == 1, np.isfinite(d_ref))
