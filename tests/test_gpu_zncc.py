"""The ZNCC cost volume (s2p_census_params.cost = 1; north_star's "census/ZNCC"): HIP == oracle bit for bit at every stage, on
shapes that exercise the borders, NaNs, both windows, every lane layout, the multi-scale mode and both aggregations; and it
recovers a synthetic disparity field under a gain and an offset between the images (what a normalised correlation is for)."""
import numpy as np
import pytest

from helpers import same, synth_pair

pytestmark = pytest.mark.gpu

CASES = [
    # seed, H, W, dmin, dmax, nan, params
    (301, 40, 60, -3, 3, False, {}),
    (302, 50, 90, -20, 25, True, {"census_win": 3}),
    (303, 70, 200, -64, 63, False, {"recursion": 2}),
    (304, 150, 64, -30, 33, True, {"recursion": 1, "remove_small_cc": 25, "median": 0}),
    (305, 21, 300, -250, 250, False, {"P1": 4, "P2": 20}),
    (306, 1, 80, -8, 8, False, {}),
    (307, 90, 1, -2, 2, False, {"census_win": 3}),
    (308, 300, 256, -24, 40, False, {"scales": 6, "recursion": 1, "median": 0}),
    (309, 33, 517, 5, 70, False, {"nb_dir": 4}),
    # half-pixel candidates (round 5): the odd ones correlate with image 2 sampled half way between its columns
    (310, 40, 60, -3, 3, False, {"subpix": 2}),
    (311, 64, 130, -20, 25, True, {"subpix": 2, "census_win": 3, "recursion": 2}),
    (312, 1, 80, -8, 8, False, {"subpix": 2}),
    (313, 70, 1, -2, 2, False, {"subpix": 2}),
    (314, 280, 256, -24, 40, False, {"subpix": 2, "scales": 6, "recursion": 1, "median": 0, "remove_small_cc": 25}),
    (315, 25, 300, -200, 200, False, {"subpix": 2, "P1": 4, "P2": 20}),
]


@pytest.mark.parametrize("seed,H,W,dmin,dmax,nan,kw", CASES)
def test_zncc_every_stage_matches_the_oracle(oracle, seed, H, W, dmin, dmax, nan, kw):
    from s2p_amd import _lib as hip
    kw = dict(kw, cost=1)
    mid, amp = 0.5 * (dmin + dmax), 0.2 * (dmax - dmin)
    im1, im2 = synth_pair(seed, H, W, lambda x, y: mid + amp * np.sin(x / 23.) * np.cos(y / 19.), nan=nan)
    im2 = (0.7 * im2 + 40.0).astype(np.float32)                 # another gain and an offset: census-like invariance is the point of ZNCC
    r = hip.census_sgm(im1, im2, dmin, dmax, params=hip.default_census_params(**{"recursion": 0, **kw}), dump="full")
    o = oracle.oracle_census_sgm(im1, im2, dmin, dmax, params=oracle.census_params(**kw), dump="full")
    assert o["rc"] == 0 and (o["dmin0"], o["D0"]) == (r["dmin0"], r["D0"])
    for k in ("C", "S", "disp_raw", "disp_med", "disp", "conf", "mask"):
        assert same(o[k], r[k]), "stage %s: HIP != oracle" % k
    c = o["C"][o["C"] < 255]
    assert c.size == 0 or c.max() <= 24


def test_zncc_recovers_a_field_under_gain_and_offset(oracle):
    from s2p_amd import _lib as hip
    f = lambda x, y: 6 + 5 * np.sin(x / 31.) * np.cos(y / 23.)
    im1, im2 = synth_pair(5, 200, 320, f)
    im2 = (0.5 * im2 + 300.0).astype(np.float32)
    r = hip.census_sgm(im1, im2, -4, 16, params=hip.default_census_params(cost=1, recursion=2))["disp"]
    c = hip.census_sgm(im1, im2, -4, 16, params=hip.default_census_params(cost=0, recursion=2))["disp"]
    both = np.isfinite(r) & np.isfinite(c)
    assert both.mean() > 0.9
    assert (np.abs(r[both] - c[both]) <= 0.5).mean() > 0.97      # the two costs agree on a textured scene


def test_zncc_refusals():
    from s2p_amd import _lib as hip
    a = np.zeros((8, 64), np.float32)
    with pytest.raises(hip.HipError) as e:
        hip.census_sgm(a, a, -4, 4, params=hip.default_census_params(cost=2))
    assert e.value.code == hip.BAD_ARGUMENT
    wide = np.zeros((2, 4200), np.float32)
    with pytest.raises(hip.HipError) as e:
        hip.census_sgm(wide, wide, -4, 4, params=hip.default_census_params(cost=1))
    assert e.value.code == hip.UNSUPPORTED
    wide2 = np.zeros((2, 2600), np.float32)                       # with half-pixel candidates a third image's window rows share the LDS
    with pytest.raises(hip.HipError) as e:
        hip.census_sgm(wide2, wide2, -4, 4, params=hip.default_census_params(cost=1, subpix=2))
    assert e.value.code == hip.UNSUPPORTED


def test_shim_selects_the_cost_from_cfg():
    from s2p_amd import block_matching as bm
    from s2p_amd.config import cfg
    c = dict(cfg, hip_mgm_cost="zncc")
    assert bm.matcher_params("mgm", c)[1].cost == 1 and bm.matcher_params("mgm_multi", c)[1].cost == 1
    assert bm.matcher_params("mgm")[1].cost == 0 and bm.matcher_params("mgm_multi")[1].subpix == 1
    assert bm.matcher_params("mgm_multi", dict(cfg, hip_mgm_multi_subpix=2))[1].subpix == 2
