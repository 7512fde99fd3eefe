"""GPU parity tests of the census / 8-path SGM matcher (the `mgm` / `mgm_multi` stand-in) through
the C ABI.  Two bars (DESIGN_PARITY.md):
  * INTERNAL: bit-exact against oracle/census_oracle.c at every stage (integer pipeline + one IEEE
    division) -- this is what the tests below assert on seeded inputs;
  * EXTERNAL: statistical agreement with the one mgm output the reference's tests hold
    (tests/golden/mgm_tile.npz, from tests/data/input_triangulation/pair_1) -- mgm's source is not in
    the reference tree, so this can only be statistical (tests/test_gpu_fixture_tile.py)."""
import numpy as np
import pytest

from helpers import same, synth_pair

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from s2p_amd import _lib
    assert _lib.device_count() > 0, "no MI355X visible: the HIP path has no fallback"
    return _lib


CASES = [
    # seed, H, W, dmin, dmax (inclusive), nan, params
    (51, 40, 60, -3, 3, False, {}),                       # Dt=7   D=16  G=2
    (52, 33, 70, -10, 20, True, {}),                      # Dt=31  D=32  G=4
    (53, 50, 90, -20, 25, False, {"median": 0}),          # Dt=46  D=48  G=8 padded
    (54, 64, 96, -32, 31, True, {"remove_small_cc": 25}), # Dt=64  D=64  G=8
    (55, 45, 130, -40, 50, False, {"lr_check": 0}),       # Dt=91  D=96  G=16 padded
    (56, 70, 200, -64, 63, False, {}),                    # Dt=128 D=128 G=16
    (57, 37, 260, -100, 90, False, {"census_win": 3}),    # Dt=191 D=192 G=32 padded
    (58, 30, 300, -128, 127, False, {"P1": 4, "P2": 20}), # Dt=256 D=256 G=32
    (59, 21, 520, -250, 250, False, {}),                  # Dt=501 D=512 G=64
    (64, 24, 900, -400, 399, False, {}),                  # Dt=800  D=800  16 per lane, padded
    (65, 16, 1100, -512, 511, True, {"remove_small_cc": 25}),   # Dt=1024 D=1024 the maximum
    (60, 1, 80, -8, 8, False, {}),                        # single row
    (61, 60, 90, 4, 30, False, {"remove_small_cc": 25, "median": 0}),
    (62, 60, 90, -30, -4, False, {}),
    (63, 257, 131, -24, 40, True, {"remove_small_cc": 25}),
    (66, 64, 150, -20, 43, True, {"fix_overcount": 0}),     # the plain sum of the 8 path costs
    (67, 40, 300, -128, 127, False, {"fix_overcount": 0, "P2": 100}),
    # MGM recursion (two predecessors per direction, wavefront launches)
    (70, 40, 60, -3, 3, False, {"recursion": 1}),                          # D=16  G=2
    (71, 50, 90, -20, 25, True, {"recursion": 1, "median": 0}),            # D=48  G=8 padded
    (72, 70, 200, -64, 63, False, {"recursion": 1}),                       # D=128 G=16, wider than high
    (73, 150, 64, -30, 33, True, {"recursion": 1, "remove_small_cc": 25}), # higher than wide
    (74, 21, 300, -250, 250, False, {"recursion": 1, "P1": 4, "P2": 20}),  # D=512 G=64
    (75, 12, 700, -400, 399, False, {"recursion": 1}),                     # D=800: 16 per lane, padded
    (76, 1, 80, -8, 8, False, {"recursion": 1}),                           # single row
    (77, 90, 1, -2, 2, False, {"recursion": 1, "fix_overcount": 0}),       # single column
    # three predecessors (recursion = 2: the 'mgm' call's TSGM=3 as modelled), same layouts
    (170, 40, 60, -3, 3, False, {"recursion": 2}),
    (171, 50, 90, -20, 25, True, {"recursion": 2, "median": 0}),
    (172, 70, 200, -64, 63, False, {"recursion": 2}),
    (173, 150, 64, -30, 33, True, {"recursion": 2, "remove_small_cc": 25}),
    (174, 21, 300, -250, 250, False, {"recursion": 2, "P1": 4, "P2": 20}),
    (175, 12, 700, -400, 399, False, {"recursion": 2, "P2": 127, "P1": 50}),
    (176, 1, 80, -8, 8, False, {"recursion": 2}),
    (177, 90, 1, -2, 2, False, {"recursion": 2, "fix_overcount": 0}),
    (178, 300, 256, -24, 40, False, {"scales": 6, "subpix": 2, "recursion": 2, "median": 0, "remove_small_cc": 25}),
    (179, 90, 150, -30, 33, False, {"nb_dir": 4, "recursion": 2}),
    # mgm_multi: half-pixel candidates (SUBPIX=2) and the coarse-to-fine mode (-S); sizes >= 256 have >= 2 levels
    (80, 40, 60, -3, 3, False, {"subpix": 2}),                             # Dt=13  D=16
    (81, 50, 90, -20, 25, True, {"subpix": 2, "median": 0}),               # Dt=91  D=96 padded, NaN pixels
    (82, 33, 200, -64, 63, False, {"subpix": 2, "recursion": 1}),          # Dt=255 D=256
    (83, 21, 300, -250, 250, False, {"subpix": 2, "lr_tau": 0.5}),         # Dt=1001 D=1008: 16 per lane, padded
    (84, 60, 90, 4, 30, False, {"subpix": 2, "remove_small_cc": 25, "median": 0}),   # one-sided ranges
    (85, 60, 90, -31, -4, False, {"subpix": 2, "census_win": 3}),
    (86, 256, 300, -24, 40, True, {"scales": 6}),                          # 2 levels (128 x 150 parent)
    (87, 300, 256, -24, 40, False, {"scales": 6, "recursion": 1, "median": 0, "remove_small_cc": 25}),
    (88, 513, 517, -33, 31, True, {"scales": 6, "subpix": 2, "median": 0, "remove_small_cc": 25}),   # 3 levels, odd sizes
    (89, 512, 512, -96, 95, False, {"scales": 2, "subpix": 2, "recursion": 1, "median": 0}),         # -S smaller than the size allows
    (92, 300, 280, -20, 30, True, {"scales": 6, "lr_check": 2, "recursion": 1}),   # L-R test at the last scale only
    (93, 70, 120, -20, 11, True, {"nb_dir": 4}),                           # mgm -O 4: the axis directions only
    (94, 90, 150, -30, 33, False, {"nb_dir": 4, "recursion": 1, "median": 0}),
    (95, 130, 200, -128, 127, True, {"nb_dir": 4, "recursion": 1, "fix_overcount": 0, "P2": 100}),
    (96, 256, 260, -16, 20, False, {"nb_dir": 4, "scales": 6, "subpix": 2, "recursion": 1}),
    # 16 directions (cfg['mgm_nb_directions'] = 16): the knight's moves as 40 more lattices of the band kernel, 16 e-volumes in the WTA
    (270, 40, 60, -3, 3, False, {"nb_dir": 16, "recursion": 2}),                             # D=16  G=2
    (271, 50, 90, -20, 25, True, {"nb_dir": 16, "recursion": 1, "median": 0}),               # D=48  G=8 padded, NaN pixels
    (272, 70, 200, -64, 63, False, {"nb_dir": 16, "recursion": 2}),                          # D=128 G=16
    (273, 150, 64, -30, 33, True, {"nb_dir": 16, "recursion": 2, "remove_small_cc": 25}),    # higher than wide
    (274, 21, 300, -250, 250, False, {"nb_dir": 16, "recursion": 1, "P1": 4, "P2": 20}),     # D=512 G=64
    (275, 12, 700, -400, 399, False, {"nb_dir": 16, "recursion": 2, "P2": 127, "P1": 50}),   # D=800: 16 per lane (4-bit index keys: capped sums), padded
    (276, 1, 80, -8, 8, False, {"nb_dir": 16, "recursion": 2}),                              # single row: the knight lattices are single points
    (277, 90, 1, -2, 2, False, {"nb_dir": 16, "recursion": 1, "fix_overcount": 0}),          # single column
    (278, 2, 3, -2, 2, False, {"nb_dir": 16, "recursion": 2}),                               # smaller than a knight's move
    (279, 131, 257, -24, 40, True, {"nb_dir": 16, "recursion": 1, "fix_overcount": 0, "P2": 128, "P1": 100}),   # the largest sums
    (280, 300, 256, -24, 40, False, {"nb_dir": 16, "scales": 6, "subpix": 2, "recursion": 1, "median": 0, "remove_small_cc": 25}),   # mgm_multi's shape
    (281, 200, 333, -100, 90, False, {"nb_dir": 16, "recursion": 2, "mindiff": 12}),         # D=192 G=32 padded, the MINDIFF variant
    (282, 257, 300, -128, 127, True, {"nb_dir": 16, "recursion": 2, "lr_check": 0}),         # D=256 G=32
    (90, 254, 600, -10, 10, False, {"scales": 6}),                         # smaller side 254 -> 127 < 128: stays single scale
    (91, 255, 600, -10, 10, True, {"scales": 6, "subpix": 2}),             # 255 -> 128: two levels
]


@pytest.mark.parametrize("seed,H,W,dmin,dmax,nan,kw", CASES)
def test_every_stage_matches_oracle(hip, oracle, seed, H, W, dmin, dmax, nan, kw):
    mid, amp = 0.5 * (dmin + dmax), 0.2 * (dmax - dmin)
    im1, im2 = synth_pair(seed, H, W, lambda x, y: mid + amp * np.sin(x / 23.) * np.cos(y / 19.), nan=nan)
    r = hip.census_sgm(im1, im2, dmin, dmax, params=hip.default_census_params(**{"recursion": 0, **kw}), dump="full")
    o = oracle.oracle_census_sgm(im1, im2, dmin, dmax, params=oracle.census_params(**kw), dump="full")
    assert o["rc"] == 0
    assert (o["dmin0"], o["D0"]) == (r["dmin0"], r["D0"])      # layout of the C / S dumps (narrowed at the finest level of a multi-scale call)
    for k in ("C", "S", "disp_raw", "disp_med", "disp", "conf", "mask"):
        assert same(o[k], r[k]), "stage %s: HIP != oracle" % k
    # without the confidence image the packed-16 WTA kernel runs (the default of the file-level 'mgm' call and of bench.py)
    q = hip.census_sgm(im1, im2, dmin, dmax, params=hip.default_census_params(**{"recursion": 0, **kw}), want_conf=False)
    assert same(o["disp"], q["disp"]) and same(o["mask"], q["mask"])


def test_recovers_synthetic_field(hip):
    f = lambda x, y: 6 + 9 * np.sin(x / 37.) * np.cos(y / 29.)
    im1, im2 = synth_pair(5, 256, 384, f)
    d = hip.census_sgm(im1, im2, -24, 39, params=hip.default_census_params(recursion=0))["disp"]
    xx, yy = np.meshgrid(np.arange(384.), np.arange(256.))
    t = f(xx, yy)
    for _ in range(40):
        t = f(xx + t, yy)
    v = np.isfinite(d)
    assert v.mean() > 0.9
    assert np.mean(np.abs(d[v] - t[v]) <= 0.5) > 0.97


def test_error_statuses(hip):
    im = np.zeros((16, 16), np.float32)
    with pytest.raises(hip.HipError) as e:
        hip.census_sgm(im, im, 3, 2, params=hip.default_census_params(recursion=0))
    assert e.value.code == hip.EMPTY_RANGE
    with pytest.raises(hip.HipError) as e:
        hip.census_sgm(im, im, -4, 4, timeout=0.0, params=hip.default_census_params(recursion=0))
    assert e.value.code == hip.TIMEOUT
    with pytest.raises(hip.HipError) as e:
        hip.census_sgm(im, im, -4, 4, params=hip.default_census_params(nb_dir=16, recursion=0))    # the knight's moves run under the MGM recursion only
    assert e.value.code == hip.UNSUPPORTED
    with pytest.raises(hip.HipError) as e:
        hip.census_sgm(im, im, -4, 4, params=hip.default_census_params(nb_dir=12))
    assert e.value.code == hip.UNSUPPORTED
    with pytest.raises(hip.HipError) as e:
        hip.census_sgm(im, im, -4, 4, params=hip.default_census_params(subpix=3))
    assert e.value.code == hip.UNSUPPORTED
    with pytest.raises(hip.HipError) as e:
        hip.census_sgm(np.zeros((4, 1200), np.float32), np.zeros((4, 1200), np.float32), -300, 300, params=hip.default_census_params(subpix=2))
    assert e.value.code == hip.UNSUPPORTED                                 # 1201 half-pixel candidates > 1024


def test_rejection_mask_entry(hip, oracle):
    rng = np.random.default_rng(3)
    d = rng.uniform(-5, 5, (30, 50)).astype(np.float32)
    d[rng.uniform(size=d.shape) < 0.2] = np.nan
    a = rng.uniform(0, 1, d.shape).astype(np.float32)
    b = a.copy()
    a[3, 4] = np.nan
    b[10:12, 20:25] = np.nan
    assert same(oracle.oracle_rejection_mask(d, a, b), hip.rejection_mask(d, a, b))


def test_full_size_exact(hip, oracle):
    """BASELINE.json configs[1]: 1024x1024 tile, 128 disparities, census 5x5, 8 paths."""
    im1, im2 = synth_pair(7, 1024, 1024, lambda x, y: 40 * np.sin(2 * np.pi * x / 512.) * np.cos(2 * np.pi * y / 512.))
    r = hip.census_sgm(im1, im2, -64, 63, params=hip.default_census_params(recursion=0))
    r2 = hip.census_sgm(im1, im2, -64, 63, params=hip.default_census_params(recursion=0))
    assert same(r["disp"], r2["disp"])                      # deterministic despite LDS atomics
    o = oracle.oracle_census_sgm(im1, im2, -64, 63, params=oracle.census_params(recursion=0))
    assert same(o["disp"], r["disp"]) and same(o["mask"], r["mask"]) and same(o["conf"], r["conf"])
    q = hip.census_sgm(im1, im2, -64, 63, want_conf=False, params=hip.default_census_params(recursion=0))   # the benchmarked kernels (packed WTA)
    assert same(o["disp"], q["disp"]) and same(o["mask"], q["mask"])


def test_cost_volume_beyond_2_gib(hip, oracle):
    """Buffer offsets are 32-bit UNSIGNED: a cost volume between 2 and 4 GiB (1460 x 1440 x 1024 = 2.15 G candidates, 17 GB of
    e-volumes) runs and matches the oracle in both aggregation modes; 4 GiB and more is refused."""
    im1, im2 = synth_pair(95, 1440, 1460, lambda x, y: 300 * np.sin(x / 400.) * np.cos(y / 350.))
    o = oracle.oracle_census_sgm(im1, im2, -512, 511, params=oracle.census_params(recursion=0))
    r = hip.census_sgm(im1, im2, -512, 511, want_conf=False, params=hip.default_census_params(recursion=0))
    assert same(o["disp"], r["disp"]) and same(o["mask"], r["mask"])
    del o
    om = oracle.oracle_census_sgm(im1, im2, -512, 511, params=oracle.census_params(recursion=1))
    rm = hip.census_sgm(im1, im2, -512, 511, params=hip.default_census_params(recursion=1), want_conf=False)
    assert same(om["disp"], rm["disp"])
    with pytest.raises(hip.HipError) as e:
        hip.census_sgm(np.zeros((2100, 2100), np.float32), np.zeros((2100, 2100), np.float32), -512, 511, params=hip.default_census_params(recursion=0))
    assert e.value.code == hip.UNSUPPORTED


def test_widest_supported_tile_and_the_refusal_beyond(hip, oracle):
    """One image row of per-pixel state lives in the LDS of a CU (156 of its 160 KiB as dynamic LDS: include/s2p_hip.h,
    Limits): the widest tile that fits runs and matches the oracle -- in the 8-path mode and, at 9000 px, in the MGM mode the
    shim runs --; one pixel more is refused with S2P_HIP_UNSUPPORTED, not a launch failure."""
    D = 16
    wmax = (156 * 1024 - 16 - 4 * D) // 10
    assert wmax > 15000
    im1, im2 = synth_pair(81, 3, wmax, lambda x, y: 2 + 0 * x)
    r = hip.census_sgm(im1, im2, -4, 11, want_conf=False, params=hip.default_census_params(recursion=0))
    o = oracle.oracle_census_sgm(im1, im2, -4, 11, params=oracle.census_params(recursion=0))
    assert same(o["disp"], r["disp"]) and same(o["mask"], r["mask"])
    im1m, im2m = synth_pair(84, 40, 9000, lambda x, y: 3 + 2 * np.sin(x / 300.))
    rm = hip.census_sgm(im1m, im2m, -4, 11, params=hip.default_census_params(recursion=2))
    om = oracle.oracle_census_sgm(im1m, im2m, -4, 11, params=oracle.census_params(recursion=2))
    assert same(om["disp"], rm["disp"]) and same(om["mask"], rm["mask"]) and same(om["conf"], rm["conf"])
    im1w, im2w = synth_pair(82, 2, wmax + 1, lambda x, y: 0 * x)
    with pytest.raises(hip.HipError) as e:
        hip.census_sgm(im1w, im2w, -4, 11, params=hip.default_census_params(recursion=0))
    assert e.value.code == hip.UNSUPPORTED
    with pytest.raises(hip.HipError) as e:
        hip.sgbm(np.zeros((2, 8200), np.float32), np.zeros((2, 8200), np.float32), -4, 11)
    assert e.value.code == hip.UNSUPPORTED
    im1s, im2s = synth_pair(83, 3, 8100, lambda x, y: 3 + 0 * x)               # sgbm canvas 8100 + 11 + 4 < 8192
    rs = hip.sgbm(im1s, im2s, -4, 11)
    oracle.set_alias_oob(0)
    os_ = oracle.oracle_sgbm(im1s, im2s, -4, 11)
    oracle.set_alias_oob(1)
    assert same(os_["disp"], rs["disp"])
