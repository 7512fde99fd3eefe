"""GPU parity tests (run with -m gpu on an MI355X): the HIP sgbm path, called through the C ABI,
against (a) the golden vectors generated from the real reference, (b) the CPU oracle on fresh seeded
inputs at every stage, (c) the real reference library when it travelled to the box, and (d)
size-independent properties at the BASELINE.json full size.  Bar: bit-exact (integer arithmetic)."""
import numpy as np
import pytest

from helpers import golden_names, load_golden, same, synth_pair

pytestmark = pytest.mark.gpu

STAGES = ("q1", "q2", "C", "S", "disp_raw", "cost_raw", "disp_med", "disp_fin", "disp", "cost")


@pytest.fixture(scope="module")
def hip():
    from s2p_amd import _lib
    assert _lib.device_count() > 0, "no MI355X visible: the HIP path has no fallback"
    return _lib


@pytest.mark.parametrize("name", golden_names("sgbm_"))
def test_golden_vectors(hip, oracle, name):
    g = load_golden(name)
    dmin, dmax = int(g["params"][0]), int(g["params"][1])
    r = hip.sgbm(g["im1"], g["im2"], dmin, dmax, dump="full" if "C" in g else True)
    assert r["geom"] == list(g["geom"])
    assert np.array_equal(np.array(r["rminmax"], np.float32), g["rminmax"])
    if name == "sgbm_neg_range_oob":
        # the reference's out-of-bounds disp2 store fires here (DESIGN_PARITY.md 2, "padded semantics"):
        # everything up to S is still exact, the final map differs on a handful of pixels
        for k in ("q1", "q2"):
            assert same(g[k], r[k])
        frac = np.mean(~((r["disp"] == g["disp"]) | (np.isnan(r["disp"]) & np.isnan(g["disp"]))))
        assert frac < 2e-3
        oracle.set_alias_oob(0)
        o = oracle.oracle_sgbm(g["im1"], g["im2"], dmin, dmax, dump=True)
        oracle.set_alias_oob(1)
        for k in ("disp_raw", "disp_med", "disp_fin", "disp", "cost"):
            assert same(o[k], r[k]), k
        return
    for k in STAGES:
        if k in g:
            assert same(g[k], r[k]), "stage %s differs from the reference golden vector" % k


CASES = [
    # seed, H, W, dmin, dmax, nan      (D -> lane-group size G, padded or not)
    (21, 40, 60, -3, 3, False),        # D=16  G=2
    (22, 33, 70, -10, 20, True),       # D=32  G=4
    (23, 50, 90, -20, 25, False),      # D=48  G=8 padded
    (24, 64, 96, -32, 32, True),       # D=64  G=8
    (25, 45, 130, -40, 50, False),     # D=96  G=16 padded
    (26, 70, 200, -64, 64, False),     # D=128 G=16
    (27, 37, 260, -100, 90, False),    # D=192 G=32 padded
    (28, 30, 300, -128, 128, False),   # D=256 G=32
    (29, 21, 520, -250, 250, False),   # D=512 G=64
    (30, 25, 400, -150, 170, False),   # D=320 G=64 padded
    (37, 24, 900, -400, 400, False),   # D=800  16 disparities per lane, 64-lane groups, padded
    (38, 16, 1100, -512, 512, False),  # D=1024 the maximum
    (31, 1, 80, -8, 8, False),         # single row
    (32, 2, 80, -8, 8, False),         # two rows
    (33, 3, 50, -8, 8, False),
    (34, 60, 90, 4, 30, False),        # one-sided ranges
    (35, 60, 90, -30, -4, False),
    (36, 257, 131, -24, 40, True),     # odd sizes crossing strip / chunk borders
]


@pytest.mark.parametrize("seed,H,W,dmin,dmax,nan", CASES)
def test_every_stage_matches_oracle(hip, oracle, seed, H, W, dmin, dmax, nan):
    mid, amp = 0.5 * (dmin + dmax), 0.2 * (dmax - dmin)
    im1, im2 = synth_pair(seed, H, W, lambda x, y: mid + amp * np.sin(x / 23.) * np.cos(y / 19.), nan=nan)
    r = hip.sgbm(im1, im2, dmin, dmax, dump="full")
    oracle.set_alias_oob(0)          # the HIP path implements the padded semantics (DESIGN_PARITY.md 2)
    o = oracle.oracle_sgbm(im1, im2, dmin, dmax, dump="full")
    oracle.set_alias_oob(1)
    assert r["geom"] == o["geom"]
    for k in STAGES:
        assert same(o[k], r[k]), "stage %s: HIP != oracle" % k
    assert same(oracle.oracle_rejection_mask(o["disp"], im1, im2), r["mask"])


@pytest.mark.parametrize("seed,H,W,dmin,dmax", [(41, 120, 200, -24, 40), (42, 90, 160, -32, 32)])
def test_matches_live_reference(hip, oracle, seed, H, W, dmin, dmax):
    if not oracle.have_ref():
        pytest.skip("oracle/_ref/libsgbm_ref.so did not travel to this box")
    im1, im2 = synth_pair(seed, H, W, lambda x, y: 8 + 10 * np.sin(x / 31.) * np.cos(y / 23.))
    a = oracle.ref_sgbm(im1, im2, dmin, dmax, dump="full")
    r = hip.sgbm(im1, im2, dmin, dmax, dump="full")
    for k in STAGES:
        assert same(a[k], r[k]), "stage %s: HIP != real reference" % k


def test_all_nan_and_constant_inputs(hip, oracle):
    z = np.full((20, 40), np.nan, np.float32)
    c = np.full((20, 40), 7.0, np.float32)
    for a, b in ((z, z), (c, c), (c, z)):
        r = hip.sgbm(a, b, -8, 8)
        oracle.set_alias_oob(0)
        o = oracle.oracle_sgbm(a, b, -8, 8)
        oracle.set_alias_oob(1)
        assert same(o["disp"], r["disp"])


def test_error_statuses(hip):
    im = np.zeros((16, 16), np.float32)
    with pytest.raises(hip.HipError) as e:
        hip.sgbm(im, im, 3, 3)
    assert e.value.code == hip.EMPTY_RANGE
    with pytest.raises(hip.HipError) as e:
        hip.sgbm(im, im, -4, 4, timeout=0.0)
    assert e.value.code == hip.TIMEOUT
    with pytest.raises(hip.HipError) as e:
        hip.sgbm(im, im, -4, 4, params=hip.default_sgbm_params(win=5))
    assert e.value.code == hip.UNSUPPORTED
    wide = np.zeros((4, 1200), np.float32)
    with pytest.raises(hip.HipError) as e:
        hip.sgbm(wide, wide, -600, 600)                      # D = 1200 > 1024
    assert e.value.code == hip.UNSUPPORTED


def test_full_size_properties(hip, oracle):
    """BASELINE.json config 2 size (1024x1024, D=128).  The oracle needs ~10 s here, so use
    size-independent properties plus one oracle comparison on a horizontal band."""
    H = W = 1024
    im1, im2 = synth_pair(7, H, W, lambda x, y: 40 * np.sin(2 * np.pi * x / 512.) * np.cos(2 * np.pi * y / 512.))
    r1 = hip.sgbm(im1, im2, -64, 64)
    r2 = hip.sgbm(im1, im2, -64, 64)
    assert same(r1["disp"], r2["disp"]) and same(r1["mask"], r2["mask"])        # deterministic (atomics, CCL)
    d = r1["disp"]
    v = np.isfinite(d)
    assert v.mean() > 0.5
    assert np.all(d[v] * 16 == np.round(d[v] * 16))                              # 1/16 px fixed point
    assert d[v].min() >= -64 - 1 and d[v].max() <= 64                            # inside the searched range
    # synth_pair defines the field on im2's grid: the disparity seen from im1 solves delta = f(x + delta, y)
    f = lambda x, y: 40 * np.sin(2 * np.pi * x / 512.) * np.cos(2 * np.pi * y / 512.)
    xx, yy = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    truth = f(xx, yy)
    for _ in range(50):
        truth = f(xx + truth, yy)
    assert np.mean(np.abs(d[v] - truth[v]) <= 1.0) > 0.95                        # recovers the synthetic field
    # a pure horizontal shift of both images by k pixels leaves interior disparities unchanged
    k = 16
    r3 = hip.sgbm(im1[:, k:], im2[:, k:], -64, 64)
    a, b = d[200:800, 300 + k:700 + k], r3["disp"][200:800, 300:700]
    both = np.isfinite(a) & np.isfinite(b)
    assert np.mean(a[both] == b[both]) > 0.97
    # exactness on the full size against the oracle (one call, ~10 s of CPU)
    oracle.set_alias_oob(0)
    o = oracle.oracle_sgbm(im1, im2, -64, 64)
    oracle.set_alias_oob(1)
    assert same(o["disp"], d)


def test_cost_volume_beyond_2_gib(hip, oracle):
    """The int16 cost volume of a 1000 x 1100 tile over [-512, 512] is 3.3 GB (canvas 2124 px, 1612 usable columns, 1024
    disparities): buffer offsets are 32-bit unsigned, so it runs -- and matches the oracle; 4 GiB and more is refused."""
    im1, im2 = synth_pair(96, 1000, 1100, lambda x, y: 250 * np.sin(x / 300.) * np.cos(y / 280.))
    r = hip.sgbm(im1, im2, -512, 512)
    oracle.set_alias_oob(0)
    o = oracle.oracle_sgbm(im1, im2, -512, 512)
    oracle.set_alias_oob(1)
    assert same(o["disp"], r["disp"]) and same(o["cost"], r["cost"])
    assert same(oracle.oracle_rejection_mask(o["disp"], im1, im2), r["mask"])
    with pytest.raises(hip.HipError) as e:
        hip.sgbm(np.zeros((1500, 1100), np.float32), np.zeros((1500, 1100), np.float32), -512, 512)
    assert e.value.code == hip.UNSUPPORTED
