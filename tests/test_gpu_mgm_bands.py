"""The two device implementations of the MGM recursion (census matcher, recursion = 1) must agree bit for bit with
each other and with the oracle: `steps` = one launch per front (k_mgm_step), `bands` = one band-pipelined launch
with in-launch hand-offs between the waves of a workgroup (LDS ring + progress words, no barrier) and between
workgroups (tagged granules in global memory, no flags): k_mgm_bands, s2p_amd/csrc/mgm_bands.hpp.  S2P_MGM_IMPL is
read at every call, so one process can flip it.

The hand-offs cross CUs and XCDs: the shapes below include tiles with many bands per lattice, rows longer than
several chunks, ragged last bands, and the repeated full-size run looks for timing-dependent staleness."""
import os

import numpy as np
import pytest

from helpers import DevMem, same, synth_pair

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from s2p_amd import _lib
    assert _lib.device_count() > 0, "no MI355X visible: the HIP path has no fallback"
    return _lib


class impl:
    def __init__(self, name):
        self.env = {"S2P_MGM_IMPL": name}

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.env}
        os.environ.update(self.env)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


SHAPES = [
    # seed, H, W, dmin, dmax, nan, params
    (301, 1, 1, -2, 2, False, {}),
    (302, 1, 90, -8, 8, False, {}),                              # single row
    (303, 90, 1, -2, 2, False, {}),                              # single column
    (304, 40, 60, -3, 3, False, {}),                             # D=16  G=2: 128 rows per band
    (305, 300, 37, -3, 3, True, {}),                             # several bands at G=2, higher than wide
    (306, 50, 90, -20, 25, True, {"median": 0}),                 # D=48 G=8 padded
    (307, 70, 200, -64, 63, False, {}),                          # D=128 G=16: 16 rows per band
    (308, 131, 257, -24, 40, True, {"remove_small_cc": 25}),     # odd sizes, ragged last band
    (309, 257, 131, -64, 63, False, {"fix_overcount": 0}),
    (310, 33, 300, -250, 250, False, {"P1": 4, "P2": 20}),       # D=512 G=64: 4 rows per band
    (311, 12, 400, -400, 399, False, {}),                        # D=800: 16 per lane, padded
    (312, 9, 700, -512, 511, False, {}),                         # D=1024
    (313, 64, 64, -16, 15, False, {"P1": 2, "P2": 128}),         # the largest P2 the matcher accepts
    (314, 5, 6000, -4, 11, False, {}),                           # wide: 750 chunks per row, diagonal lattices of 24 bands
    (315, 3000, 7, -4, 11, True, {}),                            # tall: 24 bands per axis lattice, rows of one chunk
    (316, 129, 129, -32, 31, False, {}),                         # (W - 1) / 2 a multiple of the band height: sweeps start on a chunk
]


@pytest.mark.parametrize("seed,H,W,dmin,dmax,nan,kw", SHAPES)
def test_bands_match_steps_and_oracle(hip, oracle, seed, H, W, dmin, dmax, nan, kw):
    mid, amp = 0.5 * (dmin + dmax), 0.2 * (dmax - dmin)
    im1, im2 = synth_pair(seed, H, W, lambda x, y: mid + amp * np.sin(x / 23.) * np.cos(y / 19.), nan=nan)
    kw = dict(kw, recursion=1)
    o = oracle.oracle_census_sgm(im1, im2, dmin, dmax, params=oracle.census_params(**kw), dump="full")
    assert o["rc"] == 0
    for name in ("steps", "bands"):
        with impl(name):
            r = hip.census_sgm(im1, im2, dmin, dmax, params=hip.default_census_params(**{"recursion": 0, **kw}), dump="full")
        for k in ("C", "S", "disp", "conf", "mask"):
            assert same(o[k], r[k]), "%s stage %s: HIP != oracle" % (name, k)


DEPTHS = [
    # H, W, dmin, dmax, params: candidate counts off the multiples of 64
    (50, 90, -20, 25, {"recursion": 1}),                         # 46 candidates: 48 in the oracle's layout, 64 in the library's
    (60, 80, -35, 40, {"recursion": 2}),                         # 76 -> 80 / 128
    (60, 80, -50, 55, {"recursion": 2, "nb_dir": 16}),           # 106 -> 112 / 128
    (40, 100, -70, 72, {"recursion": 1, "median": 0}),           # 143 -> 144 / 192
    (40, 100, -80, 90, {"recursion": 2, "nb_dir": 4}),           # 171 -> 176 / 192
    (30, 120, -100, 105, {"recursion": 2}),                      # 206 -> 208 / 256
    (30, 120, -118, 120, {"recursion": 1, "P1": 8, "P2": 115}),  # 239 -> 240 / 256, the largest P2 the padding argument covers
    (30, 120, -118, 120, {"recursion": 1, "P1": 8, "P2": 116}),  # above it: the depth stays at the multiple of 16
    (24, 160, -130, 140, {"recursion": 2}),                      # 271 -> 272 / 320
    (60, 80, -35, 40, {"recursion": 1, "cost": 1}),              # ZNCC costs
    (60, 80, -17, 20, {"recursion": 1, "subpix": 2}),            # 75 half-pixel candidates -> 80 / 128
]


@pytest.mark.parametrize("H,W,dmin,dmax,kw", DEPTHS)
def test_depth_rounded_up_to_64_changes_no_result(hip, oracle, H, W, dmin, dmax, kw):
    """Round 6 (csrc/census_kernels.hip: census_D): with the MGM recursion and P2 <= 115 the cost / e-volumes are laid out to the next multiple
    of 64 candidates (whole lines per pixel), the surplus excluded; a call with stage dumps keeps the oracle's multiple of 16.  Both give the
    oracle's maps, bit for bit."""
    mid, amp = 0.5 * (dmin + dmax), 0.3 * (dmax - dmin)
    im1, im2 = synth_pair(1000 + H + dmax, H, W, lambda x, y: mid + amp * np.sin(x / 17.) * np.cos(y / 13.), nan=True)
    o = oracle.oracle_census_sgm(im1, im2, dmin, dmax, params=oracle.census_params(**kw))
    assert o["rc"] == 0
    p = hip.default_census_params(**kw)
    r = hip.census_sgm(im1, im2, dmin, dmax, params=p)
    rd = hip.census_sgm(im1, im2, dmin, dmax, params=p, dump="full")
    for k in ("disp", "conf", "mask"):
        assert same(o[k], r[k]), "rounded depth, %s: HIP != oracle" % k
        assert same(o[k], rd[k]), "oracle's depth (dumps), %s: HIP != oracle" % k


@pytest.mark.parametrize("kw,conf", [
    ({"recursion": 2}, True),                                    # byte keys, consensus
    ({"recursion": 2}, False),                                   # no confidence image: the plain kernel, quad adds
    ({"recursion": 1, "P1": 10, "P2": 80}, True),                # P2 > 63: 16-bit keys
    ({"recursion": 1, "P1": 10, "P2": 80}, False),
    ({"recursion": 2, "nb_dir": 16}, True),                      # 16 e-volumes
    ({"recursion": 1, "mindiff": 3}, True),                      # MINDIFF
    ({"recursion": 2, "nb_dir": 4, "lr_check": 0}, True),
    ({"recursion": 1, "subpix": 2}, True),                       # 191 half-pixel candidates
])
def test_wta_with_12_candidates_per_lane(hip, oracle, kw, conf):
    """D = 192: the WTA runs 12 candidates per lane on the 16 lanes of a DPP row (k_wta_census_pk<16, 6>), in every variant of the kernel."""
    sp = 2 if kw.get("subpix") == 2 else 1
    dmin, dmax = (-90, 100) if sp == 1 else (-45, 50)
    im1, im2 = synth_pair(4242, 40, 120, lambda x, y: 0.5 * (dmin + dmax) + 0.3 * (dmax - dmin) * np.sin(x / 17.) * np.cos(y / 13.), nan=True)
    o = oracle.oracle_census_sgm(im1, im2, dmin, dmax, params=oracle.census_params(**kw))
    r = hip.census_sgm(im1, im2, dmin, dmax, params=hip.default_census_params(**kw), want_conf=conf)
    assert o["rc"] == 0
    for k in ("disp", "mask") + (("conf",) if conf else ()):
        assert same(o[k], r[k]), k


def test_a_large_tile_of_192_candidates_alone(hip):
    """D = 192 runs 12 candidates per lane (k_mgm_bands<16, 6>) in batches and, from 768 px on, for a tile launched alone; the front-by-front
    kernel (8 per lane on 32 lanes, no bands) is the independent check at a size the oracle does not finish in seconds."""
    im1, im2 = synth_pair(77, 776, 800, lambda x, y: 60 * np.sin(2 * np.pi * x / 400.) * np.cos(2 * np.pi * y / 300.), nan=True)
    p = hip.default_census_params(recursion=1)
    with impl("steps"):
        ref = hip.census_sgm(im1, im2, -96, 95, params=p)
    with impl("bands"):
        r = hip.census_sgm(im1, im2, -96, 95, params=p)
    for k in ("disp", "conf", "mask"):
        assert same(ref[k], r[k]), k


def test_full_size_repeated(hip, oracle):
    """1024 x 1024 x 128: 64 bands per lattice, 768 workgroups chained through 756 hand-off edges.  Ten runs, each
    compared with the front-by-front result: a stale or early-read row shows up as a run that differs."""
    im1, im2 = synth_pair(7, 1024, 1024, lambda x, y: 40 * np.sin(2 * np.pi * x / 512.) * np.cos(2 * np.pi * y / 512.))
    p = hip.default_census_params(recursion=1)
    with impl("steps"):
        ref = hip.census_sgm(im1, im2, -64, 63, params=p, want_conf=False)
    with impl("bands"):
        for i in range(20):
            r = hip.census_sgm(im1, im2, -64, 63, params=p, want_conf=False)
            assert same(ref["disp"], r["disp"]) and same(ref["mask"], r["mask"]), "run %d" % i


def test_bands_with_other_tiles_in_flight(hip):
    """Uneven load: four contexts (own streams and workspaces) run band launches of different sizes concurrently
    through the device entry; every result must equal the one computed alone."""
    import ctypes
    L, lib = hip, hip.lib()
    shapes = [(256, 384, -32, 31), (200, 300, -64, 63), (384, 256, -16, 15), (128, 512, -100, 90)]
    p = L.default_census_params(recursion=1)
    jobs, ctxs, mem = [], [], DevMem()
    with impl("bands"):
        try:
            for i, (H, W, dmin, dmax) in enumerate(shapes):
                im1, im2 = synth_pair(400 + i, H, W, lambda x, y: 0.3 * (dmin + dmax) + 5 * np.sin(x / 31.) * np.cos(y / 17.))
                alone = L.census_sgm(im1, im2, dmin, dmax, params=p, want_conf=False)
                a, b = mem.upload(im1), mem.upload(im2)
                d, m = mem.upload(np.zeros((H, W), np.float32)), mem.upload(np.zeros((H, W), np.uint8))
                c = ctypes.c_void_p()
                L.check(lib.s2p_hip_ctx_create(0, None, ctypes.byref(c)))
                ctxs.append(c)
                jobs.append((H, W, dmin, dmax, alone, a, b, d, m, c))
            for rep in range(5):
                for (H, W, dmin, dmax, alone, a, b, d, m, c) in jobs:
                    mem.fill(d, H * W * 4, 0); mem.fill(m, H * W, 7)
                for (H, W, dmin, dmax, alone, a, b, d, m, c) in jobs:
                    L.check(lib.s2p_hip_census_sgm_dev(c, a, b, W, H, dmin, dmax, ctypes.byref(p), d, None, m))
                for c in ctxs:
                    L.check(lib.s2p_hip_ctx_sync(c))
                for (H, W, dmin, dmax, alone, a, b, d, m, c) in jobs:
                    assert same(alone["disp"], mem.download(d, (H, W), np.float32)), "rep %d %dx%d" % (rep, H, W)
                    assert same(alone["mask"], mem.download(m, (H, W), np.uint8))
        finally:
            for c in ctxs:
                lib.s2p_hip_ctx_destroy(c)
            mem.free()


def test_graph_replay_rezeroes_the_control_block(hip):
    """Under hipGraph replay (s2p_hip_ctx_use_graphs) the ticket, the abort word and every tag of the row ring must be
    zeroed by a node of the graph itself: three replays of the captured MGM pipeline give the eager result."""
    import ctypes
    L, lib = hip, hip.lib()
    H, W, dmin, dmax = 200, 256, -20, 27
    im1, im2 = synth_pair(91, H, W, lambda x, y: 5 + 7 * np.sin(x / 33.) * np.cos(y / 27.))
    p = L.default_census_params(recursion=1)
    mem = DevMem()
    ctx = ctypes.c_void_p()
    L.check(lib.s2p_hip_ctx_create(0, None, ctypes.byref(ctx)))
    with impl("bands"):
        try:
            want = L.census_sgm(im1, im2, dmin, dmax, params=p, want_conf=False)
            a, b = mem.upload(im1), mem.upload(im2)
            d, m = mem.upload(np.zeros((H, W), np.float32)), mem.upload(np.zeros((H, W), np.uint8))
            L.check(lib.s2p_hip_ctx_use_graphs(ctx, 1))
            for rep in range(4):                               # rep 0 captures, 1..3 replay
                mem.fill(d, H * W * 4, 0); mem.fill(m, H * W, 7)
                L.check(lib.s2p_hip_census_sgm_dev(ctx, a, b, W, H, dmin, dmax, ctypes.byref(p), d, None, m))
                L.check(lib.s2p_hip_ctx_sync(ctx))
                assert same(want["disp"], mem.download(d, (H, W), np.float32)), "rep %d" % rep
                assert same(want["mask"], mem.download(m, (H, W), np.uint8))
        finally:
            lib.s2p_hip_ctx_destroy(ctx)
            mem.free()


def test_contended_launches_match_quiet_ones():
    """Random shapes, ranges and direction counts: every tile computed once on a quiet device, then again and again on
    three streams at once beside other tiles' launches (tools/mgm_stress.py; 90 s of it ran 176 000 launches without a
    mismatch at the end of round 2).  A race of the hand-off protocol shows as a mismatch or as a timeout error."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "mgm_stress.py"), "8"], cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "mismatches 0, errors 0" in r.stdout


@pytest.mark.parametrize("rec", [2])
def test_more_than_510_bands_per_lattice(hip, oracle, rec):
    """The hand-off tags were one byte, 1 + (band >> 1) mod 255, until round 3: two bands 510 apart on the same ring slot shared a
    tag, and on a diagonal lattice (whose rows do not cover every u) a stale entry could have passed for fresh.  The
    tag is a 16-bit count now.  A 40 x 33000 tile with 128 disparities has 516 bands of 32 rows on its diagonal lattices and 1032
    on the vertical axis ones: bit-exact against the oracle, tags beyond the old wrap included."""
    H, W, dmin, dmax = 33000, 40, -64, 63
    im1, im2 = synth_pair(411, H, W, lambda x, y: 20 * np.sin(x / 23.) * np.cos(y / 190.))
    r = hip.census_sgm(im1, im2, dmin, dmax, params=hip.default_census_params(recursion=rec), want_conf=False)
    o = oracle.oracle_census_sgm(im1, im2, dmin, dmax, params=oracle.census_params(recursion=rec))
    assert same(r["disp"], o["disp"]) and np.array_equal(r["mask"], o["mask"])
    assert np.isfinite(r["disp"]).mean() > 0.5
