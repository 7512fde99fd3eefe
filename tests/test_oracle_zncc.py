"""The ZNCC cost of the oracle (oracle/census_oracle.c: zncc_stats / zncc_cost; north_star's "census/ZNCC") against a reference-free
statement of the same definition: no call site of the reference reaches a ZNCC cost (`-t census` is hard-coded, s2p/block_matching.py:171,
:293), so nothing the reference holds can pin it -- what CAN be pinned is that the oracle computes the zero-mean normalised
cross-correlation it claims to: equal to a float64 numpy evaluation of the textbook formula up to the rounding of its float32 arithmetic
at the quantisation steps, invariant under gain and offset of either image, 0 for identical windows, and on the census scale 0..24."""
import numpy as np

from helpers import synth_pair


def zncc_volume_f64(im1, im2, dmin, dmax, win=5):
    """cost[y, x, i] = clamp(floor((1 - zncc) * 12 + 0.5), 0, 24) for the windows centred at (x, y) of image 1 and (x + dmin + i, y) of
    image 2, borders replicated; 255 where the candidate's centre lies outside image 2 (float64 throughout)."""
    h, w = im1.shape
    r = win // 2
    a = np.pad(im1.astype(np.float64), r, mode="edge")
    b = np.pad(im2.astype(np.float64), r, mode="edge")
    wa = np.lib.stride_tricks.sliding_window_view(a, (win, win)).reshape(h, w, win * win)
    wb = np.lib.stride_tricks.sliding_window_view(b, (win, win)).reshape(h, w, win * win)
    ca = wa - wa.mean(axis=2, keepdims=True)
    cb = wb - wb.mean(axis=2, keepdims=True)
    va, vb = (ca * ca).sum(axis=2), (cb * cb).sum(axis=2)
    D = dmax - dmin + 1
    out = np.full((h, w, D), 255.0)
    z_all = np.full((h, w, D), np.nan)
    for i in range(D):
        d = dmin + i
        x = np.arange(w)
        ok = (x + d >= 0) & (x + d < w)
        xs = x[ok]
        cov = (ca[:, xs, :] * cb[:, xs + d, :]).sum(axis=2)
        den = va[:, xs] * vb[:, xs + d]
        z = np.where(den > 0, cov / np.sqrt(np.where(den > 0, den, 1.0)), 0.0)
        z_all[:, xs, i] = z
        out[:, xs, i] = np.clip(np.floor((1.0 - z) * 12.0 + 0.5), 0, 24)
    return out, z_all


def _oracle_C(oracle, im1, im2, dmin, dmax, **kw):
    o = oracle.oracle_census_sgm(im1, im2, dmin, dmax, params=oracle.census_params(cost=1, recursion=0, **kw), dump="full")
    assert o["rc"] == 0
    return o["C"][:, :, : dmax - dmin + 1].astype(np.float64), o["dmin0"]


def test_the_oracle_is_the_textbook_zncc(oracle):
    for seed, (h, w, dmin, dmax, win) in enumerate([(40, 72, -9, 8, 5), (25, 50, -3, 12, 3), (33, 64, 0, 15, 5)]):
        im1, im2 = synth_pair(400 + seed, h, w, lambda x, y: 0.5 * (dmin + dmax) + 3 * np.sin(x / 17.) * np.cos(y / 13.))
        C, d0 = _oracle_C(oracle, im1, im2, dmin, dmax, census_win=win)
        assert d0 == dmin
        want, z = zncc_volume_f64(im1, im2, dmin, dmax, win)
        assert np.array_equal(C == 255, want == 255)                       # the same candidates are excluded (centre outside image 2)
        inside = want != 255
        diff = np.abs(C - want)[inside]
        assert diff.max() <= 1                                             # float32 against float64: only a quantisation step apart, and only ...
        frac = ((1.0 - z[inside]) * 12.0 + 0.5) % 1.0                      # ... where (1 - z) * 12 + 0.5 sits within float32 noise of an integer
        assert np.all(np.minimum(frac, 1.0 - frac)[diff > 0] < 1e-3)
        assert (diff > 0).mean() < 2e-3
        assert C[inside].min() >= 0 and C[inside].max() <= 24


def test_half_pixel_candidates_correlate_with_the_half_sampled_image(oracle):
    """subpix = 2: candidate j stands for dmin + j / 2; the even ones are the whole-pixel volume, the odd ones the same correlation against
    image 2 sampled half way between its columns, 0.5 (b[x] + b[min(x + 1, w - 1)]) -- the half-pixel grid of the census cost."""
    im1, im2 = synth_pair(421, 30, 70, lambda x, y: 1.5 + 3 * np.sin(x / 15.))
    dmin, dmax = -6, 7
    o = oracle.oracle_census_sgm(im1, im2, dmin, dmax, params=oracle.census_params(cost=1, recursion=0, subpix=2), dump="full")
    assert o["rc"] == 0 and o["dmin0"] == dmin
    Dt = 2 * (dmax - dmin) + 1
    C = o["C"].astype(np.float64)
    assert np.all(C[:, :, Dt:] == 255)                                     # the padding up to a multiple of 16
    whole, _ = zncc_volume_f64(im1, im2, dmin, dmax)
    im2h = (0.5 * (im2 + np.concatenate([im2[:, 1:], im2[:, -1:]], axis=1))).astype(np.float32)
    half, _ = zncc_volume_f64(im1, im2h, dmin, dmax)
    ev, od = C[:, :, 0:Dt:2], C[:, :, 1:Dt:2]
    for got, want in ((ev, whole), (od, half[:, :, : od.shape[2]])):
        assert np.array_equal(got == 255, want == 255)
        d = np.abs(got - want)[want != 255]
        assert d.max() <= 1 and (d > 0).mean() < 2e-3


def test_gain_and_offset_do_not_move_the_cost(oracle):
    im1, im2 = synth_pair(411, 48, 96, lambda x, y: 2 + 4 * np.sin(x / 21.))
    C0, _ = _oracle_C(oracle, im1, im2, -8, 8)
    for g1, o1, g2, o2 in ((1.0, 0.0, 0.5, 300.0), (3.0, -50.0, 1.0, 0.0), (0.25, 10.0, 4.0, -200.0)):
        C1, _ = _oracle_C(oracle, (g1 * im1 + o1).astype(np.float32), (g2 * im2 + o2).astype(np.float32), -8, 8)
        assert np.array_equal(C0 == 255, C1 == 255)
        d = np.abs(C0 - C1)[C0 != 255]
        assert d.max() <= 1 and (d > 0).mean() < 0.01                      # (rounding of the scaled float32 images only)


def test_identical_windows_cost_nothing_and_inverted_ones_everything(oracle):
    rng = np.random.default_rng(7)
    im = rng.uniform(0, 1000, (30, 60)).astype(np.float32)
    C, _ = _oracle_C(oracle, im, im, -2, 2)
    assert np.all(C[:, :, 2] == 0)                                         # d = 0: zncc = 1
    Cn, _ = _oracle_C(oracle, im, (1000.0 - im).astype(np.float32), -2, 2)
    assert np.all(Cn[:, :, 2] == 24)                                       # d = 0 against the negative: zncc = -1
    flat = np.full((12, 40), 5.0, np.float32)
    Cf, _ = _oracle_C(oracle, flat, flat, -1, 1)
    assert np.all(Cf[Cf != 255] == 12)                                     # no variance: zncc defined as 0, the middle of the scale
