"""Shared helpers for the tests (synthetic pairs, golden loading)."""
import glob
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
SHIM_RECURSION = 2        # the aggregation mode the file-level 'mgm' / 'mgm_multi' shim and the tile scheduler run (block_matching.matcher_params)


def golden_names(prefix):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


def same(a, b):
    """Exact equality with NaN == NaN."""
    return a.shape == b.shape and bool(np.array_equal(a, b, equal_nan=True))


def synth_pair(seed, H, W, disp_fn, gain=1.05, nan=False, sigma=1.0):
    """Blurred-noise rectified pair, s2p convention im1(x) <-> im2(x + d). numpy only (no scipy
    dependency on the GPU box path): separable gaussian by explicit convolution."""
    rng = np.random.default_rng(seed)
    pad = 256
    base = rng.uniform(0, 1000, (H, W + 2 * pad))
    r = int(4 * sigma + 0.5)
    k = np.exp(-0.5 * (np.arange(-r, r + 1) / sigma) ** 2)
    k /= k.sum()
    base = np.apply_along_axis(lambda v: np.convolve(np.pad(v, r, mode="reflect"), k, mode="valid"), 1, base)
    base = np.apply_along_axis(lambda v: np.convolve(np.pad(v, r, mode="reflect"), k, mode="valid"), 0, base)
    base = base.astype(np.float32)
    im1 = base[:, pad:pad + W].copy()
    xs = np.arange(W, dtype=np.float64)[None, :] + np.zeros((H, 1))
    ys = np.arange(H, dtype=np.float64)[:, None] + np.zeros((1, W))
    d = disp_fn(xs, ys) + 0 * xs
    src = pad + xs - d
    x0 = np.floor(src).astype(int)
    fr = (src - x0).astype(np.float32)
    rows = np.arange(H)[:, None]
    im2 = (gain * ((1 - fr) * base[rows, x0] + fr * base[rows, x0 + 1])).astype(np.float32)
    if nan:
        im1[rng.uniform(size=im1.shape) < 0.01] = np.nan
        im2[5:9, 10:30] = np.nan
    return im1, im2


def tile_views(seed, size, ndisp, nviews):
    """`nviews` views of one synthetic scene (the generator of bench.py's job workloads, seed = 1000 ty + tx):
    view 0 is the reference, view k sees k x the parallax."""
    amp = 0.3125 * ndisp / max(1, nviews - 1)
    f = lambda x, y: amp * np.sin(2 * np.pi * x / (size / 2.)) * np.cos(2 * np.pi * y / (size / 2.))
    im0, im1 = synth_pair(seed, size, size, f)
    out = [im0, im1]
    for k in range(2, nviews):
        out.append(synth_pair(seed, size, size, lambda x, y, k=k: k * f(x, y))[1])
    return out


def synth_cloud(seed, h, w, gsd=0.5, outliers=0.06, holes=0.05):
    """Gridded (h, w, 3) float64 cloud: a smooth surface sampled every `gsd` metres, isolated outliers, small outlier
    clusters, slanted chains that leave the surface gradually (rescued point by point) and NaN holes."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    z = 5 * np.sin(xx / 9.0) * np.cos(yy / 7.0) + rng.normal(0, 0.05, (h, w))
    xyz = np.stack([500000.0 + gsd * xx + rng.normal(0, 0.02, (h, w)), 4000000.0 - gsd * yy + rng.normal(0, 0.02, (h, w)), z], axis=2)
    out = rng.uniform(size=(h, w)) < outliers
    xyz[out, 2] += rng.choice([-1, 1], out.sum()) * rng.uniform(3, 30, out.sum())
    for _ in range(max(1, h * w // 400) if min(h, w) > 3 else 0):    # 2x2 .. 3x3 clusters far from the surface
        y0, x0, k = rng.integers(0, h - 3), rng.integers(0, w - 3), rng.integers(2, 4)
        xyz[y0:y0 + k, x0:x0 + k, 2] += 15.0
    for _ in range(max(1, h * w // 800)):                       # ramps: each step 0.4 m higher than the previous one
        y0, x0, L = rng.integers(0, h), rng.integers(0, max(1, w - 12)), rng.integers(4, 12)
        xyz[y0, x0:x0 + L, 2] = xyz[y0, x0, 2] + 0.4 * np.arange(len(xyz[y0, x0:x0 + L, 2]))
    xyz[rng.uniform(size=(h, w)) < holes] = np.nan
    return xyz


class DevMem:
    """Minimal device buffers through the HIP runtime the library itself uses (ctypes on
    libamdhip64): the tests stay independent of torch, whose wheel bundles a second HIP runtime."""

    def __init__(self):
        import ctypes
        self.ct = ctypes
        self.rt = ctypes.CDLL("libamdhip64.so.7")
        self.ptrs = []

    def upload(self, a):
        p = self.ct.c_void_p()
        assert self.rt.hipMalloc(self.ct.byref(p), self.ct.c_size_t(a.nbytes)) == 0
        assert self.rt.hipMemcpy(p, a.ctypes.data_as(self.ct.c_void_p), self.ct.c_size_t(a.nbytes), 1) == 0
        self.ptrs.append(p)
        return p

    def download(self, p, shape, dtype):
        out = np.empty(shape, dtype)
        assert self.rt.hipMemcpy(out.ctypes.data_as(self.ct.c_void_p), p, self.ct.c_size_t(out.nbytes), 2) == 0
        return out

    def fill(self, p, nbytes, byte):
        assert self.rt.hipMemset(p, byte, self.ct.c_size_t(nbytes)) == 0

    def free(self):
        for p in self.ptrs:
            self.rt.hipFree(p)


def config2_tiles(hmargin=44, vmargin=5, tile=512):
    """BASELINE.json configs[2] on the data the reference's tests hold (tests/golden/input_pair.npz): the full
    1024 x 1024 ROI of img_01 tiled `tile` x `tile` in the image (s2p tiles in image space, config tile_size), every
    tile rectified into the frame of the stored homographies H_ref / H_sec shifted by INTEGER amounts -- so the stored
    rectified_disp.tif (origin of that frame) overlays every tile pixel to pixel -- with the margins the stored tile
    has (44 / 5 px).  Returns [(x0, y0, fx0, fy0, w, h, H1, H2)] and the fixture dict; (fx0, fy0) = position of the
    rectified tile in the stored frame."""
    g = load_golden("input_pair")
    Href, Hsec = g["H_ref"], g["H_sec"]
    H, W = g["img_01"].shape
    out = []
    for y0 in range(0, H, tile):
        for x0 in range(0, W, tile):
            c = np.array([[x0, y0, 1], [x0 + tile, y0, 1], [x0, y0 + tile, 1], [x0 + tile, y0 + tile, 1]], np.float64).T
            p = Href @ c
            p = p[:2] / p[2]
            fx0, fx1 = int(np.floor(p[0].min())) - hmargin, int(np.ceil(p[0].max())) + hmargin
            fy0, fy1 = int(np.floor(p[1].min())) - vmargin, int(np.ceil(p[1].max())) + vmargin
            T = np.array([[1, 0, -fx0], [0, 1, -fy0], [0, 0, 1]], np.float64)
            out.append((x0, y0, fx0, fy0, fx1 - fx0, fy1 - fy0, T @ Href, T @ Hsec))
    return out, g


def overlap_agreement(d, fx0, fy0, d_ref):
    """Agreement of a rectified tile's disparity map `d` (placed at (fx0, fy0) of the stored frame) with the stored
    map on their overlap: (fraction of commonly valid pixels within 0.5 px, within 1 px, number of common pixels)."""
    Hs, Ws = d_ref.shape
    h, w = d.shape
    ys0, ys1, xs0, xs1 = max(0, fy0), min(Hs, fy0 + h), max(0, fx0), min(Ws, fx0 + w)
    if ys1 <= ys0 or xs1 <= xs0:
        return None
    a, b = d_ref[ys0:ys1, xs0:xs1], d[ys0 - fy0:ys1 - fy0, xs0 - fx0:xs1 - fx0]
    both = np.isfinite(a) & np.isfinite(b)
    if not both.any():
        return None
    e = np.abs(a[both] - b[both])
    return float((e <= 0.5).mean()), float((e <= 1.0).mean()), int(both.sum())
