"""Shared helpers for the tests (synthetic pairs, golden loading)."""
import glob
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")


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
