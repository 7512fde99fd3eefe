"""GPU parity tests of the height-map merge (fusion.merge_n, s2p/fusion.py:26-68) through the C ABI: bit-exact
float32 output against the fixture made with the reference's own average_if_close and against numpy running
the reference's apply_along_axis formulation for every supported operator."""
import os

import numpy as np
import pytest

from helpers import load_golden, same

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from s2p_amd import _lib
    assert _lib.device_count() > 0, "no MI355X visible: the HIP path has no fallback"
    return _lib


def test_fixture_from_the_reference_function(hip):
    g = load_golden("fusion_stack")
    for k in range(4):
        out = hip.merge_n(list(g["stack%d" % k]), list(g["offsets%d" % k]), "average_if_close", float(g["threshold%d" % k]))
        assert same(out, g["expected%d" % k]), k


def stack(seed, n, shape=(41, 67), nan=0.2):
    rng = np.random.default_rng(seed)
    base = rng.uniform(-50, 300, shape)
    st = [(base + rng.normal(0, 1.5, shape) * 10.0 ** rng.integers(-3, 2) + 3.0 * i).astype(np.float32) for i in range(n)]
    for a in st:
        a[rng.uniform(size=shape) < nan] = np.nan
        a[rng.uniform(size=shape) < 0.01] = np.inf
        a[rng.uniform(size=shape) < 0.01] = -np.inf
        a[rng.uniform(size=shape) < 0.01] = -0.0
    st[0][0, :] = np.nan
    for a in st:
        a[1, :5] = np.nan
    return st, [3.0 * i + float(rng.normal(0, 0.4)) for i in range(n)]


@pytest.mark.parametrize("op", ["average_if_close", "np.nanmedian", "np.median", "np.nanmean", "np.mean",
                                "np.nanmin", "np.nanmax", "np.min", "np.max"])
@pytest.mark.parametrize("n", [1, 2, 3, 4, 7, 8, 9, 17, 33, 64])
def test_every_operator_against_numpy(hip, oracle, op, n):
    st, offs = stack(100 + n, n, nan=0.2 if n < 16 else 0.5)
    out = hip.merge_n(st, offs, op, threshold=4.0)
    ref = oracle.oracle_merge_n(st, offs, op, 4.0)
    assert same(out, ref)


def test_numpy_prefix_and_bad_arguments(hip):
    st, offs = stack(5, 3)
    assert same(hip.merge_n(st, offs, "numpy.nanmean"), hip.merge_n(st, offs, "np.nanmean"))
    with pytest.raises(ValueError):
        hip.merge_n(st, offs, "np.std")
    st65, offs65 = stack(6, 65, shape=(4, 4))
    with pytest.raises(hip.HipError) as e:
        hip.merge_n(st65, offs65)
    assert e.value.code == hip.BAD_ARGUMENT


def test_file_level_mirror(hip, oracle, tmp_path):
    """s2p_amd.fusion.merge_n with the reference's signature: paths in, float32 TIFF out."""
    from s2p_amd import fusion, io as rio
    st, offs = stack(9, 3, nan=0.1)
    for a in st:                                              # file formats carry NaN, not inf
        a[~np.isfinite(a)] = np.nan
    paths = []
    for i, a in enumerate(st):
        p = os.path.join(tmp_path, "height_map_%d.tif" % i)
        rio.write_image(p, a)
        paths.append(p)
    out = os.path.join(tmp_path, "height_map.tif")
    fusion.merge_n(out, paths, offs, averaging="average_if_close", threshold=3, debug=True)
    assert same(rio.read_image(out), oracle.oracle_merge_n(st, offs, "average_if_close", 3))
    assert os.path.exists(os.path.join(tmp_path, "height_map_0_registered.tif"))


def test_config4_tristereo_per_pair_sgm_then_merge(hip, oracle):
    """BASELINE.json configs[4] in miniature: three views -> pairs (0,1) and (0,2) (the reference pairs image 0 with
    every other one, s2p/__init__.py:561-562), per-pair SGM on sharded tiles, then the per-pixel fusion of the two
    registered maps -- every stage bit-exact against the oracles chained the same way."""
    from helpers import synth_pair
    from s2p_amd import tiles as T
    H, W = 96, 160
    field = lambda x, y: 6 * np.sin(x / 23.) * np.cos(y / 19.)
    im0, im1 = synth_pair(71, H, W, field)
    _, im2 = synth_pair(71, H, W, lambda x, y: 2.0 * field(x, y))          # same base image, twice the parallax
    tl = [T.Tile(0, im0, im1, -12, 12), T.Tile(1, im0, im2, -20, 20)]
    res = T.match_tiles(tl, algo="mgm", in_flight=2)                       # the 'mgm' call's parameters: MGM recursion, median
    o1 = oracle.oracle_census_sgm(im0, im1, -12, 12, params=oracle.census_params(recursion=2))["disp"]
    o2 = oracle.oracle_census_sgm(im0, im2, -20, 20, params=oracle.census_params(recursion=2))["disp"]
    assert same(res[0], o1) and same(res[1], o2)
    # "heights": disparity x a per-pair baseline factor; offsets = the per-pair mean heights (s2p/__init__.py:320-353)
    h1, h2 = (res[0] * np.float32(2.0)).astype(np.float32), (res[1] * np.float32(1.0)).astype(np.float32)
    offs = [float(np.nanmean(h1)), float(np.nanmean(h2))]
    fused = hip.merge_n([h1, h2], offs, "average_if_close", threshold=3)
    assert same(fused, oracle.oracle_merge_n([h1, h2], offs, "average_if_close", 3))
    both = np.isfinite(h1) & np.isfinite(h2)
    assert np.isfinite(fused).mean() > 0.5 and np.nanmedian(np.abs(h1 - h2)[both]) < 1.0
