"""The untouched orchestrator forks one worker per tile x pair (s2p/parallel.py:76-98: multiprocessing.Pool, fork start
method) AFTER the package was imported, and every worker calls compute_disparity_map on files.  The boundary's contract
(SURVEY.md 8b): nothing touches the HIP runtime at import or dlopen time, each child creates its own context lazily, and
children spread over the visible GPUs (device = pid mod count unless S2P_HIP_DEVICE / LOCAL_RANK says otherwise).

Two cases: a parent that has not used the GPU (the orchestrator's situation), and a parent that already created a
context before forking -- the HIP runtime does not survive fork, so the library must detect the inherited state and
refuse loudly in the child rather than hang or corrupt (create contexts after the fork)."""
import multiprocessing as mp
import os
import sys

import numpy as np
import pytest

from helpers import same, synth_pair

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write_pair(d, k):
    from s2p_amd import io as rio
    im1, im2 = synth_pair(300 + k, 96, 160, lambda x, y: 6 + 9 * np.sin(x / 37.) * np.cos(y / 29.))
    p1, p2 = os.path.join(d, "ref_%d.tif" % k), os.path.join(d, "sec_%d.tif" % k)
    rio.write_image(p1, im1)
    rio.write_image(p2, im2)
    return p1, p2


def _child(args):
    """What s2p.stereo_matching does inside a Pool worker (s2p/__init__.py:166-196)."""
    d, k, algo = args
    from s2p_amd import _lib, block_matching as bm
    from s2p_amd import io as rio
    disp, mask = os.path.join(d, "disp_%d_%s.tif" % (k, algo)), os.path.join(d, "mask_%d_%s.png" % (k, algo))
    sys.stdout = open(os.devnull, "w")                       # workers redirect stdout to the tile log (parallel.py:37-40)
    bm.compute_disparity_map(os.path.join(d, "ref_%d.tif" % k), os.path.join(d, "sec_%d.tif" % k), disp, mask, algo, -24, 39, timeout=600)
    return k, algo, os.getpid(), _lib.default_device(), _lib.device_count(), rio.read_image(disp), rio.read_image(mask, np.uint8)


def _expected(k, algo):
    from s2p_amd import _lib
    from s2p_amd.block_matching import matcher_params
    im1, im2 = synth_pair(300 + k, 96, 160, lambda x, y: 6 + 9 * np.sin(x / 37.) * np.cos(y / 29.))
    kind, p = matcher_params(algo)
    r = (_lib.sgbm if kind == "sgbm" else _lib.census_sgm)(im1, im2, -24, 39, params=p)
    return r["disp"], r["mask"]


def _run_in_fresh_interpreter(body):
    """The fork tests need a parent process whose HIP state is known: each runs in its own interpreter."""
    import subprocess
    code = "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n" % (ROOT, os.path.join(ROOT, "tests")) + body
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return r.stdout


def test_forked_pool_workers_with_a_cold_parent(tmp_path):
    out = _run_in_fresh_interpreter('''
import multiprocessing as mp, os, numpy as np
import s2p_amd                                            # imported BEFORE the fork, like the orchestrator does
from s2p_amd import _lib
import test_gpu_fork as t
d = %r
jobs = [(d, k, algo) for k in range(4) for algo in ("mgm", "sgbm")]
for k in range(4):
    t._write_pair(d, k)
_lib.lib()                                                # dlopen in the parent: must not initialise HIP
ctx = mp.get_context("fork")
with ctx.Pool(4) as pool:
    res = pool.map(t._child, jobs)
pids = sorted({r[2] for r in res})
assert len(pids) >= 2, pids
for k, algo, pid, dev, ndev, disp, mask in res:
    assert dev == pid %% ndev                                # workers spread over the visible GPUs
    want_d, want_m = t._expected(k, algo)                    # parent computes AFTER the pool: its own context, created now
    assert t.same(want_d, disp) and np.array_equal(want_m, mask), (k, algo)
print("ok", len(res), "results from", len(pids), "forked workers")
''' % str(tmp_path))
    assert "ok 8 results" in out


def test_fork_after_the_parent_used_the_gpu(tmp_path):
    """A parent that already holds a context forks: the children must not reuse the inherited (dead) runtime state
    silently.  They either work on a fresh context of their own or fail with HipError -- never hang, never return wrong
    numbers.  (ROCm does not support using HIP in a forked child of an initialised parent: a loud failure is the
    contract; the orchestrator never does this, it forks first.)"""
    out = _run_in_fresh_interpreter('''
import multiprocessing as mp, os, numpy as np
from s2p_amd import _lib
import test_gpu_fork as t
d = %r
t._write_pair(d, 0)
want_d, want_m = t._expected(0, "mgm")                    # the parent creates its context here
def child(_):
    try:
        k, algo, pid, dev, ndev, disp, mask = t._child((d, 0, "mgm"))
        return "same" if (t.same(want_d, disp) and np.array_equal(want_m, mask)) else "DIFFERENT"
    except Exception as e:
        return "error: %%s" %% type(e).__name__
ctx = mp.get_context("fork")
with ctx.Pool(2) as pool:
    r = pool.map_async(child, [0, 1])
    try:
        res = r.get(timeout=120)
    except mp.TimeoutError:
        res = ["HANG"]
print("children:", res)
assert all(x == "same" or x.startswith("error") for x in res), res
again_d, _ = t._expected(0, "mgm")                        # and the parent's own context still works
assert t.same(want_d, again_d)
print("ok")
''' % str(tmp_path))
    assert "ok" in out
