"""The reference's end-to-end acceptance tests on the GPU path: the tiles of input_pair /
input_triplet through tiles.process_queue(algo='mgm') -- one s2p_hip_tile_host call per tile: rectify, match, mask,
triangulate -- then the host mirrors of the tail (height_transfer, cargarse_basura, merge_n, height_map_to_lonlatalt,
remove_isolated_3d_points, plyflatten), compared with the rasters the reference holds under its own compare_dsm
(tests/end2end_test.py:21-55).  The figures go to gpurun_out/e2e_compare_dsm.json (round 3's run is committed as profiles/r03/e2e_compare_dsm.json)."""
import json
import os

import numpy as np
import pytest

import e2e
from helpers import same

pytestmark = pytest.mark.gpu
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")


def _record(key, value):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, "e2e_compare_dsm.json")
    d = json.load(open(path)) if os.path.exists(path) else {}
    d[key] = value
    json.dump(d, open(path, "w"), indent=1, sort_keys=True)


@pytest.mark.parametrize("recursion", [2, 1, 0])
def test_pair_dsm(recursion):
    """input_pair -> dsm.tif: |mean| <= 0.025 m, p99 <= 1 m, count within 1 %, same grid -- in the drop-in's mode (MGM
    recursion with three predecessors), with two predecessors, and in the 8-path preview mode."""
    fx = e2e.load("e2e_pair")
    origin, dsm, disps = e2e.run_pair(fx, e2e.Hip(recursion=recursion))
    r = e2e.compare_dsm(dsm, fx["dsm"], 0.025, 1.0)
    _record("pair_dsm_recursion%d" % recursion, r)
    print("pair dsm, recursion", recursion, r)
    assert tuple(origin) == tuple(fx["dsm_origin"])
    assert r["ok"], r


def test_pair_tiles_equal_the_oracle(oracle):
    """The GPU disparity maps of two of the four tiles are the oracle's, bit for bit (rectification included)."""
    fx = e2e.load("e2e_pair")
    hip, cpu = e2e.Hip(recursion=2), e2e.Cpu(recursion=2)
    rp = [hip.rpc(fx["rpc_0"]), hip.rpc(fx["rpc_1"])]
    rc = [cpu.rpc(fx["rpc_0"]), cpu.rpc(fx["rpc_1"])]
    jh, jc = e2e._jobs(fx, hip, 1, rp, 0), e2e._jobs(fx, cpu, 1, rc, 0)
    got = hip.run_tiles([jh[1], jh[2]])
    want = cpu.run_tiles([jc[1], jc[2]])
    for (d, m, lla), (do, mo, llo) in zip(got, want):
        assert same(d, do) and np.array_equal(m, mo)
        assert same(lla, llo)


@pytest.mark.parametrize("recursion", [2, 1, 0])
def test_triplet_height_map_and_dsm(recursion):
    """input_triplet -> pair_1/height_map.tif mosaic and dsm.tif: |mean| <= 0.05 m, p99 <= 2 m, count within 1 %."""
    fx = e2e.load("e2e_triplet")
    out = e2e.run_triplet(fx, e2e.Hip(recursion=recursion))
    r1 = e2e.compare_dsm(out["hm1"], fx["height_map_pair_1"], 0.05, 2.0)
    r2 = e2e.compare_dsm(out["dsm"], fx["dsm"], 0.05, 2.0)
    _record("triplet_height_map_recursion%d" % recursion, r1)
    _record("triplet_dsm_recursion%d" % recursion, r2)
    print("triplet, recursion", recursion, r1, r2)
    if recursion >= 1:                            # the MGM modes carry the assertion; the preview mode is reported
        assert r1["ok"], r1
        assert r2["ok"], r2


def test_triplet_equals_the_oracle_pipeline(oracle):
    """The whole tri-stereo tail on the GPU gives the rasters of the CPU pipeline: height maps and fused maps bit for
    bit, the DSM bit for bit (same clouds in the same order)."""
    fx = e2e.load("e2e_triplet")
    a = e2e.run_triplet(fx, e2e.Hip(recursion=2))
    b = e2e.run_triplet(fx, e2e.Cpu(recursion=2))
    assert same(a["hm1"], b["hm1"])
    assert same(a["fused"], b["fused"])
    assert same(a["dsm"], b["dsm"])


def test_pair_dsm_through_the_file_level_mirrors_in_pool_workers(tmp_path):
    """The reference's end-to-end test on input_pair with the tiles handled the way the reference's orchestrator handles them: a
    forked multiprocessing.Pool whose workers call the FILE-level drop-ins (image_apply_homography, compute_disparity_map('mgm'),
    the triangulation call) -- through the GPU broker, no worker ever creating a HIP context -- and the reference's own compare_dsm
    tolerances on the result; the DSM also agrees with the in-process tile pipeline's (the resampler sees other crop windows, so to
    a tolerance, not bit for bit)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = '''
import sys, os, json
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
os.environ["S2P_HIP_BROKER_DIR"] = os.path.join(%r, "broker")
import numpy as np
import e2e
from s2p_amd import broker
fx = e2e.load("e2e_pair")
be = e2e.FilePool(%r, workers=4)
origin, dsm, disps = e2e.run_pair(fx, be)                    # the Pool forks while this process is still cold
assert be.worker_contexts and all(n == 0 for n in be.worker_contexts), be.worker_contexts
r = e2e.compare_dsm(dsm, fx["dsm"], 0.025, 1.0)
o2, dsm2, disps2 = e2e.run_pair(fx, e2e.Hip(recursion=2))    # the in-process tile pipeline, now that the Pool is gone
both = np.isfinite(dsm) & np.isfinite(dsm2)
diff = np.abs(dsm[both] - dsm2[both])
agree = {"same_origin": tuple(origin) == tuple(o2), "both": float(both.mean()), "either": float((np.isfinite(dsm) | np.isfinite(dsm2)).mean()),
         "p999": float(np.percentile(diff, 99.9)), "max_disp_diff": float(max(np.nanmax(np.abs(a - b)) for a, b in zip(disps, disps2)))}
for c in list(broker._clients.values()): c.close()
broker.shutdown(0)
print("RESULT " + json.dumps({"compare": r, "agree": agree}))
''' % (root, root, str(tmp_path), str(tmp_path))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    import json
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    print(out)
    assert out["compare"]["ok"], out
    a = out["agree"]
    assert a["same_origin"] and a["either"] - a["both"] < 0.002 and a["p999"] < 0.05, a
