"""The drop-in under the reference's own execution model at full size: 16 forked Pool workers of a cold
parent (s2p/parallel.py:76-110) x 20 file-level compute_disparity_map('mgm') calls each on 1024 x 1024 x 128 tiles -- the
multi-PROCESS twin of tools/mgm_stress.py.  k_mgm_bands is a persistent-worker kernel with bounded spin waits; launches of
16 processes share the CUs here.  Every output file (disparity, confidence, mask) must be byte-identical to a quiet
single-process run of the same input, and no worker may raise (a spurious hand-off time-out would surface as HipError)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout=900):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench_pool.py")] + args, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert lines, r.stdout[-2000:] + r.stderr[-4000:]
    return r.returncode, json.loads(lines[-1])


def test_sixteen_processes_of_full_size_mgm_calls_match_a_quiet_run():
    rc, res = _run(["--workers", "16", "--tiles", "320", "--verify"])
    assert res["errors"] == 0, res
    assert res["verify"]["outputs_compared"] == 320 and res["verify"]["different_from_quiet_run"] == 0, res
    assert rc == 0
    assert res["pools"][0]["workers_used"] >= 8, res        # the Pool really spread the calls over its processes


@pytest.mark.parametrize("algo,size,ndisp", [("sgbm", 512, 64), ("mgm_multi", 512, 192)])
def test_pool_of_other_matchers(algo, size, ndisp):
    rc, res = _run(["--workers", "6", "--tiles", "72", "--verify", "--algo", algo, "--size", str(size), "--ndisp", str(ndisp)])
    assert rc == 0 and res["errors"] == 0 and res["verify"]["different_from_quiet_run"] == 0, res
