"""The host side of a node-wide Pool run without a GPU: 8 stand-in brokers -- real broker.Server processes, numpy lanes
-- x forked workers calling the file-level drop-in (tools/pool_dryrun.py).  What the first 8-GPU run must not die on: workers that
all land on one device, a broker that misses its workers, descriptors, shared memory."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_eight_brokers_serve_the_workers_of_one_pool():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pool_dryrun.py"), "--gpus", "8", "--workers-per-gpu", "6", "--size", "128",
                        "--ndisp", "32", "--tiles-per-worker", "4"], capture_output=True, text=True, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert lines, r.stdout[-1000:] + r.stderr[-3000:]
    d = json.loads(lines[-1])
    assert r.returncode == 0 and d.get("ok"), d
    assert d["workers"] == 48 and d["tiles"] == 192 and sum(d["requests_per_broker"]) == 192
    assert all(n > 0 for n in d["requests_per_broker"]) and sum(d["errors_per_broker"]) == 0, d
    assert d["workers_per_device"] == [6] * 8, d                      # consecutive pids of one fork Pool: pid mod 8 spreads them evenly
    # a worker keeps ONE broker connection (its own device's): the count it asked device 0 for does not stay open there
    assert d["peak_fds_of_a_broker"] < 6 * 6 + 40, d
