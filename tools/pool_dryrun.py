#!/usr/bin/env python3
"""tools/pool_dryrun.py -- the HOST side of a node-wide Pool run without any GPU.

What `bench.py --workload pool --gpus 8` / a real s2p job on an 8-GPU node asks of the host -- 64 forked workers per device, one broker
per device, one memfd arena per worker, TIFFs in /dev/shm -- is exercised here with N stand-in brokers: real `broker.Server` processes
(framing, descriptor passing, arenas, queueing, batching, replies: all the product's code) whose lanes run a numpy stub instead of
libs2p_hip (`Server(backend=...)`, the hook tests/test_broker_protocol.py uses) and sleep what the device would take per tile.  The
workers are bench_pool.py's own: `compute_disparity_map('mgm')` on float32 TIFFs, device = pid mod N through the brokers' hello.

So the first multi-GPU run cannot fail for a reason that has nothing to do with GPUs: process and descriptor limits, /dev/shm space,
socket back-logs, workers that all land on one device, a broker that misses its workers.  Prints one JSON line: per-broker request
counts, workers used, descriptors and shared memory at the peak, the limits they were checked against.

    python tools/pool_dryrun.py                       # 8 brokers x 64 workers, 256 x 256 tiles
    python tools/pool_dryrun.py --gpus 8 --workers-per-gpu 8 --size 128 --tiles-per-worker 3      # what tests/test_pool_dryrun.py runs
"""
import argparse
import json
import multiprocessing as mp
import os
import resource
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


class StubBackend:
    """disp = im1 - im2, mask = im1 > 0, conf = 1; `ms_per_tile` of sleep per tile of a call stands for the device."""

    def __init__(self, ndev, ms_per_tile):
        self.ndev, self.ms, self.pins, self.pinned_bytes = ndev, ms_per_tile, 0, 0

    def start(self, device, nlanes):
        return self.ndev

    def pin(self, addr, size):
        self.pins += 1
        self.pinned_bytes += size
        return True

    def unpin(self, addr):
        self.pins -= 1

    def run(self, lane, grp, tmo, cap=1):
        time.sleep(self.ms * 1e-3 * len(grp))
        for r in grp:
            m = r.msg
            v = lambda k, dt: r.arena.plane(m["off"][k], (m["h"], m["w"]), dt)
            v("disp", np.float32)[:] = v("im1", np.float32) - v("im2", np.float32)
            v("mask", np.uint8)[:] = v("im1", np.float32) > 0
            if m["op"] == "census" and "conf" in m["off"]:
                v("conf", np.float32)[:] = 1.0


def serve(device, ndev, ms_per_tile):
    from s2p_amd import broker
    broker._serving[0] = True
    be = StubBackend(ndev, ms_per_tile)
    srv = broker.Server(device, lanes=3, max_batch=8, idle_s=600.0, max_wait_ms=3.0, backend=be)
    srv.serve()


def _fds(pid):
    try:
        return len(os.listdir("/proc/%d/fd" % pid))
    except OSError:
        return -1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=8, help="stand-in brokers (devices)")
    ap.add_argument("--workers-per-gpu", type=int, default=64)
    ap.add_argument("--size", type=int, default=256, help="tile width = height (a full-size 1024 x 1024 run needs 36 MB of arena per worker)")
    ap.add_argument("--ndisp", type=int, default=64)
    ap.add_argument("--tiles-per-worker", type=int, default=6)
    ap.add_argument("--ms-per-tile", type=float, default=0.7, help="what a stand-in lane sleeps per tile (the resident rate of one MI355X)")
    ap.add_argument("--serve", type=int, default=None, help=argparse.SUPPRESS)
    a = ap.parse_args()
    if a.serve is not None:
        return serve(a.serve, a.gpus, a.ms_per_tile)

    P = a.gpus * a.workers_per_gpu
    lim_nofile = resource.getrlimit(resource.RLIMIT_NOFILE)
    lim_nproc = resource.getrlimit(resource.RLIMIT_NPROC)
    shm = "/dev/shm"
    shm_ok = os.path.isdir(shm) and shutil.disk_usage(shm).free > (1 << 30)
    base = tempfile.mkdtemp(prefix="s2p_dryrun_", dir=shm if shm_ok else None)
    bdir = os.path.join(base, "broker")
    os.makedirs(bdir, mode=0o700)
    os.environ["S2P_HIP_BROKER_DIR"] = bdir
    os.environ["S2P_HIP_BROKER"] = "1"
    for k in ("S2P_HIP_DEVICE", "LOCAL_RANK", "RANK", "WORLD_SIZE"):
        os.environ.pop(k, None)
    import bench_pool as bp
    import s2p_amd                                            # noqa: F401
    from s2p_amd import broker
    res = {"what": "host-side dry run of a node-wide Pool: %d stand-in brokers (real broker.Server processes, numpy lanes sleeping %.2f ms per tile) x "
                   "%d forked workers each = %d workers, compute_disparity_map('mgm') on %dx%d float32 TIFFs in %s"
                   % (a.gpus, a.ms_per_tile, a.workers_per_gpu, P, a.size, a.size, base),
           "host": {"cpus": os.cpu_count(), "rlimit_nofile": list(lim_nofile), "rlimit_nproc": list(lim_nproc),
                    "shm_free_GB": round(shutil.disk_usage(base).free / 1e9, 1)}}
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--serve", str(d), "--gpus", str(a.gpus), "--ms-per-tile", str(a.ms_per_tile)],
                              env=env, stdout=subprocess.DEVNULL, stderr=open(os.path.join(bdir, "gpu%d.err" % d), "wb")) for d in range(a.gpus)]
    rc = 1
    try:
        deadline = time.monotonic() + 60
        while not all(os.path.exists(broker.sock_path(d)) for d in range(a.gpus)):
            if time.monotonic() > deadline or any(p.poll() is not None for p in procs):
                raise RuntimeError("a stand-in broker did not come up: " + " | ".join(open(os.path.join(bdir, "gpu%d.err" % d), "rb").read().decode()[-300:] for d in range(a.gpus)))
            time.sleep(0.02)
        inputs = bp.write_inputs(base, 0, a.size, a.ndisp, 8)
        dmin, dmax = -a.ndisp // 2, a.ndisp // 2 - 1
        n = P * a.tiles_per_worker
        tasks = [(i, inputs[i % 8][0], inputs[i % 8][1], base, "mgm", dmin, dmax, False, False) for i in range(n)]
        # fork -> join with a sampler of the brokers' descriptors and of the shared memory in the middle
        ctx = mp.get_context("fork")
        t0 = time.monotonic()
        pool = ctx.Pool(P)
        rs = [pool.apply_async(bp.task, (t,)) for t in tasks]
        peak_fds, peak_shm = 0, 0
        while not all(r.ready() for r in rs):
            peak_fds = max([peak_fds] + [_fds(p.pid) for p in procs])
            with open("/proc/meminfo") as f:
                for line in f:
                    if line.startswith("Shmem:"):
                        peak_shm = max(peak_shm, int(line.split()[1]))
            time.sleep(0.05)
        out = [r.get(60) for r in rs]
        pool.close()
        pool.join()
        wall = time.monotonic() - t0
        stats = [broker.stats(d) for d in range(a.gpus)]
        served = [int(s.get("requests", 0)) for s in stats]
        pids = {r[1] for r in out}
        by_dev = {}
        for pid in pids:
            by_dev[pid % a.gpus] = by_dev.get(pid % a.gpus, 0) + 1
        res.update({"workers": P, "workers_used": len(pids), "workers_per_device": [by_dev.get(d, 0) for d in range(a.gpus)],
                    "tiles": n, "wall_s": round(wall, 2), "tiles_per_s_fork_to_join": round(n / wall, 1),
                    "requests_per_broker": served, "calls_per_broker": [int(s.get("calls", 0)) for s in stats],
                    "errors_per_broker": [int(s.get("errors", 0)) for s in stats],
                    "arenas_attached_per_broker": [int(s.get("attached", 0)) for s in stats],
                    "peak_fds_of_a_broker": peak_fds, "peak_Shmem_GB": round(peak_shm / 1e6, 2),
                    "arena_bytes_per_worker_at_1024x1024": 36 << 20,
                    "projected_arena_GB_at_full_size": round(P * 36 * 2 ** 20 / 1e9, 1)})
        ok = (sum(served) == n and all(s > 0 for s in served) and sum(res["errors_per_broker"]) == 0 and len(pids) >= P // 2
              and (lim_nofile[0] < 0 or peak_fds < 0.5 * lim_nofile[0]))
        res["ok"] = bool(ok)
        rc = 0 if ok else 1
    except Exception as e:
        res["error"] = repr(e)[:500]
    finally:
        for d in range(a.gpus):
            try:
                if os.path.exists(broker.sock_path(d)):
                    broker.shutdown(d)
            except Exception:
                pass
        for p in procs:
            try:
                p.wait(timeout=10)
            except Exception:
                p.kill()
        shutil.rmtree(base, ignore_errors=True)
    print(json.dumps(res), flush=True)
    return rc


if __name__ == "__main__":
    sys.exit(main())
