#!/usr/bin/env python3
"""bench_pool.py -- the drop-in under the reference's OWN execution model.

The reference runs every step as `multiprocessing.Pool(nb_workers)` forked from a parent that never touches the matcher
itself, one tile x pair per task, file paths in and out (s2p/parallel.py:76-110, s2p/__init__.py:166-196, 584-591).  This
script does exactly that with `s2p_amd.block_matching.compute_disparity_map` as the task:

    parent (cold: imports the package, never makes a HIP call)
      for P in --workers:  pool = get_context("fork").Pool(P); apply_async(task) x tiles; get(); close(); join()
    task = one compute_disparity_map(rectified_ref.tif, rectified_sec.tif, rectified_disp.tif, rectified_mask.png, algo, dmin, dmax)
           on float32 TIFFs in /dev/shm, stdout redirected like tilewise_wrapper does

and reports per P: tiles/s over the whole Pool (fork -> join: what a step of the reference sees, cold start included), the
cold start of a worker (fork -> its first result: HIP initialisation, code-object load, workspace allocation, first
transfers), and the steady state (tasks finished after every worker had returned its first result and before the last task was
handed out: every worker busy): tiles/s and the mean ms per call split into read / library call / write.  One JSON line on stdout.

    python bench_pool.py --workers 1,4,16,64 --tiles 256
    python bench_pool.py --workers 16 --tiles 320 --verify        # every output compared with a quiet single-process run

Nothing here imports torch; the GPU of a worker is S2P_HIP_DEVICE / LOCAL_RANK / pid mod device count (s2p_amd/_lib.py).
"""
import argparse
import hashlib
import json
import multiprocessing as mp
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def write_inputs(d, k, size, ndisp, distinct, ragged=False):
    """`distinct` seeded rectified pairs (the synthetic pair of SURVEY.md 8(d), seeds 2000 + i) as float32 TIFFs.  ragged: every pair has
    its own size (a few pixels apart, as the rectified tiles of a real job are) -- and the tasks their own disparity ranges."""
    from helpers import synth_pair
    from s2p_amd import io as rio
    amp = 0.3125 * ndisp
    paths = []
    for i in range(distinct):
        w, h = (size - 8 * (i % 8) - i // 8, size - 24 + 4 * (i % 8) + i // 8) if ragged else (size, size)   # within 64 px of `size`, all different
        a, b = synth_pair(2000 + i, h, w,
                          lambda x, y: amp * np.sin(2 * np.pi * x / (size / 2.)) * np.cos(2 * np.pi * y / (size / 2.)))
        p1, p2 = os.path.join(d, "rectified_ref_%d.tif" % i), os.path.join(d, "rectified_sec_%d.tif" % i)
        rio.write_image(p1, a)
        rio.write_image(p2, b)
        paths.append((p1, p2))
    return paths


def _digest(paths):
    h = []
    for p in paths:
        with open(p, "rb") as f:
            h.append(hashlib.blake2b(f.read(), digest_size=12).hexdigest())
    return h


def apply_cfg(spec):
    """key=value[,key=value...] into s2p_amd.config.cfg (ints and floats parsed, everything else a string)."""
    from s2p_amd.config import cfg
    for kv in [x for x in (spec or "").split(",") if x]:
        k, v = kv.split("=", 1)
        for cast in (int, float, str):
            try:
                cfg[k] = cast(v)
                break
            except ValueError:
                pass


def task(args):
    """What s2p.stereo_matching does inside a Pool worker (s2p/__init__.py:166-196): one file-level matcher call."""
    i, im1, im2, out_dir, algo, dmin, dmax, keep, digest = args
    from s2p_amd import block_matching as bm
    disp = os.path.join(out_dir, "rectified_disp_%d.tif" % i)
    mask = os.path.join(out_dir, "rectified_mask_%d.png" % i)
    conf = os.path.join(out_dir, "rectified_disp_%d_confidence.tif" % i)
    fh = float(os.environ.get("S2P_POOL_FAULTHANDLER", "0"))
    if fh > 0:                                            # (diagnosis: where is a worker that does not come back?)
        import faulthandler
        faulthandler.dump_traceback_later(fh, exit=False)
    so = sys.stdout
    sys.stdout = open(os.devnull, "w")                    # tilewise_wrapper sends a worker's prints to the tile's log
    t0 = time.monotonic()                                 # CLOCK_MONOTONIC: one clock for every process of the box
    c0 = time.process_time()                              # user + system CPU of this worker, its encoder threads included
    try:
        bm.compute_disparity_map(im1, im2, disp, mask, algo, dmin, dmax, timeout=600)
    finally:
        sys.stdout.close()
        sys.stdout = so
    t1 = time.monotonic()
    cpu_call = (time.process_time() - c0) * 1e3
    if fh > 0:
        faulthandler.cancel_dump_traceback_later()
    outs = [disp, mask] + ([conf] if algo != "sgbm" else [])
    dg = _digest(outs) if digest else None
    if not keep:
        for p in outs:
            os.unlink(p)
    return i, os.getpid(), t0, t1, dict(bm.last_call_ms, cpu=cpu_call), dg


def run_pool(P, tasks, task_timeout=600):
    """One step of the reference: a fresh fork Pool of P workers over all tasks (s2p/parallel.py:76-110; its r.get(timeout) too)."""
    ctx = mp.get_context("fork")
    t_fork = time.monotonic()
    pool = ctx.Pool(P)
    res = [pool.apply_async(task, (t,)) for t in tasks]
    try:
        out = [r.get(task_timeout) for r in res]
    except BaseException:
        pool.terminate()                                   # (launch_calls does the same on KeyboardInterrupt; a lost worker must not leave 63 others behind)
        pool.join()
        raise
    pool.close()
    pool.join()
    t_end = time.monotonic()
    return t_fork, t_end, out


def cgroup_cpu(root="/sys/fs/cgroup"):
    """(quota in CPUs or None, usage_usec, throttled_usec, nr_throttled) of this process's control group (cgroup v2): the GPU boxes of this pool
    grant 16 CPUs' worth of time to everything a run starts -- the Pool's workers and the broker included."""
    quota, use, thr, nthr = None, None, None, None
    try:
        q, per = open(os.path.join(root, "cpu.max")).read().split()[:2]
        quota = None if q == "max" else float(q) / float(per)
        for line in open(os.path.join(root, "cpu.stat")):
            k, v = line.split()
            if k == "usage_usec":
                use = int(v)
            elif k == "throttled_usec":
                thr = int(v)
            elif k == "nr_throttled":
                nthr = int(v)
    except Exception:
        pass
    return quota, use, thr, nthr


def summarise(P, t_fork, t_end, out):
    n = len(out)
    by_pid = {}
    for r in out:
        by_pid.setdefault(r[1], []).append(r)
    first = {pid: min(r[3] for r in rs) for pid, rs in by_pid.items()}
    cold = sorted(v - t_fork for v in first.values())
    t_warm = max(first.values())
    t_tail = max(r[2] for r in out)                        # when the last task was STARTED: until then no worker ran out of work
    steady = [r for r in out if t_warm < r[3] <= t_tail]
    s = {"workers": P, "workers_used": len(by_pid), "tiles": n, "wall_s": round(t_end - t_fork, 4),
         "tiles_per_s_fork_to_join": round(n / (t_end - t_fork), 1),
         "cold_start_s": {"min": round(cold[0], 3), "median": round(cold[len(cold) // 2], 3), "max": round(cold[-1], 3)}}
    setup = [r[4].get("setup", 0.0) for r in out if r[4].get("setup", 0.0) > 0]
    if setup:
        s["broker_connect_attach_ms"] = {"median": round(float(np.median(setup)), 2), "max": round(float(np.max(setup)), 2)}
    if len(steady) >= max(8, P):
        span = t_tail - t_warm
        s["steady"] = {"tiles": len(steady), "tiles_per_s": round(len(steady) / span, 1), "ms_per_tile": round(span / len(steady) * 1e3, 4),
                       "call_ms": round(float(np.mean([(r[3] - r[2]) * 1e3 for r in steady])), 3),
                       "worker_cpu_ms": round(float(np.mean([r[4].get("cpu", 0.0) for r in steady])), 3),
                       "read_ms": round(float(np.mean([r[4].get("read", 0.0) for r in steady])), 3),
                       "gpu_ms": round(float(np.mean([r[4].get("gpu", 0.0) for r in steady])), 3),
                       "write_ms": round(float(np.mean([r[4].get("write", 0.0) for r in steady])), 3)}
    else:
        s["steady"] = None                                 # too few tasks left after the slowest worker's first result
    return s


def quiet_digests(inputs, algo, ranges, out_dir):
    """The same calls in ONE quiet process (a fresh interpreter: the parent of the pools must stay cold)."""
    code = ("import sys, json; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import bench_pool as bp, os\n"
            "bp.apply_cfg(os.environ.get('S2P_POOL_CFG', ''))\n"
            "inputs = json.loads(sys.argv[1]); ranges = json.loads(sys.argv[4])\n"
            "out = [bp.task((1000000 + k, p1, p2, sys.argv[2], sys.argv[3], ranges[k][0], ranges[k][1], False, True))[5] for k, (p1, p2) in enumerate(inputs)]\n"
            "print('DIGESTS ' + json.dumps(out))\n") % (ROOT, os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code, json.dumps(inputs), out_dir, algo, json.dumps([list(x) for x in ranges])],
                       capture_output=True, text=True, timeout=900)
    if r.returncode != 0:
        raise RuntimeError("quiet run failed: " + r.stderr[-2000:])
    line = [l for l in r.stdout.splitlines() if l.startswith("DIGESTS ")][-1]
    return json.loads(line[8:])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workers", default="1,4,16,64", help="comma-separated Pool sizes, one fresh fork Pool each")
    ap.add_argument("--tiles", type=int, default=256, help="tasks per Pool (raised to 24 per worker so a steady state exists)")
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--ndisp", type=int, default=128)
    ap.add_argument("--algo", default="mgm", choices=["mgm", "mgm_multi", "sgbm"])
    ap.add_argument("--distinct", type=int, default=8, help="distinct seeded input pairs, cycled over the tasks")
    ap.add_argument("--ragged-depth", action="store_true", help="--ragged, and every pair its own range LENGTH as well (48 ... 223 candidates: "
                    "ranges come from the matches of a tile and are whatever they are)")
    ap.add_argument("--ragged", action="store_true", help="every distinct pair has its own size and disparity range (what a real job's tiles look "
                    "like): through the broker, requests of different shapes share a launch when their depths are close (s2p_hip_census_sgm_host_batch_v; "
                    "S2P_HIP_BROKER_HETERO=0: only equal shapes do)")
    ap.add_argument("--dir", default=None, help="where the TIFFs live (default: a fresh directory under /dev/shm)")
    ap.add_argument("--keep", action="store_true", help="keep every output file (default: a worker unlinks its outputs after the call, "
                    "as cfg['clean_intermediate'] does, so 1000 tiles do not need 9 GB of /dev/shm)")
    ap.add_argument("--broker", default="1", choices=["0", "1"], help="1 (default, what a Pool worker does by itself): the workers hand their tiles "
                    "to the device's GPU broker (s2p_amd/broker.py: one GPU-owning process, batched launches); 0: every worker drives the GPU itself")
    ap.add_argument("--lanes", type=int, default=None, help="broker: library contexts taking batches side by side (S2P_HIP_BROKER_LANES)")
    ap.add_argument("--procs", type=int, default=None, help="broker: GPU-owning broker processes per device (S2P_HIP_BROKER_PROCS; a worker talks to shard pid mod N)")
    ap.add_argument("--max-batch", type=int, default=None, help="broker: tiles per library call at most (S2P_HIP_BROKER_BATCH)")
    ap.add_argument("--max-wait-ms", type=float, default=None, help="broker: S2P_HIP_BROKER_WAIT_MS")
    ap.add_argument("--use-running-broker", action="store_true", help="do not restart the broker at the beginning (it runs under a profiler, say)")
    ap.add_argument("--keep-broker", action="store_true", help="leave the broker running at the end (default: it is asked to leave)")
    ap.add_argument("--trace", action="store_true", help="broker: latency accounting per request (S2P_HIP_BROKER_TRACE): worker -> broker, broker -> worker")
    ap.add_argument("--task-timeout", type=float, default=120.0, help="seconds r.get() waits for a task (the reference: 600)")
    ap.add_argument("--cfg", default="", help="cfg overrides for the matcher, key=value[,key=value...] (e.g. hip_mgm_multi_scales=6: the coarse-to-fine "
                    "mode of 'mgm_multi'); set in the parent before the fork, as a config.json is")
    ap.add_argument("--verify", action="store_true", help="hash every output in the worker (outside the timed call) and compare with a quiet "
                    "single-process run of the same inputs")
    a = ap.parse_args()
    workers = [int(x) for x in a.workers.split(",") if x]
    base = a.dir
    if base is None:
        shm = "/dev/shm"
        ok = os.path.isdir(shm) and shutil.disk_usage(shm).free > (2 << 30)
        base = tempfile.mkdtemp(prefix="s2p_pool_", dir=shm if ok else None)
    os.makedirs(base, exist_ok=True)
    os.environ["S2P_HIP_BROKER"] = a.broker              # inherited by the forked workers
    if a.trace:
        os.environ["S2P_HIP_BROKER_TRACE"] = "1"
    for k, v in (("S2P_HIP_BROKER_LANES", a.lanes), ("S2P_HIP_BROKER_BATCH", a.max_batch), ("S2P_HIP_BROKER_WAIT_MS", a.max_wait_ms), ("S2P_HIP_BROKER_PROCS", a.procs)):
        if v is not None:
            os.environ[k] = str(v)                       # ... and by the broker the first of them starts
    import s2p_amd                                       # noqa: F401  imported BEFORE the fork, as the orchestrator does
    from s2p_amd import _lib, broker
    apply_cfg(a.cfg)
    os.environ["S2P_POOL_CFG"] = a.cfg                   # (the quiet run of --verify is a fresh interpreter)
    if a.broker == "1" and not a.use_running_broker:
        broker.shutdown(0)                               # a broker left over from an earlier run: this run measures its own start
    _lib.lib()                                           # dlopen in the parent: no HIP call happens
    dmin, dmax = -a.ndisp // 2, a.ndisp // 2 - 1
    a.ragged = a.ragged or a.ragged_depth
    inputs = write_inputs(base, 0, a.size, a.ndisp, a.distinct, a.ragged)
    span = lambda k: 48 + (37 * k) % 176                 # --ragged-depth: 48 ... 223 candidates, all different for up to 176 pairs
    if a.ragged_depth:
        rng = lambda k: (-(span(k) // 2) - k % 7, -(span(k) // 2) - k % 7 + span(k) - 1)
    else:
        rng = lambda k: (dmin - (k if a.ragged else 0), dmax - (k if a.ragged else 0))     # ragged: every shape its own range, all of the same length
    res = {"workload": "fork Pool(P) x compute_disparity_map('%s') on %dx%d float32 TIFFs, %d disparities, files in %s; %d distinct pairs cycled; "
                       "outputs %s%s" % (a.algo, a.size, a.size, a.ndisp, base, a.distinct, "kept" if a.keep else "unlinked by the worker after each call",
                                        ("; RAGGED: every pair its own size (within 64 px of the nominal one) and its own range of its own length (48 ... 223 candidates)" if a.ragged_depth else
                                         "; RAGGED: every pair its own size (within 64 px of the nominal one) and its own range (shifted by its index, same length)") if a.ragged else ""),
           "reference_model": "s2p/parallel.py:76-110 (a fresh multiprocessing.Pool per step, fork start method), s2p/__init__.py:166-196",
           "mode": "GPU broker (one process owns the device; the workers read / write files and wait)" if a.broker == "1" else
                   "direct (every worker initialises HIP and launches its own kernels)",
           "pools": [], "errors": 0}
    all_digests = []
    totals = {}
    try:
        for P in workers:
            n = max(a.tiles, 24 * P)
            tasks = [(P * 100000 + i, inputs[i % len(inputs)][0], inputs[i % len(inputs)][1], base, a.algo) + rng(i % len(inputs)) + (a.keep, a.verify)
                     for i in range(n)]
            cg0 = cgroup_cpu()
            try:
                t_fork, t_end, out = run_pool(P, tasks, a.task_timeout)
            except Exception as e:                          # a HipError in a worker arrives here through r.get()
                res["errors"] += 1
                res["pools"].append({"workers": P, "error": repr(e)[:600]})
                continue
            cg1 = cgroup_cpu()
            res["pools"].append(summarise(P, t_fork, t_end, out))
            if cg0[1] is not None and cg1[1] is not None:   # how much CPU the whole Pool (workers + broker + this parent) used, against what the box grants
                wall = max(t_end - t_fork, 1e-9)
                res["pools"][-1]["cgroup_cpu"] = {"quota_cpus": cg1[0], "used_cpus": round((cg1[1] - cg0[1]) / 1e6 / wall, 2),
                                                  "throttled_s": round(((cg1[2] or 0) - (cg0[2] or 0)) / 1e6, 3), "throttle_events": (cg1[3] or 0) - (cg0[3] or 0),
                                                  "cpu_ms_per_tile": round((cg1[1] - cg0[1]) / 1e3 / max(1, len(out)), 3)}
            if a.broker == "1":
                try:                                        # this Pool's share of the broker's counters: how busy its lanes were
                    st = broker.stats(0, reset=True)
                    run = sum(v[1] for v in st.get("run_ms", {}).values())
                    res["pools"][-1]["broker"] = {"calls": st.get("calls"), "batch_hist": st.get("batch_hist"), "lanes": st.get("lanes"), "procs": len(st.get("shards", [1])),
                                                  "lane_busy_ms": round(run, 1), "queue_ms_per_request": round(st.get("queue_ms", 0.0) / max(1, st.get("requests", 1)), 3),
                                                  "lane_busy_frac_of_wall": round(run / (st.get("lanes", 1) * (t_end - t_fork) * 1e3), 3),
                                                  "arenas_new": int(st.get("attached", 0)) - int(st.get("recycled", 0)), "arenas_recycled": int(st.get("recycled", 0))}
                    cpu_now = float(st.get("cpu_s", 0.0))          # (summed over the shards)
                    if st.get("requests"):
                        res["pools"][-1]["broker"]["cpu_ms_per_request"] = round((cpu_now - totals.get("_cpu_s", 0.0)) * 1e3 / max(1, int(st.get("requests", 1))), 3)
                    totals["_cpu_s"] = cpu_now
                    tr = st.get("trace")
                    if tr and tr.get("n"):
                        res["pools"][-1]["broker"]["trace_ms_per_request"] = {k: round(tr[k] / tr["n"], 3) for k in ("ingress_ms", "egress_ms")}
                        res["pools"][-1]["broker"]["trace_ms_per_call"] = {"reply_ms": round(tr["reply_ms"] / max(1, st.get("calls", 1)), 3)}
                    for k in ("requests", "calls", "errors", "attached", "pinned", "recycled"):
                        totals[k] = totals.get(k, 0) + int(st.get(k, 0))
                except Exception as e:
                    res["pools"][-1]["broker"] = {"error": repr(e)[:200]}
                res["pools"][-1]["broker_start_inside_cold_start"] = P == workers[0] and not a.use_running_broker
                bs = [r[4].get("batch", 1) for r in out]
                res["pools"][-1]["mean_tiles_per_library_call"] = round(float(np.mean(bs)), 2)
            if a.verify:
                all_digests += [((r[0] % 100000) % len(inputs), r[5]) for r in out]
        if a.verify:
            os.environ["S2P_HIP_BROKER"] = "0"            # the quiet run drives the GPU itself, in its own process
            want = quiet_digests(inputs, a.algo, [rng(k) for k in range(len(inputs))], base)
            bad = sum(1 for k, dg in all_digests if dg != want[k])
            res["verify"] = {"outputs_compared": len(all_digests), "different_from_quiet_run": bad}
        if a.broker == "1":
            totals.pop("_cpu_s", None)
            res["broker"] = totals
    finally:
        if a.broker == "1" and not a.keep_broker:
            c = broker._clients.get((os.getpid(), 0))
            if c is not None:
                c.close()
            broker.shutdown(0)
        if a.dir is None:
            shutil.rmtree(base, ignore_errors=True)
    best = max((p for p in res["pools"] if p.get("steady")), key=lambda p: p["steady"]["tiles_per_s"], default=None)
    if best:
        res["best"] = {"workers": best["workers"], "steady_tiles_per_s": best["steady"]["tiles_per_s"],
                       "fork_to_join_tiles_per_s": best["tiles_per_s_fork_to_join"],
                       "Mdisp_per_s": None if a.ragged else round(best["steady"]["tiles_per_s"] * a.size * a.size * a.ndisp / 1e6, 1)}
        if a.ragged:                                     # mean W x H x candidates of the distinct pairs (the tasks cycle over them evenly)
            def shape(i):
                return (a.size - 8 * (i % 8) - i // 8, a.size - 24 + 4 * (i % 8) + i // 8)           # write_inputs' sizes
            mean = sum(shape(i)[0] * shape(i)[1] * (rng(i)[1] - rng(i)[0] + 1) for i in range(a.distinct)) / float(a.distinct)
            res["best"]["Mdisp_per_s"] = round(best["steady"]["tiles_per_s"] * mean / 1e6, 1)
            res["best"]["mean_candidates_per_tile"] = round(mean)
    print(json.dumps(res), flush=True)
    return 0 if (res["errors"] == 0 and not (a.verify and res["verify"]["different_from_quiet_run"])) else 1


if __name__ == "__main__":
    sys.exit(main())
