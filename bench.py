#!/usr/bin/env python3
"""bench.py -- the hot path on N GPUs of one node, one process per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one BATCH of `--batch` rectified tiles (default 256) through the matcher hot path (cost volume ->
semi-global aggregation -> WTA/sub-pixel/L-R -> median -> disparity + rejection mask), inputs already
resident in HBM, outputs left in HBM, `--streams` tiles in flight.  Workload = BASELINE.json configs[1]:
1024x1024 rectified tiles, 128 disparities, census 5x5.  The aggregation is the one the drop-in runs --
MGM's recursion with three predecessors (`--recursion 2`, the mode that meets the parity bar against the
reference's stored `mgm` outputs; 8 tiles per library call, three calls in flight); the 8 independent path sets north_star names (`--recursion 0`) are
reported beside it as `preview_8path`.  Tiles are independent, so ranks share nothing on the data path
(weak scaling: the same batches per GPU); the only collective is the final gather of the per-rank disparity
tiles ("DSM mosaic gather"), outside the timed region.  The `job` object of the same line is BASELINE
configs[3] as a job -- a FIXED list of 400 seeded 1000x1000x256 tiles from host windows through the tile
scheduler, split over the ranks by a shared work queue (strong scaling), with the RCCL mosaic gather.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel:
the aggregation launch, timed with HIP events on the stream it runs on) and `cpu_baseline`
(the reference matcher -- oracle/_ref, built from /root/reference -- or, if it did not travel,
the CPU oracle port, on a bounded sample of the same workload on this box's host cores).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
# RCCL between the ranks of one node shares device memory through dmabuf IPC on this stack; the legacy mode fails with
# `hipIpcGetMemHandle: invalid argument` (the image exports this already: set here so that a bare environment works too)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=None, help="tile width = height (default 1024; 1000 for the config3/4/5 workloads)")
    ap.add_argument("--ndisp", type=int, default=None, help="disparities (default 128; 256 for config3/config4)")
    ap.add_argument("--workload", default="tile", choices=["tile", "config3", "config4", "config5", "pool"],
                    help="tile (default): BASELINE.json configs[1], one resident 1024x1024x128 tile per step.  config3: the tile shape of "
                         "configs[3] (1000x1000, 256 disparities: a 256 MB cost volume, beyond the Infinity Cache), resident, same step.  "
                         "config4: configs[3] as a job -- 400 seeded 1000x1000x256 tiles (20 x 20 of a 20000^2 pair) from host windows through "
                         "tiles.process_queue (rectify -> match -> mask -> D2H per tile, shared work queue over the ranks), then the RCCL mosaic "
                         "gather; --steps = tiles per rank (default 400 / gpus).  config5: configs[4] -- tri-stereo, 2 pairs x 100 tiles of "
                         "1000x1000x128, per-pair matcher then fusion.merge_n per tile; --steps = tiles per rank (default 100 / gpus).  pool: the drop-in as "
                         "the reference runs it -- bench_pool.py: ONE cold parent forks multiprocessing.Pool(64 x gpus) workers x "
                         "compute_disparity_map('mgm') on 1024^2 TIFFs in /dev/shm; the workers spread over the node's GPUs (pid mod gpus), one GPU "
                         "broker per device; rank 0 runs the Pool, the other ranks only hold their place in the launch")
    ap.add_argument("--in-flight", type=int, default=None, help="config4/config5: tiles in flight per GPU (worker threads = HIP streams); default 3, 5 for the two-pair tiles of config5")
    ap.add_argument("--job-batch", type=int, default=None, help="config4/config5: tiles a worker takes from the queue per library call "
                    "(s2p_hip_tile_host_batch: one batched matcher launch); default 4 for the MGM matcher ('mgm') from 256 disparities, 1 otherwise")
    ap.add_argument("--pool", type=int, default=8, help="config4/config5: distinct synthetic tiles generated per rank (seed = 1000 ty + tx) and cycled")
    ap.add_argument("--tile-algo", default="mgm", choices=["mgm", "mgm_multi", "sgbm"], help="config4/config5: matching_algorithm of the jobs")
    ap.add_argument("--algo", default="census", choices=["census", "sgbm"],
                    help="census: 8-path SGM on a census 5x5 cost (BASELINE.json configs[1], the mgm stand-in); "
                         "sgbm: the bit-exact OpenCV StereoSGBM path")
    ap.add_argument("--batch", type=int, default=256, help="tile workloads: tiles per step (a step = one batch of independent tiles; "
                    "256 x 0.8 ms keeps the GPU busy for ~0.2 s per step, long enough for an outside observer to see it)")
    ap.add_argument("--batch-launch", type=int, default=0,
                    help="census MGM modes: tiles per library call (s2p_hip_census_sgm_dev_batch: one aggregation launch for all of them; "
                         "default 8 -- 1 = one tile per call; 3 tile streams either way)")
    ap.add_argument("--distinct", type=int, default=8, help="tile workloads: distinct seeded input pairs resident in HBM, cycled over the tiles of a step")
    ap.add_argument("--no-conf", action="store_true", help="census: skip the confidence image (the file-level 'mgm' call always computes it: s2p/block_matching.py:165)")
    ap.add_argument("--no-pool", action="store_true", help="skip the `pool` object (bench_pool.py: the drop-in under the reference's fork-Pool model)")
    ap.add_argument("--no-job", action="store_true", help="skip the `job` object (the fixed 400-tile config4 job through the scheduler)")
    ap.add_argument("--job-tiles", type=int, default=400, help="tiles of the `job` object (BASELINE configs[3]: 20 x 20)")
    ap.add_argument("--cpu-tiles", type=int, default=None, help="tiles for the cpu_baseline sample (default: ~10-20 s)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--graphs", action="store_true",
                    help="replay a captured hipGraph per tile instead of launching the kernels one by one "
                         "(measured on MI355X / ROCm 7.2: no gain, 0.074 vs 0.067 ms on 256x256x64 tiles; off by default)")
    ap.add_argument("--recursion", type=int, default=2, choices=(0, 1, 2),
                    help="census only: 2 (default) = MGM's recursion with three predecessors per direction (TSGM=3 of the 'mgm' call site as "
                         "modelled): what compute_disparity_map('mgm') runs and the mode that meets the parity bar (one band-pipelined launch per "
                         "tile, 3 tiles in flight); 1 = two predecessors; 0 = 8 independent path sets (north_star's wording; a faster preview mode "
                         "below the parity bar, reported as `preview_8path` when the headline is an MGM mode)")
    ap.add_argument("--streams", type=int, default=0,
                    help="tiles in flight per GPU, one HIP stream (libs2p_hip context) each; steps are issued round-robin. "
                         "Default: 1 for census, 3 with --recursion 1 (every 8-path kernel is bandwidth-bound and one tile's 134 MB cost volume lives in the "
                         "256 MB Infinity Cache between its 8 re-reads -- a second tile in flight evicts it: 0.528 vs 0.56-0.58 ms), "
                         "3 for sgbm (its compute-bound cost kernel overlaps the other tiles' memory-bound ones: 1.12 / 0.95 / 0.90 / 0.94 ms "
                         "with 1 / 2 / 3 / 4 streams)")
    a = ap.parse_args()
    if a.size is None:
        a.size = 1024 if a.workload in ("tile", "pool") else 1000
    if a.ndisp is None:
        a.ndisp = 256 if a.workload in ("config3", "config4") else 128
    if a.batch_launch <= 0:
        a.batch_launch = 8 if (a.algo == "census" and a.recursion and a.workload in ("tile", "config3")) else 1
    if a.streams <= 0:
        # MGM modes: three contexts in flight whatever the tiles per call (round 4, with the confidence image in the WTA: 0.657-0.669 ms per
        # tile on three streams against 0.698-0.704 on two, same box; profiles/r04/streams_probe.txt) -- the broker's lanes and the job's
        # workers are three as well
        a.streams = (3 if a.recursion else 1) if a.algo == "census" else 3
    if a.workload == "config3" or a.size > 1536:
        a.batch = min(a.batch, 64)               # larger tiles: keep a step around 0.1-0.3 s
    return a


def make_tile(seed, size, ndisp):
    """blurred-noise pair, smooth sinusoidal disparity field, s2p convention."""
    from helpers import synth_pair
    amp = 0.3125 * ndisp                      # 40 px at D = 128
    return synth_pair(seed, size, size,
                      lambda x, y: amp * np.sin(2 * np.pi * x / (size / 2.)) * np.cos(2 * np.pi * y / (size / 2.)))


def cpu_quota(root="/sys/fs/cgroup"):
    """CPUs' worth of time the control group of this process may use (cgroup v2 cpu.max, v1 cfs quota), None when unlimited.  The GPU boxes of
    this pool show 256 hardware threads and grant 16 (cpu.max = 1600000 100000): every many-process figure of the line -- the all-core CPU
    baseline, the forked-Pool model -- runs inside that budget, whatever os.cpu_count() says."""
    try:
        q, per = open(os.path.join(root, "cpu.max")).read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except Exception:
        pass
    try:
        q = float(open(os.path.join(root, "cpu", "cpu.cfs_quota_us")).read())
        per = float(open(os.path.join(root, "cpu", "cpu.cfs_period_us")).read())
        return None if q <= 0 else q / per
    except Exception:
        return None


def cpu_baseline(im1, im2, dmin, dmax, ntiles, algo):
    """The reference's CPU path on this box's host cores.  The only matcher whose source is in the
    reference tree is `sgbm` (3rdparty/sgbm), built as oracle/_ref/libsgbm_ref.so; `mgm` cannot be
    timed (sources absent).  kind == "reference": that library; kind == "port": our C restatement
    (oracle/) when the reference build did not travel to this box.  Two figures: one thread (the
    contract's `value`), and `all_cores`: N one-tile single-thread processes side by side -- the
    reference's own parallel model, a multiprocessing.Pool of tile workers (s2p/parallel.py:58-110)."""
    from oracle import pyoracle as po
    if po.have_ref():
        fn, kind = po.ref_sgbm, "reference"
    else:
        po.set_alias_oob(0)
        fn, kind = po.oracle_sgbm, "port"
    h, w = im1.shape
    devnull = os.open(os.devnull, os.O_WRONLY)
    saved = os.dup(2)
    os.dup2(devnull, 2)                       # the reference's qauto prints to stderr
    try:
        t0 = time.perf_counter()
        n = 0
        while True:
            fn(im1, im2, dmin, dmax)
            n += 1
            el = time.perf_counter() - t0
            if (ntiles is not None and n >= ntiles) or (ntiles is None and (el > 8.0 or n >= 6)):
                break
    finally:
        os.dup2(saved, 2)
        os.close(devnull)
        os.close(saved)
    cand = float(w) * h * (dmax - dmin)
    out = {"value": round(n * cand / el / 1e6, 3), "unit": "Mdisp/s", "cores": 1, "kind": kind,
           "sample": "%d tile(s) of the same %dx%dx%d workload through the reference `sgbm` matcher "
                     "(the only matcher with source in the reference tree), single thread, %.1f s" % (n, w, h, dmax - dmin, el),
           "s_per_tile": round(el / n, 4)}
    if algo == "census":   # also time the CPU statement of the census matcher itself (1 tile per mode, a few seconds each)
        for rec, key in ((2, "census_mgm_port"), (0, "census_port")):
            t0 = time.perf_counter()
            po.oracle_census_sgm(im1, im2, dmin, dmax - 1, params=po.census_params(recursion=rec))
            e2 = time.perf_counter() - t0
            out[key] = {"value": round(cand / e2 / 1e6, 3), "unit": "Mdisp/s", "cores": 1, "kind": "port",
                        "sample": "1 tile, oracle/census_oracle.c (%s), %.1f s" % ("MGM recursion, three predecessors: the GPU headline's algorithm" if rec else "8 path sets", e2)}
    # ---- all cores: N worker processes, one thread each, every one matching the same tile over and over for ~8 s
    try:
        import subprocess
        import tempfile
        ncpu = os.cpu_count() or 1
        mem_kb = 0
        with open("/proc/meminfo") as f:
            for line in f:
                if line.startswith("MemAvailable"):
                    mem_kb = int(line.split()[1])
        # N = nproc, bounded by memory only: a worker holds the C and S volumes (int16 each) + buffers + the interpreter
        per_worker_gb = 2.0 * w * h * (dmax - dmin + 16) * 2 / 1e9 * 1.25 + 0.4
        nproc = int(max(1, min(ncpu, (mem_kb / 1e6 * 0.7) / per_worker_gb)))
        model = "unknown"
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
        code = ("import sys, time, os, numpy as np\n"
                "sys.path.insert(0, %r)\n"
                "from oracle import pyoracle as po\n"
                "z = np.load(sys.argv[1]); a, b = z['a'], z['b']; dmin, dmax, budget = int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])\n"
                "fn = po.ref_sgbm if po.have_ref() else po.oracle_sgbm\n"
                "if not po.have_ref(): po.set_alias_oob(0)\n"
                "if len(sys.argv) > 5 and sys.argv[5] == 'census_mgm':\n"
                "    pm = po.census_params(recursion=2)\n"
                "    fn = lambda a, b, lo, hi: po.oracle_census_sgm(a, b, lo, hi - 1, params=pm)\n"
                "os.dup2(os.open(os.devnull, os.O_WRONLY), 2)\n"
                "t0 = time.perf_counter(); n = 0\n"
                "while True:\n"
                "    fn(a, b, dmin, dmax); n += 1\n"
                "    if time.perf_counter() - t0 > budget: break\n"
                "print(n, time.perf_counter() - t0)\n") % ROOT
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "tile.npz")
            np.savez(path, a=im1, b=im2)
            env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")

            def side_by_side(n, budget, extra=()):
                t0 = time.perf_counter()
                procs = [subprocess.Popen([sys.executable, "-c", code, path, str(dmin), str(dmax), str(budget)] + list(extra), stdout=subprocess.PIPE,
                                          stderr=subprocess.DEVNULL, env=env, text=True) for _ in range(n)]
                res = [pr.communicate(timeout=300)[0].split() for pr in procs]
                ok = [r for r in res if len(r) == 2]
                return sum(int(r[0]) for r in ok), max(float(r[1]) for r in ok), time.perf_counter() - t0

            # N = nproc is what BASELINE.md section 4 names; on a 256-thread host that many 1.2 GB working sets thrash the memory system
            # (round 5: 573 M disparities/s with 256 processes against 1 024 with 64), so the quarter -- one process per pair of physical
            # cores -- is timed beside it and the line carries both
            quota = cpu_quota()
            nq = max(1, min(nproc, ncpu // 4))
            if quota is not None and quota < nq:             # a CPU quota below that: one process per granted CPU is the sensible second leg
                nq = max(1, int(quota + 0.5))
            tiles, longest, wall = side_by_side(nproc, 8.0)
            quarter = side_by_side(nq, 8.0) if nq < nproc else None
            port = side_by_side(nq, 6.0, ["census_mgm"]) if algo == "census" else None
        if port:
            out["all_cores_census_mgm_port"] = {"value": round(port[0] * cand / port[1] / 1e6, 3), "unit": "Mdisp/s", "cores": nq, "kind": "port", "tiles": port[0], "s": round(port[1], 2),
                                                "sample": "%d single-thread processes of oracle/census_oracle.c (census + MGM recursion with three predecessors: the GPU "
                                                          "headline's own algorithm), each matching the same tile repeatedly for ~6 s" % nq}
        out["all_cores"] = {"value": round(tiles * cand / longest / 1e6, 3), "unit": "Mdisp/s", "cores": nproc, "host_cores": ncpu,
                            "cpu_quota": quota, "cpu_quota_note": None if quota is None else
                            "the control group of this box grants %.1f CPUs' worth of time (cpu.max): N = %d processes share that budget" % (quota, nproc),
                            "cpu_model": model, "kind": kind, "tiles": tiles, "s": round(longest, 2),
                            "sample": "N = %d single-thread processes on the host's %d hardware threads (the reference's Pool-of-tile-workers model with "
                                      "max_processes = nproc; bounded by memory only: %.1f GB per worker, %.0f GB available), each matching the same tile "
                                      "repeatedly for ~8 s: %d tiles in %.1f s (%.1f s with process start-up)" % (nproc, ncpu, per_worker_gb, mem_kb / 1e6, tiles, longest, wall)}
        if quarter:
            out["all_cores"]["quarter"] = {"value": round(quarter[0] * cand / quarter[1] / 1e6, 3), "unit": "Mdisp/s", "cores": nq, "tiles": quarter[0], "s": round(quarter[1], 2),
                                           "sample": ("the same with N = %d processes (one per CPU the control group grants)" % nq) if (quota is not None and nq == max(1, int(quota + 0.5))) else
                                                     ("the same with N = %d processes (hardware threads / 4): fewer working sets than cores' caches and memory channels can feed" % nq)}
    except Exception as e:                                   # the one-thread figure above is the contract's value
        out["all_cores"] = {"error": repr(e)[:200]}
    return out


def run_job(a, world, rank, local, cdev, workload, per_rank, tile_algo, strong_total=None):
    """BASELINE.json configs[3] / configs[4] as jobs through the tile scheduler (s2p_amd/tiles.py): every tile goes from
    two host-side source windows through rectification, the matcher, the rejection mask (and, config5, the fusion of the
    two pairs' maps) and back to the host in ONE library call per pair, `--in-flight` tiles at a time per GPU; the ranks
    share one work queue.  A unit = one tile (config5: one tile of both pairs + merge_n).  PCIe and the rectification are
    inside the timed region.  strong_total: the job is a FIXED list of that many tiles whatever the number of ranks (strong
    scaling); otherwise per_rank x world tiles (weak).  Every rank calls this; returns the result dict on rank 0."""
    import torch
    import torch.distributed as dist
    from s2p_amd import _lib as L
    from s2p_amd import tiles as T
    from s2p_amd.block_matching import matcher_params
    size = 1000
    nd = 128 if workload == "config5" else 256
    if a.workload in ("config4", "config5"):
        size, nd = a.size, a.ndisp
    dmin, dmax = -nd // 2, nd // 2 - 1
    pairs = 2 if workload == "config5" else 1
    ntiles = strong_total if strong_total is not None else per_rank * world
    grid = 10 if workload == "config5" else 20
    # source windows: the rectified synthetic pair of SURVEY.md 8(d) (seed = 1000 ty + tx) with a 12-px frame, handed over as the
    # "original image" windows with a sub-pixel translation as rectifying homography, so the resampler does real interpolation work
    pad = 12
    Hs = np.array([[1.0, 0.0, -pad + 0.25], [0.0, 1.0, -pad + 0.5], [0.0, 0.0, 1.0]])
    pool = []
    for k in range(max(1, min(a.pool, ntiles))):
        ty, tx = divmod(k * 7 % (grid * grid), grid)
        pool.append([L.pinned_copy(v) for v in make_tile_views(1000 * ty + tx, size + 2 * pad, nd, 1 + pairs)])   # as a reader that decodes into page-locked memory
    jobs = []
    for i in range(ntiles):
        v = pool[i % len(pool)]
        jobs.append([T.TileJob(i, v[0], Hs, v[1 + p], Hs, size, size, dmin, dmax) for p in range(pairs)])
    kind, params = matcher_params(tile_algo)
    # measured (profiles/r03/job_batch_probe.txt): 1000^2 x 256 tiles 1.60 -> 1.43 ms with 4 per call and 3 calls in
    # flight (4 / 6 in flight: the same, profiles/r06/job_inflight_probe.txt); the two-pair 128-disparity tiles of configs[4] have more host
    # work per launch (two rectifications, two matcher calls, merge_n): 5 in flight and 2 per call run them in 2.19-2.22 ms against 2.55
    # with 3 and 1 (4 per call lose: 2.64; 8 in flight lose: 2.39)
    in_flight = max(1, a.in_flight if a.in_flight is not None else (5 if pairs == 2 else 3))
    batch = a.job_batch if a.job_batch is not None else (4 if (tile_algo in ("mgm", "mgm_multi") and nd >= 256) else 2 if pairs == 2 else 1)
    batch = max(1, min(batch, 64 // pairs))
    runner = T._hip_pipeline(tile_algo, local, in_flight)
    dec = 4                                                  # the mosaic keeps every 4th pixel (a DSM is coarser than the images)

    def run(job_list):
        return finish([runner(j) for j in job_list])

    def run_many(group):                                     # `batch` tiles (x pairs) through one library call
        res = runner.many([j for g in group for j in g.lst])
        return [finish(res[k * pairs:(k + 1) * pairs]) for k in range(len(group))]

    def finish(res):
        if pairs == 1:
            return res[0]["disp"][::dec, ::dec].copy()
        hs = [r["disp"] * np.float32(1.0 / (1 + p)) for p, r in enumerate(res)]       # "heights": view p sees (1 + p) x the parallax
        offs = [0.0, 0.0]
        return L.merge_n(hs, offs, "average_if_close", threshold=3.0, device=local)[::dec, ::dec].copy()

    class J:                                                 # what process_queue schedules: one tile (all its pairs)
        def __init__(self, i, lst):
            self.index, self.lst = i, lst
            self.w, self.h, self.disp_min, self.disp_max = lst[0].w, lst[0].h, lst[0].disp_min, lst[0].disp_max

    sched_jobs = [J(i, lst) for i, lst in enumerate(jobs)]

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    def run_one(j):
        return run(j.lst)
    run_one.many = run_many
    for w_ in range(max(1, min(a.warmup, 2)) * in_flight):   # every context allocates its workspace
        if batch > 1 and ntiles >= batch:
            run_many([sched_jobs[(w_ * batch + k) % ntiles] for k in range(batch)])
        else:
            run(jobs[w_ % ntiles])
    sync()
    wq = T.WorkQueue(ntiles, chunk=batch)
    t0 = time.perf_counter()
    mine = T.process_queue(sched_jobs, wq, in_flight=in_flight, runner=run_one, batch=batch)
    sync()
    el = time.perf_counter() - t0
    tt = torch.tensor([el], dtype=torch.float64, device=cdev)
    cnt = torch.tensor([float(len(mine))], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        allc = [torch.zeros_like(cnt) for _ in range(world)]
        dist.all_gather(allc, cnt)
        per = [int(c.item()) for c in allc]
    else:
        per = [len(mine)]
    el = float(tt.item())
    # the mosaic gather (outside the timed region): tiles laid out row-major on a grid, decimated
    ts = (size + dec - 1) // dec
    cols = int(np.ceil(np.sqrt(ntiles)))
    layout = [((i // cols) * ts, (i % cols) * ts, ts, ts) for i in range(ntiles)]
    mshape = (((ntiles + cols - 1) // cols) * ts, cols * ts)
    mbuf = np.full(mshape, 0.0, np.float32) if rank == 0 else None    # the job's mosaic buffer exists before the gather (np.full touches its pages; np.zeros would not)
    tg = time.perf_counter()
    mosaic = T.gather_mosaic(mine, layout, mshape, dst=0, device=cdev if world > 1 else "cpu", dynamic=True, out=mbuf)
    gather_ms = (time.perf_counter() - tg) * 1e3
    if rank != 0:
        return None
    # ---- the job's own dominant kernel: the aggregation launch of its call shape, timed with HIP events on the
    # library's streams in a short separate pass after the timed region (same contexts, same call shape, one call at a time)
    job_roof = None
    try:
        ctxs = list(T._pools.get(local, []))[:in_flight]
        lib = L.lib()
        for c in ctxs:
            L.check(lib.s2p_hip_timing_enable(c, 1))
            L.check(lib.s2p_hip_timing_reset(c))
        reps = 6
        for k in range(reps):
            if batch > 1 and ntiles >= batch:
                run_many([sched_jobs[(k * batch + i) % ntiles] for i in range(batch)])
            else:
                run(jobs[k % ntiles])
        agg_ms, agg_n = 0.0, 0
        for c in ctxs:
            ms, n = ctypes.c_double(), ctypes.c_int()
            L.check(lib.s2p_hip_timing_get(c, b"aggregate", ctypes.byref(ms), ctypes.byref(n)))
            agg_ms += ms.value
            agg_n += n.value
            L.check(lib.s2p_hip_timing_enable(c, 0))
        if agg_n:
            per_launch = batch if (kind == "census" and batch > 1) else 1
            avg = agg_ms / agg_n
            mgm_mode = kind == "census" and params.recursion >= 1
            if kind == "sgbm":
                g = L.sgbm_geometry(size, dmin, dmax + 1)
                cand_launch, bpc, width = float(size) * g["width1"] * g["D"], 24.0, "b128"
            else:
                cand_launch, bpc = float(size) * size * nd * per_launch, 16.0
                width = "b64" if (mgm_mode and nd <= 128) else "b128"
            alg = bpc * cand_launch
            ach = alg / (avg * 1e-3) / 1e9
            job_roof = {"bound": "hbm", "kernel": "k_mgm_bands" if mgm_mode else "k_aggregate", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None, "alg_bytes_per_launch": alg, "alg_bytes_per_candidate": bpc,
                        "avg_launch_ms": round(avg, 4), "launches_timed": agg_n, "tiles_per_launch": per_launch,
                        "how": "HIP events around the aggregation launch on the library's own streams, %d calls of the job's shape one at a time after the timed region" % reps}
            tr = pmc_traffic("job_%s_b%d" % (tile_algo, per_launch), size, nd, job_roof["kernel"], width)
            if tr:
                job_roof["traffic"] = tr["bytes"]
                job_roof["frac_alg"] = job_roof["frac"]
                job_roof["frac"] = round(min(alg, tr["bytes"]) / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                job_roof["traffic_source"] = tr["source"] + ": " + tr["calibration"]
                job_roof["traffic_same_round"] = tr["same_round"]
    except Exception as e:                                   # the job's figures stand without it
        job_roof = {"error": "%s: %s" % (e.__class__.__name__, e)}
    cand = float(size) * size * nd * pairs
    return {"value": round(cand * ntiles / el / 1e6, 1), "unit": "Mdisp/s", "seconds": round(el, 4), "ms_per_tile": round(el / ntiles * 1e3, 4),
            "roofline": job_roof,
            "tiles": ntiles, "tiles_per_s": round(ntiles / el, 2), "tiles_per_rank": per, "n_gpus": world,
            "scaling": "strong" if strong_total is not None else "weak",
            "mosaic_gather_ms": round(gather_ms, 2), "mosaic_backend": "rccl" if (world > 1 and str(cdev) != "cpu") else ("gloo" if world > 1 else "none"),
            "mosaic_shape": list(mosaic.shape), "mosaic_valid": round(float(np.isfinite(mosaic).mean()), 4),
            "dtype": "u8" if kind == "census" else "int16", "pairs": pairs, "tile": [size, size], "ndisp": nd, "in_flight": in_flight, "tiles_per_call": batch,
            "workload": "%s: %d tiles of %dx%d, %d disparities%s, matching_algorithm '%s', host windows -> rectify -> match -> mask -> host, %d per library call, "
                        "%d in flight per GPU, one work queue shared by the ranks; %d distinct synthetic tiles (seed = 1000 ty + tx) cycled"
                        % (workload, ntiles, size, size, nd, " x 2 pairs + fusion.merge_n" if pairs == 2 else "", tile_algo, batch, in_flight, len(pool))}


def scheduler_workload(a, world, rank, local, dev, cdev, backend):
    """--workload config4 | config5: the job alone as the bench line (weak scaling: --steps tiles per rank)."""
    import torch.distributed as dist
    total = 100 if a.workload == "config5" else 400
    per_rank = a.steps if a.steps is not None else max(1, total // world)
    j = run_job(a, world, rank, local, cdev, a.workload, per_rank, a.tile_algo)
    if rank == 0:
        res = {
            "metric": "Mdisparities/s (WxHxD/s), whole job through the tile scheduler", "value": j["value"], "unit": "Mdisp/s",
            "n_gpus": world, "steps": per_rank, "warmup": a.warmup, "ms_per_step": round(j["seconds"] / per_rank * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": j["dtype"], "data": "synthetic",
            "config": {"workload": j["workload"], "tile": j["tile"], "ndisp": j["ndisp"], "pairs": j["pairs"], "tiles": j["tiles"],
                       "parallelism": "tiles x%d GPUs (no data-path collective; one mosaic gather at the end)" % world},
            "tiles_per_s": j["tiles_per_s"], "tiles_per_rank": j["tiles_per_rank"], "in_flight": j["in_flight"], "tiles_per_call": j["tiles_per_call"],
            "mosaic_gather_ms": j["mosaic_gather_ms"], "mosaic_shape": j["mosaic_shape"], "mosaic_valid": j["mosaic_valid"],
            "roofline": j.get("roofline"), "cpu_baseline": None,
            "note": "job-level figure: PCIe transfers and the rectification are inside the timed region; `roofline` is the job's own dominant kernel "
                    "(HIP events, separate pass); `cpu_baseline` is that of the resident-tile workloads (default, config3)",
        }
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def pool_object(size, nd, gpus=1, legs=("broker", "ragged", "direct")):
    """The drop-in as the reference runs it: bench_pool.py in a process of its own (its parent must never have
    touched HIP: it forks the Pools) -- P forked workers x compute_disparity_map('mgm') on TIFFs in /dev/shm, through the GPU
    broker (what a Pool worker does by default; `ragged`: 64 tile shapes, sizes and ranges a few pixels apart, as a real job has them)
    and, for comparison, with every worker driving the GPU itself."""
    import subprocess
    out = {}
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "S2P_HIP_DEVICE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "LOCAL_WORLD_SIZE", "ROLE_RANK"):
        env.pop(k, None)
    if gpus > 1:
        env["HIP_VISIBLE_DEVICES"] = ",".join(str(d) for d in range(gpus))
    P = 64 * gpus
    all_legs = (("broker", ["--workers", "4,16,%d" % P if gpus == 1 else str(P), "--tiles", str(512 * gpus), "--broker", "1"]),
                ("ragged", ["--workers", str(P), "--tiles", str(1024 * gpus), "--broker", "1", "--ragged", "--distinct", "64"]),
                ("direct", ["--workers", str(6 * gpus), "--tiles", str(384 * gpus), "--broker", "0", "--task-timeout", "60"]))     # (6: the device fence admits 8 processes, and the caller of this function is one)
    for key, args in [l for l in all_legs if l[0] in legs]:
        try:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench_pool.py"), "--size", str(size), "--ndisp", str(nd)] + args,
                               capture_output=True, text=True, timeout=300, env=env)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            out[key] = json.loads(line[-1]) if line else {"error": (r.stderr or r.stdout)[-300:]}
        except Exception as e:
            out[key] = {"error": repr(e)[:300]}
    b = out.get("broker", {}).get("best")
    if b:
        out["tiles_per_s"] = b["steady_tiles_per_s"]
        out["Mdisp_per_s"] = b["Mdisp_per_s"]
        out["workers"] = b["workers"]
    out["what"] = ("fork Pool(P) x s2p_amd.block_matching.compute_disparity_map('mgm') on %dx%d float32 TIFFs in /dev/shm, %d disparities "
                   "(s2p/parallel.py:76-110); steady state = every worker busy; fork_to_join includes each worker's cold start" % (size, size, nd))
    return out


def make_tile_views(seed, size, ndisp, nviews):
    from helpers import tile_views
    return tile_views(seed, size, ndisp, nviews)


ROUND = "r06"            # the round whose profiles/ directory this tree's evidence lives in


def pmc_calibration():
    """counter bytes / known bytes of rocprofv3's FETCH_SIZE and WRITE_SIZE for the access widths of the matcher's kernels, measured on
    known byte counts (tools/probes/pmc_calib.hip, tools/pmc_calib.sh -> profiles/rNN/pmc_calibration.json).
    MI355X_MICROARCH.md calibrates FETCH_SIZE for 16-byte-per-lane loads only (it reports 1/2 of the bytes); k_mgm_bands loads its
    costs 8 bytes per lane (raw_buffer_load_b64) and stores its e-bytes 8 bytes per lane non-temporally.  Returns
    {"fetch": {"b64": f, "b128": f}, "write": {"b64": f, "b128": f}, "source": path}; the guide's figures, flagged, when no file is committed."""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_calibration.json")))
    if paths:
        try:
            with open(paths[-1]) as f:
                d = json.load(f)
            return {"fetch": {"b64": d["calib_read_b64_band"]["FETCH_SIZE_bytes_over_known"], "b128": d["calib_read_b128_stream"]["FETCH_SIZE_bytes_over_known"]},
                    "write": {"b64": d["calib_write_b64_band_nt"]["WRITE_SIZE_bytes_over_known"], "b128": d["calib_write_b128_stream"]["WRITE_SIZE_bytes_over_known"]},
                    "source": os.path.relpath(paths[-1], ROOT), "calibrated": True}
        except Exception:
            pass
    return {"fetch": {"b64": 0.5, "b128": 0.5}, "write": {"b64": 1.0, "b128": 1.0}, "calibrated": False,
            "source": "MI355X_MICROARCH.md (16 B per lane only; the 8 B per lane figure is ASSUMED equal)"}


def pmc_traffic(algo, size, nd, kernel="k_aggregate", width="b128"):
    """L2 <-> fabric bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this same command
    (profiles/rNN/<algo>_<size>x<size>x<nd>_pmc_fetch_write.json; FETCH_SIZE and WRITE_SIZE are collected in two separate --pmc
    runs, in KiB), each counter divided by its known-byte calibration factor for the kernel's access width (pmc_calibration():
    FETCH_SIZE reports exactly 1/2 of the bytes at 8 and at 16 bytes per lane alike, WRITE_SIZE 1.00).  The newest round's file wins;
    the answer says which round it is from.  None when no matching profile is committed."""
    import glob
    cal = pmc_calibration()
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "%s_%dx%dx%d_pmc_fetch_write.json" % (algo, size, size, nd)))):
        try:
            with open(path) as f:
                d = json.load(f)
            for name, v in d.items():
                if kernel in name and "FETCH_SIZE_KiB_avg" in v and "WRITE_SIZE_KiB_avg" in v:
                    best = {"bytes": (v["FETCH_SIZE_KiB_avg"] / cal["fetch"][width] + v["WRITE_SIZE_KiB_avg"] / cal["write"][width]) * 1024.0,
                            "source": os.path.relpath(path, ROOT), "same_round": os.sep + ROUND + os.sep in path,
                            "calibration": "FETCH_SIZE / %.4g + WRITE_SIZE / %.4g (%s-per-lane accesses; %s)" % (cal["fetch"][width], cal["write"][width],
                                                                                                            "8-byte" if width == "b64" else "16-byte", cal["source"]),
                            "calibrated": cal["calibrated"]}
        except Exception:
            pass
    return best


def service_rate():
    """What the memory system gives the band kernel's own read / write mix when nothing computes and nothing waits (tools/probes/pmc_calib.hip:
    calib_rw_b64_band -- per wave-step one 8 B-per-lane load and one non-temporal 8 B-per-lane store of a 128-byte pixel, 4 skewed rows per wave,
    as many waves as the band launch has) and a streaming copy of the same bytes, from THIS round's profiles/<ROUND>/pmc_calibration_timing.txt.
    A measured ceiling beside the 8 TB/s spec peak the contract prices against; None when the file is not committed."""
    path = os.path.join(ROOT, "profiles", ROUND, "pmc_calibration_timing.txt")
    try:
        out = {}
        for line in open(path):
            f = line.split()
            if len(f) >= 5 and f[0] in ("calib_rw_b64_band", "calib_copy_b128_stream", "calib_read_b64_band", "calib_write_b64_band_nt"):
                out[f[0]] = float(f[3]) * 1e3
        if "calib_rw_b64_band" not in out:
            return None
        return {"rw_band_GBs": round(out["calib_rw_b64_band"], 1), "copy_stream_GBs": round(out.get("calib_copy_b128_stream", 0.0), 1) or None,
                "read_band_GBs": round(out.get("calib_read_b64_band", 0.0), 1) or None, "write_band_GBs": round(out.get("calib_write_b64_band_nt", 0.0), 1) or None,
                "what": "read + written bytes per second of 1 GiB moved by a kernel that only loads and stores (no arithmetic, no neighbour to wait for), "
                        "issued by as many waves as the band launch has (2 048); more waves get more: profiles/%s/rw_mix_sweep.txt" % ROUND,
                "source": os.path.relpath(path, ROOT)}
    except Exception:
        return None


def inflight_union(size, nd, tiles_per_call):
    """The measured per-launch cost of k_mgm_bands with calls in flight, from the committed rocprofv3 kernel trace of THIS command in
    THIS round (tools/inflight_union.py writes profiles/<ROUND>/mgm_inflight_b<tiles per call>_<size>x<size>x<nd>.json): union of the busy
    intervals of all k_mgm_bands launches / number of launches.  None when no such file is committed -- a trace of another round or of
    another call shape is not quoted."""
    path = os.path.join(ROOT, "profiles", ROUND, "mgm_inflight_b%d_%dx%dx%d.json" % (tiles_per_call, size, size, nd))
    try:
        with open(path) as f:
            d = json.load(f)
        per = float(d["union_ms_per_launch"])
        alg = 16.0 * size * size * nd * tiles_per_call
        return {"measured_union_ms_per_launch": per, "measured_launches": d.get("launches"), "measured_tiles_per_launch": tiles_per_call,
                "measured_mean_duration_ms_in_flight": d.get("mean_duration_ms_in_flight"),
                "measured_achieved_GBs": round(alg / (per * 1e-3) / 1e9, 1),
                "measured_frac": round(alg / (per * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "measured_source": os.path.relpath(path, ROOT)}
    except Exception:
        return None


def main():
    a = parse()
    import torch
    import torch.distributed as dist

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher (the way the driver calls --gpus 1): become the launch the contract names --
        # one process per GPU under torch.distributed.run on this node -- instead of dying before RCCL sees N ranks
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stderr.write("bench.py: --gpus %d without WORLD_SIZE: re-launching as `%s`\n" % (a.gpus, " ".join(cmd[1:])))
        sys.stderr.flush()
        os.execv(sys.executable, cmd)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus, "WORLD_SIZE = %d but --gpus %d: launch with torch.distributed.run --nproc-per-node %d" % (world, a.gpus, a.gpus)
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    # test hooks (tests of the N > 1 control flow on a 1-GPU box): S2P_BENCH_DEVICE pins every rank to one device,
    # S2P_BENCH_BACKEND=gloo replaces RCCL (two ranks cannot share a GPU under RCCL); the driver sets neither
    backend = os.environ.get("S2P_BENCH_BACKEND", "nccl")
    if "S2P_BENCH_DEVICE" in os.environ:
        local = int(os.environ["S2P_BENCH_DEVICE"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cdev = dev if backend == "nccl" else torch.device("cpu")      # where collective payloads live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    if a.workload in ("config4", "config5"):
        return scheduler_workload(a, world, rank, local, dev, cdev, backend)
    if a.workload == "pool":
        res = None
        if rank == 0:
            size, nd = a.size or 1024, a.ndisp or 128
            torch.cuda.synchronize()
            p = pool_object(size, nd, gpus=world, legs=("broker", "ragged"))
            best = (p.get("broker") or {}).get("best") or {}
            pools = (p.get("broker") or {}).get("pools") or [{}]
            res = {"metric": "Mdisparities/s (WxHxD/s), forked Pool workers x compute_disparity_map(files) through the GPU broker(s), steady state",
                   "value": best.get("Mdisp_per_s"), "unit": "Mdisp/s", "n_gpus": world, "steps": pools[-1].get("tiles"), "warmup": 0,
                   "ms_per_step": (pools[-1].get("steady") or {}).get("ms_per_tile"), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                   "dtype": "u8", "data": "synthetic",
                   "config": {"workload": p["what"], "tile": [size, size], "ndisp": nd, "workers": 64 * world,
                              "parallelism": "one forked Pool of %d workers over %d GPU(s), one broker per device" % (64 * world, world)},
                   "tiles_per_s": best.get("steady_tiles_per_s"), "fork_to_join_tiles_per_s": best.get("fork_to_join_tiles_per_s"),
                   "pool": p, "roofline": None, "cpu_baseline": None,
                   "note": "a step = one tile (one file-level call); `roofline` / `cpu_baseline` are those of the default workload"}
        if world > 1:
            dist.barrier()
        if rank == 0:
            print(json.dumps(res), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return
    if a.steps is None:
        a.steps = 5                          # 5 batches of 256 tiles: ~1 s of timed region at the default workload
    from s2p_amd import _lib as L
    lib = L.lib()
    size, nd = a.size, a.ndisp
    dmin, dmax = -nd // 2, nd // 2
    im1, im2 = make_tile(1000 + rank, size, nd)
    batch = max(1, a.batch)

    # inputs/outputs resident in HBM (torch is only the allocator / stream / collective plumbing): `--distinct` seeded pairs
    # (seeds 1000 + rank + 100 k), cycled over the tiles of a step so that the slots of a batched call hold different images
    d_pairs = [(torch.from_numpy(im1).to(dev), torch.from_numpy(im2).to(dev))]
    for k in range(1, max(1, a.distinct)):
        x1, x2 = make_tile(1000 + rank + 100 * k, size, nd)
        d_pairs.append((torch.from_numpy(x1).to(dev), torch.from_numpy(x2).to(dev)))
    d_im1, d_im2 = d_pairs[0]
    torch.cuda.synchronize()

    def new_out():
        return (torch.empty((size, size), dtype=torch.float32, device=dev), torch.empty((size, size), dtype=torch.float32, device=dev),
                torch.empty((size, size), dtype=torch.uint8, device=dev))

    class Mode:
        """One matcher configuration on its own contexts (one context = one non-blocking HIP stream + its own workspace)."""

        def __init__(self, algo, recursion, nstreams):
            self.algo, self.recursion = algo, recursion
            self.conf = not a.no_conf                            # the `<disp>_confidence.tif` image (s2p/block_matching.py:165, read at s2p/__init__.py:263-265)
            self.ctxs, self.outs, self.issued = [], [], 0
            self.nb = max(1, a.batch_launch) if (algo == "census" and recursion >= 1) else 1      # tiles per library call
            for _ in range(max(1, nstreams)):
                p = ctypes.c_void_p()
                L.check(lib.s2p_hip_ctx_create(local, None, ctypes.byref(p)))
                if a.graphs:
                    L.check(lib.s2p_hip_ctx_use_graphs(p, 1))    # device buffers are reused every step: capture once, replay
                self.ctxs.append(p)
                self.outs.append([new_out() for _ in range(self.nb)])
            self.params = L.default_sgbm_params() if algo == "sgbm" else L.default_census_params(recursion=recursion)

        def tile(self, k=None):
            """Enqueue one library call (= self.nb tiles) on stream k (round-robin when None)."""
            if k is None:
                k = self.issued % len(self.ctxs)
            self.issued += 1
            first = (self.issued - 1) * self.nb                  # tiles cycle over the distinct resident pairs
            if self.nb > 1:
                n = self.nb
                P = ctypes.c_void_p * n
                prs = [d_pairs[(first + j) % len(d_pairs)] for j in range(n)]
                ins1, ins2 = P(*[q[0].data_ptr() for q in prs]), P(*[q[1].data_ptr() for q in prs])
                dd, mm = P(*[o[0].data_ptr() for o in self.outs[k]]), P(*[o[2].data_ptr() for o in self.outs[k]])
                cc = P(*[o[1].data_ptr() for o in self.outs[k]]) if self.conf else None
                L.check(lib.s2p_hip_census_sgm_dev_batch(self.ctxs[k], n, ins1, ins2, size, size, dmin, dmax - 1, ctypes.byref(self.params), dd, cc, mm))
                return
            o = self.outs[k][0]
            q = d_pairs[first % len(d_pairs)]
            if self.algo == "sgbm":
                L.check(lib.s2p_hip_sgbm_dev(self.ctxs[k], q[0].data_ptr(), q[1].data_ptr(), size, size, dmin, dmax,
                                             ctypes.byref(self.params), o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr()))
            else:   # [dmin, dmax-1] inclusive = exactly `nd` candidates; the confidence image is what compute_disparity_map('mgm') always computes
                L.check(lib.s2p_hip_census_sgm_dev(self.ctxs[k], q[0].data_ptr(), q[1].data_ptr(), size, size, dmin, dmax - 1,
                                                   ctypes.byref(self.params), o[0].data_ptr(), o[1].data_ptr() if self.conf else None, o[2].data_ptr()))

        def sync(self, n=None):
            for c in self.ctxs[:n]:
                L.check(lib.s2p_hip_ctx_sync(c))

        def stage_ms(self, ntiles):
            """Per-kernel timing with HIP events on stream 0 (separate, un-timed pass; one stream: kernels of a tile back to back)."""
            c = self.ctxs[0]
            L.check(lib.s2p_hip_timing_enable(c, 1))
            L.check(lib.s2p_hip_timing_reset(c))
            for _ in range(max(2, ntiles // self.nb)):
                self.tile(0)
            st = {}
            for name in ("quantize", "cost", "aggregate", "wta", "median", "speckle", "epilogue", "total"):
                ms, n = ctypes.c_double(), ctypes.c_int()
                L.check(lib.s2p_hip_timing_get(c, name.encode(), ctypes.byref(ms), ctypes.byref(n)))
                st[name] = ms.value / max(n.value, 1)
            L.check(lib.s2p_hip_timing_enable(c, 0))
            return st

        def time_tiles(self, ntiles, nstreams):
            ncalls = max(nstreams, ntiles // self.nb)
            for i in range(2 * nstreams):
                self.tile(i % nstreams)
            self.sync(nstreams)
            t = time.perf_counter()
            for i in range(ncalls):
                self.tile(i % nstreams)
            self.sync(nstreams)
            return (time.perf_counter() - t) / (ncalls * self.nb) * 1e3

        def destroy(self):
            for c in self.ctxs:
                lib.s2p_hip_ctx_destroy(c)

    head = Mode(a.algo, a.recursion if a.algo == "census" else 0, a.streams)
    nstreams_head = len(head.ctxs)

    def sync_all():
        head.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    batch = max(head.nb, batch // head.nb * head.nb)   # whole library calls per step
    for _ in range(max(a.warmup, 1)):             # every context allocates its workspace during warm-up
        for _ in range(max(len(head.ctxs), min(batch // head.nb, 8))):
            head.tile()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        for _ in range(batch // head.nb):
            head.tile()
    sync_all()
    el = time.perf_counter() - t0
    tt = torch.tensor([el], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    el = float(tt.item())
    ntl = a.steps * batch                          # tiles per rank inside the timed region
    stages = head.stage_ms(10)
    one_stream_ms = head.time_tiles(max(12, min(ntl, 120)), 1) if (rank == 0 and len(head.ctxs) > 1) else None

    # ---- the other aggregation mode of the census matcher on the same tile, a short separate pass (rank 0, N = 1)
    other = None
    if rank == 0 and world == 1 and a.algo == "census":
        om = Mode("census", 0 if a.recursion else 2, 1 if a.recursion else 3)
        ns = len(om.ctxs)
        nm = max(12, min(ntl, 120))
        o_ms = om.time_tiles(nm, ns)
        o_ms1 = om.time_tiles(nm, 1) if ns > 1 else o_ms
        o_st = om.stage_ms(5)
        other = {"mode": om, "ms": o_ms, "ms1": o_ms1, "stages": o_st, "streams": ns, "tiles": nm, "nb": om.nb}

    # ---- achievable-copy ceiling of this device in the same run: a 1 GiB device-to-device copy,
    # read + write bytes over the elapsed time of 10 copies (torch is plumbing here: allocator + copy engine kernel)
    copy_gbs = None
    if rank == 0:
        try:
            src = torch.empty(1 << 28, dtype=torch.float32, device=dev).fill_(1.0)
            dst = torch.empty_like(src)
            for _ in range(2):
                dst.copy_(src)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                dst.copy_(src)
            e1.record()
            torch.cuda.synchronize()
            copy_gbs = 2.0 * src.numel() * 4 * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9
            del src, dst
        except Exception:
            copy_gbs = None

    # ---- final mosaic gather over RCCL/xGMI (not timed: once per run in the pipeline)
    gather_ms = None
    if world > 1:
        payload = head.outs[0][0][0] if backend == "nccl" else head.outs[0][0][0].cpu()
        out = [torch.empty_like(payload) for _ in range(world)] if rank == 0 else None
        torch.cuda.synchronize()
        tg = time.perf_counter()
        dist.gather(payload, out, dst=0)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - tg) * 1e3

    # ---- BASELINE configs[3] as a job, on every rank: a fixed list of tiles split by the shared work queue (strong scaling)
    job = None
    if not a.no_job and a.workload == "tile":
        if other is not None:
            other["mode"].destroy()
        head.destroy()                             # the resident-tile contexts (streams + up to 10 GB of workspace each) are done
        torch.cuda.synchronize()
        jt = a.job_tiles
        if size < 1024 or nd < 128:                # reduced runs (tests): a small job of the same shape
            jt = min(jt, 8 * world)
        job = run_job(a, world, rank, local, cdev, "config4", None, a.tile_algo, strong_total=jt)

    def roofline_of(algo, recursion, st_ms, tiles_per_launch=1):
        """The dominant kernel's line: algorithmic bytes of ONE launch / its average duration (HIP events on its own stream), and the
        same with min(algorithmic, PMC-measured) bytes -- SURVEY.md 8d: no credit for traffic the kernel does not generate."""
        if algo == "sgbm":
            g = L.sgbm_geometry(size, dmin, dmax)
            cand_k = float(size) * g["width1"] * g["D"]          # candidates the kernels visit (crop-trick canvas)
            agg_bpc, pipe_bpc = 24.0, 36.0                       # int16 C, uint8 e: 8 reads of C + 8 writes of e; pipeline: + C write + WTA (2 + 8)
        else:
            cand_k = float(size) * size * nd
            agg_bpc, pipe_bpc = 16.0, 26.0                       # uint8 C, uint8 e: 8 x (1 + 1); pipeline: + C write 1 + WTA (1 + 8) (25)
        mgm_mode = algo != "sgbm" and recursion
        cand_k *= tiles_per_launch                              # a batch call aggregates all its tiles in ONE launch
        agg_bytes, agg_s = agg_bpc * cand_k, st_ms["aggregate"] * 1e-3
        achieved = agg_bytes / agg_s / 1e9 if agg_s > 0 else 0.0
        roof = {"bound": "hbm", "boundary": "L2 <-> fabric (HBM + Infinity Cache): what FETCH_SIZE / WRITE_SIZE count",
                "kernel": "k_mgm_bands" if mgm_mode else "k_aggregate", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                "alg_bytes_per_launch": agg_bytes, "alg_bytes_per_candidate": agg_bpc,
                "avg_launch_ms": round(st_ms["aggregate"], 4),
                "copy_ceiling_GBs": round(copy_gbs, 1) if copy_gbs else None}
        if mgm_mode:
            roof["limiter"] = ("one tile alone: its dependency chain, (W + H) lattice steps of ~0.33-0.4 us on a lone in-order wave + one hand-off per band.  "
                               "Several tiles in one launch: the memory system's service rate for the kernel's traffic -- 8 B read + 8 B written per candidate in "
                               "128-byte granules.  Timing builds of round 6 (profiles/r06/decompose_probe.txt): with its bytes but without its flow control the "
                               "8-tile launch takes 4.5-5.1 ms whatever its workers (256 / 512 / 768); coupled by the flow control it reaches 4.13 ms and a third "
                               "band per CU changes nothing; without its bytes the same instructions run in 3.59 ms at two bands per CU, 3.16 at three.  The device "
                               "streams reads at 6.3, writes at 4.1-4.3 TB/s (profiles/r01/hbm_probe.txt) and moves this kernel's own read/write mix, with nothing "
                               "computed and as many waves as the launch has (2 048), at `service_rate.rw_band_GBs`; the same mix issued by 8-16 K waves reaches "
                               "5.3-5.9 TB/s (profiles/r06/rw_mix_sweep.txt) -- memory-level parallelism the launch cannot field: a band is a latency chain, and a third "
                               "band per CU measured no gain.  ONE fat band per CU (12 waves, 48 rows; or 16 candidates per lane) runs the launch alone in 3.83-3.86 ms but the three-stream "
                               "pipeline 2-4 % slower (profiles/r06/band_shape_probe.txt): the pipeline as a whole (`pipeline_alg_GBs`) already moves its bytes at the device's "
                               "copy rate (`copy_ceiling_GBs`, `service_rate.copy_stream_GBs`).  SQ counters and instruction census of this round: profiles/r06/sq_counters_mgm.txt, "
                               "mgm_step_isa.txt (153 instructions per wave-step, 92 VALU; a wave issues 44 % of its resident cycles); CU partitioning: cumask_sweep.txt")
            sr = service_rate()
            if sr:
                roof["service_rate"] = sr
                if sr.get("rw_band_GBs"):
                    roof["service_rate"]["frac_of_rw_band"] = round(achieved / sr["rw_band_GBs"], 4)
        # HBM-side MODEL -- not a measurement: the PMC counters sit at the L2 <-> fabric boundary and
        # count Infinity-Cache hits; what HBM itself moves is somewhere between two bounds.  Lower bound: every re-read of C served on
        # die (1 read of C + the e-writes) -- only possible while ALL the cost volumes a launch re-reads fit the 256 MiB cache beside
        # the streaming e-traffic, i.e. a single tile of <= ~150 MB.  Upper bound: no hit at all = the algorithmic bytes.  A batched
        # launch re-reads n volumes at once (8 x 134 MB = 1 GB at the headline shape): they do NOT stay, so the model takes the upper bound
        # there and frac_hbm_model equals the min-rule fraction.
        c_bytes = cand_k * (2.0 if algo == "sgbm" else 1.0)                   # all the cost volumes of the launch
        l3_resident = c_bytes <= 0.6 * 256 * 2 ** 20
        hbm_bytes = (c_bytes if l3_resident else 8.0 * c_bytes) + 8.0 * cand_k
        roof["hbm_bytes_model"] = hbm_bytes
        roof["hbm_model"] = ("MODEL, not measured: the launch's cost volume(s) (%.0f MB) fit the 256 MiB Infinity Cache between their 8 reads: HBM proper "
                             "would see 1 read of C + the 8 e-volume writes (lower bound of the HBM traffic)" if l3_resident else
                             "MODEL, not measured: the launch's cost volume(s) (%.0f MB) do not fit the 256 MiB Infinity Cache: every read of C is "
                             "charged to HBM (upper bound = the algorithmic bytes)") % (c_bytes / 1e6)
        roof["frac_hbm_model"] = round(hbm_bytes / agg_s / 1e9 / HBM_PEAK_GBS, 4) if agg_s > 0 else None
        roof["tiles_per_launch"] = tiles_per_launch
        base = ("census_mgm3" if recursion == 2 else "census_mgm") if mgm_mode else algo
        # access width of the dominant kernel's loads and stores: the band kernel moves 8 bytes per lane up to 256 disparities (K = 4),
        # the 8-path kernel and the 16-per-lane layouts 16
        width = "b64" if (mgm_mode and nd <= 128) else "b128"
        tr = pmc_traffic(base + "_b%d" % tiles_per_launch, size, nd, "k_mgm_bands", width) if tiles_per_launch > 1 else None
        if tr is None:
            tr = pmc_traffic(base, size, nd, "k_mgm_bands" if mgm_mode else "k_aggregate", width)
            if tr and tiles_per_launch > 1:                  # no PMC pass of the batched launch committed: the one-tile launch's, times the tiles
                tr["bytes"] *= tiles_per_launch
                tr["source"] += " x %d tiles" % tiles_per_launch
        if tr:
            roof["traffic"] = tr["bytes"]
            roof["frac_alg"] = roof["frac"]
            roof["frac"] = round(min(agg_bytes, tr["bytes"]) / agg_s / 1e9 / HBM_PEAK_GBS, 4) if agg_s > 0 else None   # the min-rule figure IS the headline fraction
            roof["frac_min_alg_traffic"] = roof["frac"]
            roof["traffic_source"] = tr["source"] + ": " + tr["calibration"]
            roof["traffic_same_round"] = tr["same_round"]
            roof["traffic_calibrated"] = tr["calibrated"]
        return roof, cand_k / tiles_per_launch, pipe_bpc

    if rank == 0:
        cand_tile = float(size) * size * nd                      # W x H x D of the tile (metric unit)
        dtype = "int16" if a.algo == "sgbm" else "u8"
        mgm_mode = a.algo != "sgbm" and a.recursion
        what = ("sgbm matcher (BT cost on Sobel-prefiltered u8, 3x3 blocks), 8-path SGM" if a.algo == "sgbm" else
                "census 5x5 / Hamming cost (mgm stand-in), " + ("MGM recursion over 8 directions, %d predecessors each%s" % (a.recursion + 1, " (the drop-in's mode)" if a.recursion == 2 else "") + "" if a.recursion else "8-path SGM"))
        value = cand_tile * ntl * world / el / 1e6
        roof, cand_k, pipe_bpc = roofline_of(a.algo, a.recursion if a.algo == "census" else 0, stages, head.nb)
        ms_tile = el / ntl * 1e3
        if mgm_mode:
            # with tiles in flight the launches of different tiles overlap; what one launch "costs" then is the tile time minus the
            # un-overlapped other stages -- a DERIVED figure; the MEASURED one is the union of k_mgm_bands' busy intervals in a
            # rocprofv3 kernel trace of this command (tools/inflight_union.py), committed under profiles/ and quoted here when present
            roof["in_flight"] = {"streams": nstreams_head, "ms_per_tile": round(ms_tile, 4),
                                 "pipeline_alg_GBs": round(pipe_bpc * cand_k / (ms_tile * 1e-3) / 1e9, 1),
                                 "pipeline_frac": round(pipe_bpc * cand_k / (ms_tile * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            iu = inflight_union(size, nd, head.nb)
            if iu:
                roof["in_flight"].update(iu)
        res = {
            "metric": "Mdisparities/s (WxHxD/s) per tile", "value": round(value, 1), "unit": "Mdisp/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(el / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": "%s%dx%d rectified tiles, %d disparities, %s%s; a step = a batch of %d independent tiles resident in HBM (%d distinct seeded pairs cycled)"
                                   % ("config3 (tile shape of BASELINE configs[3]): " if a.workload == "config3" else "", size, size, nd, what,
                                      "" if a.algo == "sgbm" else (", confidence image included" if not a.no_conf else ", WITHOUT the confidence image"), batch, len(d_pairs)),
                       "tile": [size, size], "ndisp": nd, "algo": a.algo, "recursion": int(a.recursion) if mgm_mode else 0, "tiles_per_step": batch,
                       "tiles_per_call": head.nb, "confidence": bool(a.algo != "sgbm" and not a.no_conf), "distinct_pairs": len(d_pairs),
                       "parallelism": "tiles x%d GPUs (no data-path collective), %d tile streams per GPU, %d tile(s) per library call" % (world, nstreams_head, head.nb)},
            "ms_per_tile": round(ms_tile, 4),
            "tiles_per_s": round(ntl * world / el, 2),
            "Mpx_per_s": round(size * size * ntl * world / el / 1e6, 1),
            "stage_ms": {k: round(v, 4) for k, v in stages.items()},
            "pipeline_alg_GBs": round(pipe_bpc * cand_k / (ms_tile * 1e-3) / 1e9, 1),
            "roofline": roof,
        }
        if one_stream_ms is not None:
            res["ms_per_tile_1_stream"] = round(one_stream_ms, 4)
        if other is not None:
            oroof, ocand, opipe = roofline_of("census", 0 if a.recursion else 2, other["stages"], other["nb"])
            res["preview_8path" if a.recursion else "mgm_recursion"] = {
                "what": ("8 independent path sets per direction (north_star's wording): a faster preview mode, BELOW the parity bar (98.9 % of the "
                         "reference's stored mgm tile within 0.5 px; the MGM recursion: 99.5 %)") if a.recursion else
                        "MGM's two-predecessor recursion (what compute_disparity_map('mgm') runs)",
                "ms_per_tile": round(other["ms"], 4), "value": round(cand_tile / (other["ms"] * 1e-3) / 1e6, 1), "unit": "Mdisp/s",
                "tiles": other["tiles"], "streams": other["streams"], "ms_per_tile_1_stream": round(other["ms1"], 4),
                "stage_ms": {k: round(v, 4) for k, v in other["stages"].items()}, "roofline": oroof}
        if job is not None:
            res["job"] = job
        if gather_ms is not None:
            res["mosaic_gather_ms"] = round(gather_ms, 3)
        if not a.no_pool and world == 1 and a.workload == "tile" and a.algo == "census" and size >= 256:
            res["pool"] = pool_object(size, nd)
        if not a.no_cpu and world == 1:      # contract: the CPU baseline is timed on rank 0 at N = 1 only
            res["cpu_baseline"] = cpu_baseline(im1, im2, dmin, dmax, a.cpu_tiles, a.algo)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
