#!/usr/bin/env python3
"""bench.py -- the hot path on N GPUs of one node, one process per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one rectified tile through the matcher hot path (quantise -> cost volume -> 8-path
semi-global aggregation -> WTA/sub-pixel/L-R -> median -> speckle -> disparity + rejection mask),
inputs already resident in HBM, outputs left in HBM.  Workload = BASELINE.json configs[1]:
single 1024x1024 rectified tile, 128 disparities.  Tiles are independent, so ranks share nothing on
the data path (weak scaling: one tile stream per GPU); the only collective is the final gather of
the per-rank disparity tiles ("DSM mosaic gather"), outside the timed region.

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

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=None, help="tile width = height (default 1024; 1000 for the config3/4/5 workloads)")
    ap.add_argument("--ndisp", type=int, default=None, help="disparities (default 128; 256 for config3/config4)")
    ap.add_argument("--workload", default="tile", choices=["tile", "config3", "config4", "config5"],
                    help="tile (default): BASELINE.json configs[1], one resident 1024x1024x128 tile per step.  config3: the tile shape of "
                         "configs[3] (1000x1000, 256 disparities: a 256 MB cost volume, beyond the Infinity Cache), resident, same step.  "
                         "config4: configs[3] as a job -- 400 seeded 1000x1000x256 tiles (20 x 20 of a 20000^2 pair) from host windows through "
                         "tiles.process_queue (rectify -> match -> mask -> D2H per tile, shared work queue over the ranks), then the RCCL mosaic "
                         "gather; --steps = tiles per rank (default 400 / gpus).  config5: configs[4] -- tri-stereo, 2 pairs x 100 tiles of "
                         "1000x1000x128, per-pair matcher then fusion.merge_n per tile; --steps = tiles per rank (default 100 / gpus)")
    ap.add_argument("--in-flight", type=int, default=3, help="config4/config5: tiles in flight per GPU (worker threads = HIP streams)")
    ap.add_argument("--pool", type=int, default=8, help="config4/config5: distinct synthetic tiles generated per rank (seed = 1000 ty + tx) and cycled")
    ap.add_argument("--tile-algo", default="mgm", choices=["mgm", "mgm_multi", "sgbm"], help="config4/config5: matching_algorithm of the jobs")
    ap.add_argument("--algo", default="census", choices=["census", "sgbm"],
                    help="census: 8-path SGM on a census 5x5 cost (BASELINE.json configs[1], the mgm stand-in); "
                         "sgbm: the bit-exact OpenCV StereoSGBM path")
    ap.add_argument("--cpu-tiles", type=int, default=None, help="tiles for the cpu_baseline sample (default: ~10-20 s)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--graphs", action="store_true",
                    help="replay a captured hipGraph per tile instead of launching the kernels one by one "
                         "(measured on MI355X / ROCm 7.2: no gain, 0.074 vs 0.067 ms on 256x256x64 tiles; off by default)")
    ap.add_argument("--recursion", type=int, default=0, choices=(0, 1),
                    help="census only: 0 = 8 independent path sets (the north_star workload, default), 1 = MGM's two-predecessor "
                         "recursion (the `mgm` binaries' aggregation: one band-pipelined launch per tile, ~2.5 x the tile time; 3 tiles in flight by default)")
    ap.add_argument("--streams", type=int, default=0,
                    help="tiles in flight per GPU, one HIP stream (libs2p_hip context) each; steps are issued round-robin. "
                         "Default: 1 for census, 3 with --recursion 1 (every 8-path kernel is bandwidth-bound and one tile's 134 MB cost volume lives in the "
                         "256 MB Infinity Cache between its 8 re-reads -- a second tile in flight evicts it: 0.528 vs 0.56-0.58 ms), "
                         "3 for sgbm (its compute-bound cost kernel overlaps the other tiles' memory-bound ones: 1.12 / 0.95 / 0.90 / 0.94 ms "
                         "with 1 / 2 / 3 / 4 streams)")
    a = ap.parse_args()
    if a.size is None:
        a.size = 1024 if a.workload == "tile" else 1000
    if a.ndisp is None:
        a.ndisp = 256 if a.workload in ("config3", "config4") else 128
    if a.streams <= 0:
        a.streams = (3 if a.recursion else 1) if a.algo == "census" else 3
    return a


def make_tile(seed, size, ndisp):
    """blurred-noise pair, smooth sinusoidal disparity field, s2p convention."""
    from helpers import synth_pair
    amp = 0.3125 * ndisp                      # 40 px at D = 128
    return synth_pair(seed, size, size,
                      lambda x, y: amp * np.sin(2 * np.pi * x / (size / 2.)) * np.cos(2 * np.pi * y / (size / 2.)))


def cpu_baseline(im1, im2, dmin, dmax, ntiles, algo):
    """The reference's CPU path on this box's host cores.  The only matcher whose source is in the
    reference tree is `sgbm` (3rdparty/sgbm), built as oracle/_ref/libsgbm_ref.so; `mgm` cannot be
    timed (sources absent).  kind == "reference": that library; kind == "port": our C restatement
    (oracle/) when the reference build did not travel to this box."""
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
            if (ntiles is not None and n >= ntiles) or (ntiles is None and (el > 10.0 or n >= 8)):
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
    if algo == "census":   # also time the CPU statement of the census matcher itself (1 tile, a few seconds)
        t0 = time.perf_counter()
        po.oracle_census_sgm(im1, im2, dmin, dmax - 1)
        e2 = time.perf_counter() - t0
        out["census_port"] = {"value": round(cand / e2 / 1e6, 3), "unit": "Mdisp/s", "cores": 1, "kind": "port",
                              "sample": "1 tile, oracle/census_oracle.c, %.1f s" % e2}
    return out


def scheduler_workload(a, world, rank, local, dev, cdev, backend):
    """BASELINE.json configs[3] / configs[4] as jobs through the tile scheduler (s2p_amd/tiles.py): every tile goes from
    two host-side source windows through rectification, the matcher, the rejection mask (and, config5, the fusion of the
    two pairs' maps) and back to the host in ONE library call per pair, `--in-flight` tiles at a time per GPU; the ranks
    share one work queue.  A "step" = one tile (config5: one tile of both pairs + merge_n).  Reported: whole-job tiles/s and
    W x H x D per second, host windows in / disparity + mask out (PCIe inside the timed region: this is the job-level
    figure, the resident-tile figure is the default workload)."""
    import torch
    import torch.distributed as dist
    from s2p_amd import _lib as L
    from s2p_amd import tiles as T
    from s2p_amd.block_matching import matcher_params
    size, nd = a.size, a.ndisp
    dmin, dmax = -nd // 2, nd // 2 - 1
    pairs = 2 if a.workload == "config5" else 1
    total = 100 if a.workload == "config5" else 400
    per_rank = a.steps if a.steps is not None else max(1, total // world)
    ntiles = per_rank * world
    grid = 10 if a.workload == "config5" else 20
    # source windows: the rectified synthetic pair of SURVEY.md 8(d) (seed = 1000 ty + tx) with a 12-px frame, handed over as the
    # "original image" windows with a sub-pixel translation as rectifying homography, so the resampler does real interpolation work
    pad = 12
    Hs = np.array([[1.0, 0.0, -pad + 0.25], [0.0, 1.0, -pad + 0.5], [0.0, 0.0, 1.0]])
    pool = []
    for k in range(max(1, min(a.pool, ntiles))):
        ty, tx = divmod(k * 7 % (grid * grid), grid)
        views = [make_tile_views(1000 * ty + tx, size + 2 * pad, nd, 1 + pairs)]
        pool.append(views[0])
    jobs = []
    for i in range(ntiles):
        v = pool[i % len(pool)]
        jobs.append([T.TileJob(i, v[0], Hs, v[1 + p], Hs, size, size, dmin, dmax) for p in range(pairs)])
    kind, params = matcher_params(a.tile_algo)
    in_flight = max(1, a.in_flight)
    runner = T._hip_pipeline(a.tile_algo, local, in_flight)
    dec = 4                                                  # the mosaic keeps every 4th pixel (a DSM is coarser than the images)

    def run(job_list):
        res = [runner(j) for j in job_list]
        if pairs == 1:
            return res[0]["disp"][::dec, ::dec].copy()
        hs = [r["disp"] * np.float32(1.0 / (1 + p)) for p, r in enumerate(res)]       # "heights": view p sees (1 + p) x the parallax
        offs = [0.0, 0.0]
        return L.merge_n(hs, offs, "average_if_close", threshold=3.0, device=local)[::dec, ::dec].copy()

    class J:                                                 # what process_queue schedules: one tile (all its pairs)
        def __init__(self, i, lst):
            self.index, self.lst = i, lst

    sched_jobs = [J(i, lst) for i, lst in enumerate(jobs)]

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for w_ in range(max(1, min(a.warmup, 2)) * in_flight):   # every context allocates its workspace
        run(jobs[w_ % ntiles])
    sync()
    wq = T.WorkQueue(ntiles)
    t0 = time.perf_counter()
    mine = T.process_queue(sched_jobs, wq, in_flight=in_flight, runner=lambda j: run(j.lst))
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
    tg = time.perf_counter()
    mosaic = T.gather_mosaic(mine, layout, (((ntiles + cols - 1) // cols) * ts, cols * ts), dst=0, device=cdev if world > 1 else "cpu", dynamic=True)
    gather_ms = (time.perf_counter() - tg) * 1e3
    if rank == 0:
        cand = float(size) * size * nd * pairs
        res = {
            "metric": "Mdisparities/s (WxHxD/s), whole job through the tile scheduler", "value": round(cand * ntiles / el / 1e6, 1), "unit": "Mdisp/s",
            "n_gpus": world, "steps": per_rank, "warmup": a.warmup, "ms_per_step": round(el / per_rank * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8" if kind == "census" else "int16", "data": "synthetic",
            "config": {"workload": "%s: %d tiles of %dx%d, %d disparities%s, matching_algorithm '%s', host windows -> rectify -> match -> mask -> host, "
                                   "%d in flight per GPU, shared work queue; %d distinct synthetic tiles cycled"
                                   % (a.workload, ntiles, size, size, nd, " x 2 pairs + fusion.merge_n" if pairs == 2 else "", a.tile_algo, in_flight, len(pool)),
                       "tile": [size, size], "ndisp": nd, "pairs": pairs, "tiles": ntiles,
                       "parallelism": "tiles x%d GPUs (no data-path collective; one mosaic gather at the end)" % world},
            "tiles_per_s": round(ntiles / el, 2), "tiles_per_rank": per,
            "mosaic_gather_ms": round(gather_ms, 2), "mosaic_shape": list(mosaic.shape), "mosaic_valid": round(float(np.isfinite(mosaic).mean()), 4),
            "roofline": None, "cpu_baseline": None,
            "note": "job-level figure: PCIe transfers and the rectification are inside the timed region; `roofline` / `cpu_baseline` are those "
                    "of the resident-tile workloads (default, config3)",
        }
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def make_tile_views(seed, size, ndisp, nviews):
    """`nviews` views of one synthetic scene: view 0 is the reference, view k sees k x the parallax."""
    from helpers import synth_pair
    amp = 0.3125 * ndisp / max(1, nviews - 1)
    f = lambda x, y: amp * np.sin(2 * np.pi * x / (size / 2.)) * np.cos(2 * np.pi * y / (size / 2.))
    im0, im1 = synth_pair(seed, size, size, f)
    out = [im0, im1]
    for k in range(2, nviews):
        out.append(synth_pair(seed, size, size, lambda x, y, k=k: k * f(x, y))[1])
    return out


def pmc_traffic(algo, size, nd, kernel="k_aggregate"):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this
    same command (profiles/rNN/<algo>_<size>x<size>x<nd>_pmc_fetch_write.json; FETCH_SIZE and
    WRITE_SIZE are collected in two separate --pmc runs, in KiB).  gfx950 correction
    (MI355X_MICROARCH.md, HBM section): FETCH_SIZE tallies 128-B requests at 64 B for wide
    coalesced streaming reads -> doubled; WRITE_SIZE is exact for our stores (it equals the output
    volume to the byte).  None when no matching profile is committed."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "%s_%dx%dx%d_pmc_fetch_write.json" % (algo, size, size, nd)))):
        try:
            with open(path) as f:
                d = json.load(f)
            for name, v in d.items():
                if kernel in name and "FETCH_SIZE_KiB_avg" in v and "WRITE_SIZE_KiB_avg" in v:
                    best = {"bytes": (2.0 * v["FETCH_SIZE_KiB_avg"] + v["WRITE_SIZE_KiB_avg"]) * 1024.0,
                            "source": os.path.relpath(path, ROOT)}
        except Exception:
            pass
    return best


def main():
    a = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus, "launch with torch.distributed.run --nproc-per-node %d" % a.gpus
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
    if a.steps is None:
        a.steps = 1000                       # ~0.5 s of timed region at the default workload: long enough for an outside observer to see the GPU busy
    from s2p_amd import _lib as L
    lib = L.lib()
    size, nd = a.size, a.ndisp
    dmin, dmax = -nd // 2, nd // 2
    im1, im2 = make_tile(1000 + rank, size, nd)

    # inputs/outputs resident in HBM (torch is only the allocator / stream / collective plumbing)
    d_im1 = torch.from_numpy(im1).to(dev)
    d_im2 = torch.from_numpy(im2).to(dev)
    d_disp = torch.empty((size, size), dtype=torch.float32, device=dev)
    d_cost = torch.empty((size, size), dtype=torch.float32, device=dev)
    d_mask = torch.empty((size, size), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    # one context = one non-blocking HIP stream + its own workspace; `--streams` tiles in flight per GPU
    ctxs = []
    for _ in range(max(1, a.streams)):
        p = ctypes.c_void_p()
        L.check(lib.s2p_hip_ctx_create(local, None, ctypes.byref(p)))
        if a.graphs:
            L.check(lib.s2p_hip_ctx_use_graphs(p, 1))    # device buffers are reused every step: capture once, replay
        ctxs.append(p)
    ctx = ctxs[0]
    outs = [(d_disp, d_cost, d_mask)] + [(torch.empty_like(d_disp), torch.empty_like(d_cost), torch.empty_like(d_mask))
                                         for _ in ctxs[1:]]
    issued = [0]
    if a.algo == "sgbm":
        params = L.default_sgbm_params()

        def step(c=None):
            k = issued[0] % len(ctxs) if c is None else 0
            issued[0] += 1
            o = outs[k]
            L.check(lib.s2p_hip_sgbm_dev(ctxs[k], d_im1.data_ptr(), d_im2.data_ptr(), size, size, dmin, dmax,
                                         ctypes.byref(params), o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr()))
    else:
        params = L.default_census_params(recursion=a.recursion)   # the 'mgm' call of s2p: census 5x5, P1 8, P2 32, 8 dirs, vfit, LR, median

        def step(c=None):   # [dmin, dmax-1] inclusive = exactly `nd` candidates; no confidence image (optional output)
            k = issued[0] % len(ctxs) if c is None else 0
            issued[0] += 1
            o = outs[k]
            L.check(lib.s2p_hip_census_sgm_dev(ctxs[k], d_im1.data_ptr(), d_im2.data_ptr(), size, size, dmin, dmax - 1,
                                               ctypes.byref(params), o[0].data_ptr(), None, o[2].data_ptr()))

    def sync_all():
        for c in ctxs:
            L.check(lib.s2p_hip_ctx_sync(c))
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(max(a.warmup, len(ctxs))):     # every context allocates its workspace during warm-up
        step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    sync_all()
    el = time.perf_counter() - t0
    tt = torch.tensor([el], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    el = float(tt.item())

    # ---- per-kernel timing with HIP events on the ctx stream (separate, un-timed pass: the events
    # themselves add launches between kernels)
    stages = {}
    L.check(lib.s2p_hip_timing_enable(ctx, 1))
    L.check(lib.s2p_hip_timing_reset(ctx))
    nt = max(3, min(a.steps, 10))
    for _ in range(nt):
        step(0)                                   # single stream: kernels of one tile back to back
    for s in ("quantize", "cost", "aggregate", "wta", "median", "speckle", "epilogue", "total"):
        ms, n = ctypes.c_double(), ctypes.c_int()
        L.check(lib.s2p_hip_timing_get(ctx, s.encode(), ctypes.byref(ms), ctypes.byref(n)))
        stages[s] = ms.value / max(n.value, 1)
    L.check(lib.s2p_hip_timing_enable(ctx, 0))

    # ---- the same tile in the MGM two-predecessor mode (the aggregation of the reference's `mgm` binary; what the
    # file-level 'mgm' shim runs): a short separate pass, reported next to the headline as `mgm_recursion`
    mgm = None
    if rank == 0 and world == 1 and a.algo == "census" and not a.recursion:
        pm = L.default_census_params(recursion=1)
        mctx = list(ctxs)
        while len(mctx) < 3:                                  # MGM mode: three tiles in flight (its launch is a dependency chain)
            p_ = ctypes.c_void_p()
            L.check(lib.s2p_hip_ctx_create(local, None, ctypes.byref(p_)))
            mctx.append(p_)
        mouts = list(outs) + [(torch.empty_like(d_disp), torch.empty_like(d_cost), torch.empty_like(d_mask)) for _ in range(len(mctx) - len(outs))]

        def mgm_step(k):
            o = mouts[k]
            L.check(lib.s2p_hip_census_sgm_dev(mctx[k], d_im1.data_ptr(), d_im2.data_ptr(), size, size, dmin, dmax - 1,
                                               ctypes.byref(pm), o[0].data_ptr(), None, o[2].data_ptr()))
        nm = max(6, min(a.steps, 120))
        res_ms = {}
        for ns in (1, 3):
            for i in range(2 * ns):
                mgm_step(i % ns)
            for c in mctx[:ns]:
                L.check(lib.s2p_hip_ctx_sync(c))
            tm = time.perf_counter()
            for i in range(nm):
                mgm_step(i % ns)
            for c in mctx[:ns]:
                L.check(lib.s2p_hip_ctx_sync(c))
            res_ms[ns] = (time.perf_counter() - tm) / nm * 1e3
        L.check(lib.s2p_hip_timing_enable(mctx[0], 1))
        L.check(lib.s2p_hip_timing_reset(mctx[0]))
        for _ in range(5):
            mgm_step(0)
        ms_, n_ = ctypes.c_double(), ctypes.c_int()
        L.check(lib.s2p_hip_timing_get(mctx[0], b"aggregate", ctypes.byref(ms_), ctypes.byref(n_)))
        L.check(lib.s2p_hip_timing_enable(mctx[0], 0))
        agg_ms = ms_.value / max(n_.value, 1)
        cand_ = float(size) * size * nd
        mgm = {"ms_per_step": round(res_ms[3], 4), "value": round(cand_ / (res_ms[3] * 1e-3) / 1e6, 1), "unit": "Mdisp/s",
               "steps": nm, "streams": 3, "ms_per_step_1_stream": round(res_ms[1], 4),
               "kernel": "k_mgm_bands (one launch per tile)",
               "roofline": {"bound": "dependency chain (W + H steps of ~0.33-0.4 us + one hand-off per band of 32 rows), not HBM", "kernel": "k_mgm_bands",
                            "avg_launch_ms": round(agg_ms, 4), "alg_bytes_per_candidate": 16.0,
                            "achieved": round(16.0 * cand_ / (agg_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(16.0 * cand_ / (agg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                            "achieved_3_in_flight": round(16.0 * cand_ / (max(res_ms[3] - (res_ms[1] - agg_ms), 1e-3) * 1e-3) / 1e9, 1)}}
        for c in mctx[len(ctxs):]:
            lib.s2p_hip_ctx_destroy(c)

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
        payload = d_disp if backend == "nccl" else d_disp.cpu()
        out = [torch.empty_like(payload) for _ in range(world)] if rank == 0 else None
        torch.cuda.synchronize()
        tg = time.perf_counter()
        dist.gather(payload, out, dst=0)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - tg) * 1e3

    if rank == 0:
        cand_tile = float(size) * size * nd                      # W x H x D of the tile (metric unit)
        if a.algo == "sgbm":
            g = L.sgbm_geometry(size, dmin, dmax)
            cand_k = float(size) * g["width1"] * g["D"]          # candidates the kernels visit (crop-trick canvas)
            # int16 C, uint8 e = C - L:  aggregation = 8 reads of C + 8 writes of e = 24 B / candidate;
            # whole pipeline = C write 2 + 24 + WTA (2 + 8) = 36 B / candidate
            agg_bpc, pipe_bpc, dtype = 24.0, 36.0, "int16"
            what = "sgbm matcher (BT cost on Sobel-prefiltered u8, 3x3 blocks)"
        else:
            cand_k = cand_tile
            # uint8 C, uint8 e: aggregation = 8 x (1 + 1) = 16 B / candidate;
            # whole pipeline = C write 1 + 16 + WTA (1 + 8) = 26 B / candidate (25)
            agg_bpc, pipe_bpc, dtype = 16.0, 26.0, "u8"
            what = "census 5x5 / Hamming cost (mgm stand-in)" + (", MGM two-predecessor recursion" if a.recursion else "")
        value = cand_tile * a.steps * world / el / 1e6
        agg_bytes = agg_bpc * cand_k
        agg_s = stages["aggregate"] * 1e-3
        achieved = agg_bytes / agg_s / 1e9 if agg_s > 0 else 0.0
        roof = {"bound": "hbm", "kernel": "k_mgm_bands" if (a.algo != "sgbm" and a.recursion) else "k_aggregate", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                "alg_bytes_per_launch": agg_bytes, "alg_bytes_per_candidate": agg_bpc,
                "avg_launch_ms": round(stages["aggregate"], 4),
                "copy_ceiling_GBs": round(copy_gbs, 1) if copy_gbs else None}
        # HBM-side model: the PMC counters sit at the L2 <-> fabric boundary and count Infinity-Cache hits; what HBM
        # itself moves is the first read of C and the e-volume writes when C (re-read by the 8 directions) fits the 256 MiB cache, and
        # every read of C when it does not
        c_bytes = cand_k * (2.0 if a.algo == "sgbm" else 1.0)
        l3_resident = c_bytes <= 0.6 * 256 * 2 ** 20
        hbm_bytes = (c_bytes if l3_resident else 8.0 * c_bytes) + 8.0 * cand_k
        roof["hbm_bytes_model"] = hbm_bytes
        roof["hbm_model"] = ("C (%.0f MB) stays in the 256 MiB Infinity Cache between its 8 reads: HBM sees 1 read of C + the 8 e-volume writes"
                             if l3_resident else "C (%.0f MB) does not fit the 256 MiB Infinity Cache: HBM sees all 8 reads of C + the 8 e-volume writes") % (c_bytes / 1e6)
        roof["frac_hbm"] = round(hbm_bytes / agg_s / 1e9 / HBM_PEAK_GBS, 4) if agg_s > 0 else None
        mgm_mode = a.algo != "sgbm" and a.recursion
        tr = pmc_traffic("census_mgm" if mgm_mode else a.algo, size, nd, "k_mgm_bands" if mgm_mode else "k_aggregate")
        if tr:
            roof["traffic"] = tr["bytes"]
            # SURVEY.md 8d rule: no credit for traffic the kernel does not generate -> the fraction with min(algorithmic, measured)
            roof["frac_min_alg_traffic"] = round(min(agg_bytes, tr["bytes"]) / agg_s / 1e9 / HBM_PEAK_GBS, 4) if agg_s > 0 else None
            roof["traffic_source"] = tr["source"] + " (2 x FETCH_SIZE + WRITE_SIZE, KiB -> bytes)"
        pipe_bytes = pipe_bpc * cand_k
        res = {
            "metric": "Mdisparities/s (WxHxD/s) per tile", "value": round(value, 1), "unit": "Mdisp/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(el / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": "%ssingle %dx%d rectified tile, %d disparities, 8-path SGM, %s"
                                   % ("config3 (tile shape of BASELINE configs[3]): " if a.workload == "config3" else "", size, size, nd, what),
                       "tile": [size, size], "ndisp": nd, "algo": a.algo,
                       "parallelism": "tiles x%d GPUs (no data-path collective), %d tile streams per GPU" % (world, len(ctxs))},
            "tiles_per_s": round(a.steps * world / el, 2),
            "Mpx_per_s": round(size * size * a.steps * world / el / 1e6, 1),
            "stage_ms": {k: round(v, 4) for k, v in stages.items()},
            "pipeline_alg_GBs": round(pipe_bytes / (el / a.steps) / 1e9, 1),
            "roofline": roof,
        }
        if mgm is not None:
            res["mgm_recursion"] = mgm
        if gather_ms is not None:
            res["mosaic_gather_ms"] = round(gather_ms, 3)
        if not a.no_cpu and world == 1:      # contract: the CPU baseline is timed on rank 0 at N = 1 only
            res["cpu_baseline"] = cpu_baseline(im1, im2, dmin, dmax, a.cpu_tiles, a.algo)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
