#!/usr/bin/env python3
"""tools/pinned_probe.py -- what page-locked host buffers buy the host-buffer entry points (GPU box): the config4 job tile
(1000 x 1000, 256 disparities, 'mgm') through tiles.process_queue with pageable / page-locked source windows and result
arrays, 1 and 3 tiles in flight; and the file-level 'mgm' call."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import tile_views  # noqa: E402
from s2p_amd import _lib as L, tiles as T  # noqa: E402

size, nd, pad, n = 1000, 256, 12, 60
Hs = np.array([[1.0, 0.0, -pad + 0.25], [0.0, 1.0, -pad + 0.5], [0.0, 0.0, 1.0]])
views = [tile_views(1000 * k + 1, size + 2 * pad, nd, 2) for k in range(4)]
for src_pinned in (False, True):
    pool = [[L.pinned_copy(v) if src_pinned else v for v in vs] for vs in views]
    jobs = [T.TileJob(i, pool[i % 4][0], Hs, pool[i % 4][1], Hs, size, size, -nd // 2, nd // 2 - 1) for i in range(n)]
    for out_pinned in (False, True):
        for fl in (1, 3):
            runner = T._hip_pipeline("mgm", 0, fl, pinned=out_pinned)
            T.process_queue(jobs[:2 * fl], T.WorkQueue(2 * fl), in_flight=fl, runner=lambda j: runner(j) and None)
            t = time.perf_counter()
            T.process_queue(jobs, T.WorkQueue(n), in_flight=fl, runner=lambda j: runner(j) and None)
            ms = (time.perf_counter() - t) / n * 1e3
            print("sources %-9s results %-9s in flight %d: %.3f ms per tile" % ("pinned" if src_pinned else "pageable", "pinned" if out_pinned else "pageable", fl, ms), flush=True)
