"""tools/config2_time.py -- BASELINE configs[2] (the reference's input_pair tiled 512 x 512, mgm_multi, 192 disparities) through the
tile scheduler: ms per tile, 1 and 2 tiles in flight, for 'mgm_multi' and 'mgm'."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from helpers import config2_tiles
from s2p_amd import tiles as T
tl, g = config2_tiles()
jobs = [T.TileJob(i, g["img_01"], H1, g["img_02"], H2, w, h, -96, 95) for i, (x0, y0, fx0, fy0, w, h, H1, H2) in enumerate(tl)] * 4
for algo in ("mgm_multi", "mgm"):
    for fl in (1, 2):
        T.process_tiles(jobs[:4], algo=algo, in_flight=fl)
        t = time.perf_counter(); T.process_tiles(jobs, algo=algo, in_flight=fl); dt = time.perf_counter() - t
        print("config2 tiles (%dx%d, 192 disparities), '%s', %d in flight: %.2f ms per tile" % (jobs[0].w, jobs[0].h, algo, fl, dt / len(jobs) * 1e3))
