"""tools/config2_time.py -- BASELINE configs[2] (the reference's input_pair tiled 512 x 512, mgm_multi, 192 disparities) through the
tile scheduler (host windows -> rectify -> match -> mask -> host): ms per tile with 1, 2 and 3 tiles in flight, for 'mgm_multi' as the
shim runs it (one scale), as the config names it ("3-scale": cfg['hip_mgm_multi_scales'] = 6) and for 'mgm'."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import warnings
from helpers import config2_tiles
from s2p_amd import tiles as T
from s2p_amd.config import cfg
warnings.simplefilter("ignore")
tl, g = config2_tiles()
jobs = [T.TileJob(i, g["img_01"], H1, g["img_02"], H2, w, h, -96, 95) for i, (x0, y0, fx0, fy0, w, h, H1, H2) in enumerate(tl)] * 8
for algo, over in (("mgm_multi", {}), ("mgm_multi", {"hip_mgm_multi_scales": 6}), ("mgm", {})):
    c = dict(cfg); c.update(over)
    for fl in (1, 2, 3):
        T.process_tiles(jobs[:4], algo=algo, in_flight=fl, config=c)
        t = time.perf_counter(); T.process_tiles(jobs, algo=algo, in_flight=fl, config=c); dt = time.perf_counter() - t
        cand = jobs[0].w * jobs[0].h * 192
        print("config2 tiles (%dx%d rectified, 192 disparities), '%s' %s, %d in flight: %.2f ms per tile = %.1f G disparities/s"
              % (jobs[0].w, jobs[0].h, algo, over or "", fl, dt / len(jobs) * 1e3, cand / (dt / len(jobs)) / 1e9))
