"""tools/default_tile_time.py -- one measured point at s2p's DEFAULT tile geometry: `tile_size` 800
(s2p/config.py) gives rectified tiles of about 900 x 820 pixels once the margins and the rectifying homography are in
(s2p/initialization.py:164-185); with ~128 disparities that lies between the headline (1024^2 x 128) and the configs[2] tiles
(697 x 619 x 192) that profiles/r05/config2_time.txt measured.  Resident tiles, 'mgm' parameters with the confidence image, through
the same entry the headline uses (s2p_hip_census_sgm_dev_batch), 1 and 8 tiles per call, 1 and 3 calls in flight."""
import ctypes
import sys
import time

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import numpy as np
import torch
from helpers import synth_pair
from s2p_amd import _lib as L

lib = L.lib()
dev = torch.device("cuda", 0)
W, H, ND = 900, 820, 128
dmin, dmax = -ND // 2, ND // 2
pairs = []
for k in range(8):
    a, b = synth_pair(2000 + k, H, W, lambda x, y: 40.0 * np.sin(2 * np.pi * x / 450.) * np.cos(2 * np.pi * y / 410.))
    pairs.append((torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)))
torch.cuda.synchronize()
params = L.default_census_params(recursion=2)


def run(nb, nstreams, ncalls):
    ctxs, outs = [], []
    for _ in range(nstreams):
        p = ctypes.c_void_p()
        L.check(lib.s2p_hip_ctx_create(0, None, ctypes.byref(p)))
        ctxs.append(p)
        outs.append([(torch.empty((H, W), dtype=torch.float32, device=dev), torch.empty((H, W), dtype=torch.float32, device=dev),
                      torch.empty((H, W), dtype=torch.uint8, device=dev)) for _ in range(nb)])
    P = ctypes.c_void_p * nb

    def call(i):
        k = i % nstreams
        prs = [pairs[(i * nb + j) % len(pairs)] for j in range(nb)]
        L.check(lib.s2p_hip_census_sgm_dev_batch(ctxs[k], nb, P(*[q[0].data_ptr() for q in prs]), P(*[q[1].data_ptr() for q in prs]), W, H, dmin, dmax - 1,
                                                 ctypes.byref(params), P(*[o[0].data_ptr() for o in outs[k]]), P(*[o[1].data_ptr() for o in outs[k]]),
                                                 P(*[o[2].data_ptr() for o in outs[k]])))
    for i in range(2 * nstreams):
        call(i)
    for c in ctxs:
        L.check(lib.s2p_hip_ctx_sync(c))
    t = time.perf_counter()
    for i in range(ncalls):
        call(i)
    for c in ctxs:
        L.check(lib.s2p_hip_ctx_sync(c))
    dt = (time.perf_counter() - t) / (ncalls * nb)
    for c in ctxs:
        lib.s2p_hip_ctx_destroy(c)
    return dt


print("s2p's default tile geometry (tile_size 800 -> ~%d x %d rectified), %d disparities, 'mgm' (three predecessors, confidence image), resident" % (W, H, ND))
for nb, ns in ((1, 1), (1, 3), (8, 1), (8, 3)):
    dt = run(nb, ns, 96 // nb * (2 if nb == 1 else 4))
    print("  %d tile(s) per call, %d call(s) in flight: %.3f ms per tile = %.1f G disparities/s = %.0f tiles/s"
          % (nb, ns, dt * 1e3, W * H * ND / dt / 1e9, 1.0 / dt))
