"""tools/host_batch_time.py [size ndisp] -- the library call a broker lane makes (s2p_hip_census_sgm_host_batch: 8 tiles from page-locked host
planes laid out as an arena's, results back into them) timed WITHOUT the broker: one thread alone, then 3 and 4 threads side by side on
contexts of their own (what the lanes do).  Separates what the call itself costs (transfers + kernels + the final wait) from what the
broker adds around it."""
import ctypes
import sys
import threading
import time

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import numpy as np
from helpers import synth_pair
from s2p_amd import _lib as L

size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
nd = int(sys.argv[2]) if len(sys.argv) > 2 else 64
NB = 8
lib = L.lib()
params = L.default_census_params(recursion=2)
npx = size * size
a4 = (npx * 4 + 4095) // 4096 * 4096
slot = 4 * a4 + (npx + 4095) // 4096 * 4096
a, b = synth_pair(7, size, size, lambda x, y: 0.3 * nd * np.sin(x / 40.) * np.cos(y / 37.))


def make_lane():
    ctx = ctypes.c_void_p()
    L.check(lib.s2p_hip_ctx_create(0, None, ctypes.byref(ctx)))
    arena = L.pinned_empty((NB * slot,), np.uint8)
    base = arena.ctypes.data
    for t in range(NB):
        arena[t * slot:t * slot + npx * 4].view(np.float32)[:] = a.ravel()
        arena[t * slot + a4:t * slot + a4 + npx * 4].view(np.float32)[:] = b.ravel()
    ad = lambda k: [base + t * slot + k * a4 for t in range(NB)]
    return ctx, arena, ad


def call(lane):
    ctx, arena, ad = lane
    L.census_sgm_host_batch(ctx, ad(0), ad(1), size, size, -nd // 2, nd // 2 - 1, params, ad(2), ad(3), ad(4), 30.0)


lanes = [make_lane() for _ in range(4)]
for ln in lanes:
    call(ln)
    call(ln)
for nthreads in (1, 2, 3, 4):
    ncalls = 40
    def work(ln):
        for _ in range(ncalls):
            call(ln)
    ths = [threading.Thread(target=work, args=(lanes[k],)) for k in range(nthreads)]
    t = time.perf_counter()
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    dt = time.perf_counter() - t
    print("%d x %d x %d, %d tiles per call, %d thread(s): %.3f ms per call in the library, %.0f tiles/s"
          % (size, size, nd, NB, nthreads, dt / ncalls * 1e3, nthreads * ncalls * NB / dt))
