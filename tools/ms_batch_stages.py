"""tools/ms_batch_stages.py [size ndisp algo n,n,...] -- (GPU box) 'mgm_multi' tiles through s2p_hip_census_sgm_host_batch, n per call: wall time per
tile and the library's per-stage event times per tile (all levels summed), against one tile per call."""
import ctypes, sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from helpers import synth_pair
from s2p_amd import _lib as L, block_matching
size = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
nd = int(sys.argv[2]) if len(sys.argv) > 2 else 256
algo = sys.argv[3] if len(sys.argv) > 3 else "mgm_multi"
import warnings; warnings.simplefilter("ignore")
kind, p = block_matching.matcher_params(algo)
dmin, dmax = -nd // 2, nd // 2 - 1
lib = L.lib()
ns = [int(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else (1, 2, 4, 8)
for n in ns:
    tiles = [synth_pair(5 + t, size, size, lambda x, y: 0.3 * nd * np.sin(2 * np.pi * x / 512.) * np.cos(2 * np.pi * y / 512.)) for t in range(n)]
    pin = lambda a: L.pinned_copy(a)
    a = [pin(t[0]) for t in tiles]; b = [pin(t[1]) for t in tiles]
    d = [L.pinned_empty((size, size)) for _ in range(n)]; c = [L.pinned_empty((size, size)) for _ in range(n)]; m = [L.pinned_empty((size, size), np.uint8) for _ in range(n)]
    ctx = ctypes.c_void_p(); L.check(lib.s2p_hip_ctx_create(0, None, ctypes.byref(ctx)))
    ad = lambda xs: [x.ctypes.data for x in xs]
    call = lambda: L.census_sgm_host_batch(ctx, ad(a), ad(b), size, size, dmin, dmax, p, ad(d), ad(c), ad(m))
    for _ in range(3): call()
    lib.s2p_hip_timing_enable(ctx, 1); lib.s2p_hip_timing_reset(ctx)
    reps = 6
    t0 = time.perf_counter()
    for _ in range(reps): call()
    dt = (time.perf_counter() - t0) / (reps * n) * 1e3
    out = {}
    for st in ("cost", "aggregate", "wta", "median", "speckle", "epilogue", "total"):
        ms, k = ctypes.c_double(), ctypes.c_int()
        if lib.s2p_hip_timing_get(ctx, st.encode(), ctypes.byref(ms), ctypes.byref(k)) == 0 and k.value:
            out[st] = (round(ms.value / (reps * n), 3), k.value // reps)
    print("%s %dx%d x %d, %d per call: %.3f ms per tile host to host; stages per tile (ms, launches per call): %s" % (algo, size, size, nd, n, dt, out), flush=True)
    lib.s2p_hip_ctx_destroy(ctx)
