"""tools/cold_start.py -- what a freshly forked Pool worker pays before its first result: each line is one
stage of the first compute_disparity_map('mgm') call of a fresh process, then the steady state.  `--procs N` starts N such
processes at once (the Pool's situation: every worker initialises HIP at the same moment)."""
import os, subprocess, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
if len(sys.argv) > 1 and sys.argv[1] == "--procs":
    n = int(sys.argv[2])
    t0 = time.perf_counter()
    ps = [subprocess.Popen([sys.executable, __file__, "--quiet"], stdout=subprocess.PIPE, text=True) for _ in range(n)]
    outs = [p.communicate()[0].strip().splitlines()[-1] for p in ps]
    print("%d processes at once, wall %.2f s" % (n, time.perf_counter() - t0))
    for o in sorted(outs)[:: max(1, n // 8)]:
        print("  ", o)
    sys.exit(0)
quiet = "--quiet" in sys.argv
T = [("start", time.perf_counter())]
import ctypes
import numpy as np
from s2p_amd import _lib
T.append(("import numpy + s2p_amd", time.perf_counter()))
L = _lib.lib(); T.append(("dlopen libs2p_hip.so", time.perf_counter()))
n = _lib.device_count(); T.append(("hipGetDeviceCount (runtime init)", time.perf_counter()))
c = _lib.context(0); T.append(("ctx_create (stream, abort word)", time.perf_counter()))
a = _lib.pinned_empty((1024, 1024)); b = _lib.pinned_empty((1024, 1024)); T.append(("2 x 4 MB pinned", time.perf_counter()))
rng = np.random.default_rng(0); a[:] = rng.random((1024, 1024), np.float32) * 255; b[:] = np.roll(a, 3, 1)
T.append(("fill", time.perf_counter()))
p = _lib.default_census_params(recursion=2)
for k in range(4):
    r = _lib.census_sgm(a, b, -64, 63, params=p, pinned=True); T.append(("census_sgm call %d (1024^2 x 128, MGM)" % k, time.perf_counter()))
line = " | ".join("%s %.1f ms" % (T[i][0], (T[i][1] - T[i - 1][1]) * 1e3) for i in range(1, len(T)))
print(line if quiet else line.replace(" | ", "\n"))
