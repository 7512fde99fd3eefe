"""tools/mgm_stress.py [seconds] -- (GPU box) the band-pipelined MGM launch under contention: random tile shapes and ranges,
each computed once on a quiet device and then repeatedly on three streams at once (own context per thread, other launches
of other shapes in flight beside it).  Any race of the hand-off protocol shows as a mismatch or as a timeout error."""
import sys, threading, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from helpers import synth_pair
from s2p_amd import _lib as L

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(12345)
cases = []
for k in range(40):
    h, w = int(rng.integers(40, 900)), int(rng.integers(40, 900))
    span = int(rng.choice([6, 15, 31, 63, 100, 127, 200, 255, 400]))
    lo = -int(rng.integers(0, span))
    im1, im2 = synth_pair(100 + k, h, w, lambda x, y: np.clip(lo + 0.5 * span + 0.3 * span * np.sin(x / 37.) * np.cos(y / 29.), -200, 200), nan=bool(k % 3 == 0))
    p = L.default_census_params(recursion=1, nb_dir=int(rng.choice([8, 8, 4])), median=int(rng.integers(0, 2)))
    cases.append((im1, im2, lo, lo + span, p))
t0 = time.time()
quiet = [L.census_sgm(a, b, lo, hi, params=p, want_conf=False) for a, b, lo, hi, p in cases]
print("quiet pass: %d cases in %.1f s" % (len(cases), time.time() - t0), flush=True)
streams = []
for k in range(3):
    import ctypes
    c = ctypes.c_void_p()
    L.check(L.lib().s2p_hip_ctx_create(L.default_device(), None, ctypes.byref(c)))
    streams.append(c)
bad, done, errs = [], [0, 0, 0], []
stop = time.time() + budget

def worker(k):
    r = np.random.default_rng(k)
    while time.time() < stop:
        i = int(r.integers(0, len(cases)))
        a, b, lo, hi, p = cases[i]
        try:
            out = L.census_sgm(a, b, lo, hi, params=p, want_conf=False, ctx=streams[k])
        except Exception as e:
            errs.append((i, repr(e))); return
        if not (np.array_equal(out["disp"], quiet[i]["disp"], equal_nan=True) and np.array_equal(out["mask"], quiet[i]["mask"])):
            bad.append(i)
        done[k] += 1

th = [threading.Thread(target=worker, args=(k,)) for k in range(3)]
[t.start() for t in th]; [t.join() for t in th]
print("contended passes: %s calls on 3 streams in %.0f s, mismatches %d, errors %d %s" % (done, budget, len(bad), len(errs), errs[:2]))
sys.exit(1 if bad or errs else 0)
