"""tools/mgm_trace.py -- per-band start/end times of the band-pipelined MGM launch.
Needs the -DS2P_MGM_TRACE build (tools/sweep_mgm.sh trace -> build/trace/libs2p_hip.so copied over s2p_amd/lib/):
the kernel stamps wall_clock64() (100 MHz) per band, the library prints them to stderr after the stream drains.
  python tools/mgm_trace.py run [size [recursion]] 2> raw.log ; python tools/mgm_trace.py < raw.log"""
import os, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
if len(sys.argv) > 1 and sys.argv[1] == "run":
    from s2p_amd import _lib as L
    from helpers import synth_pair
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    im1, im2 = synth_pair(7, size, size, lambda x, y: 40 * np.sin(2 * np.pi * x / 512.) * np.cos(2 * np.pi * y / 512.))
    p = L.default_census_params(recursion=int(sys.argv[3]) if len(sys.argv) > 3 else 2)
    os.environ["S2P_MGM_IMPL"] = "bands"
    for rep in range(3):
        L.census_sgm(im1, im2, -64, 63, params=p, want_conf=False)
    sys.exit(0)
runs, cur = [], {}
for line in sys.stdin:
    f = line.split()
    if f and f[0] == "MGMTRACE_END":
        runs.append(cur); cur = {}
    elif f and f[0] == "MGMWAVES":
        cur.setdefault("waves", {})[(int(f[1]), int(f[2]))] = list(map(int, f[3:]))
    elif len(f) == 11 and f[0] == "MGMTRACE":
        q, band, s0, s1, t0, t1, wait, retries, xcc, mhz = map(int, f[1:])
        cur[(q, band)] = (s0, s1, t0, t1, wait, retries, xcc, mhz)
rows = runs[-1]
waves = rows.pop("waves", {})
tmin = min(v[2] for v in rows.values())
tick = 0.01   # us per wall_clock64 tick (100 MHz)
for q in range(12):
    bands = sorted(b for (qq, b) in rows if qq == q)
    if not bands:
        continue
    st = np.array([(rows[(q, b)][2] - tmin) * tick for b in bands]); en = np.array([(rows[(q, b)][3] - tmin) * tick for b in bands])
    steps = np.array([rows[(q, b)][1] - rows[(q, b)][0] for b in bands])
    per = (en - st) / np.maximum(steps, 1)
    gaps = np.diff(st)
    wait = np.array([rows[(q, b)][4] * tick for b in bands]); retr = np.array([rows[(q, b)][5] for b in bands])
    xcc = [rows[(q, b)][6] & 15 for b in bands]
    mhz = np.array([rows[(q, b)][7] / 10.0 for b in bands])
    print("q=%2d bands=%3d first gate %.1f last end %.1f us | steps/band med %d | us/step band0 %.3f med %.3f max %.3f | start-to-start gap med %.2f us | chunk wait med %.1f us/band, %.0f retries | shader clock med %.0f MHz | xcc %s"
          % (q, len(bands), st[0], en[-1], int(np.median(steps)), per[0], float(np.median(per)), per.max(), float(np.median(gaps)) if len(gaps) else 0.0,
             float(np.median(wait)), float(np.median(retr)), float(np.median(mhz)), "".join(str(x) for x in xcc[:24])))
    if q in (0, 5):
        for b in bands[::8] + [bands[-1]]:
            i = bands.index(b)
            print("      band %3d steps %4d gate %7.1f end %7.1f us/step %.3f wait %.1f us retries %d" % (b, steps[i], st[i], en[i], per[i], wait[i], retr[i]))
            wv = waves.get((q, b))
            if wv and len(wv) >= 24:
                n = max(steps[i], 1)
                nw = max(k + 1 for k in range(8) if wv[16 + k]) if any(wv[16:24]) else 4
                M = (1 << 40) - 1
                print("          per wave, cycles per step spent polling for data %s (in %s %% of the steps) | for back-pressure %s (%s %%)   [incl. the wait for the first point]" % (
                    " ".join("%d" % ((wv[k] & M) / n) for k in range(nw)), " ".join("%d" % (100 * (wv[k] >> 40) / n) for k in range(nw)),
                    " ".join("%d" % ((wv[8 + k] & M) / n) for k in range(nw)), " ".join("%d" % (100 * (wv[8 + k] >> 40) / n) for k in range(nw))))
print("launch span (first gate .. last end) %.1f us" % ((max(v[3] for v in rows.values()) - tmin) * tick))
