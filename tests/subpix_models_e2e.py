"""tests/subpix_models_e2e.py -- (CPU, build container or anywhere the fixtures are) three readings of mgm_multi's SUBPIX=2
 through the reference's end-to-end acceptance tests with the CPU oracle as the per-tile backend and the
'mgm_multi' call's parameters (-S 6, no median, REMOVESMALLCC 25, two predecessors):
  whole   SUBPIX=1 (what the shim runs)
  model0  half-pixel candidates, image 2 sampled half way between its columns (the library's subpix = 2)
  model1  half-pixel candidates whose COST is the mean of their whole-pixel neighbours'
  model2  whole-pixel aggregation, the winner refined on the half-pixel grid before the V fit
Prints one line per model and raster: mean / 99th percentile of |difference| / median, valid count ratio, pass or fail."""
import json, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")   # run from the repository root
import e2e
from oracle import pyoracle as po
MODELS = {"whole": dict(subpix=1), "model0": dict(subpix=2, subpix_model=0), "model1": dict(subpix=2, subpix_model=1), "model2": dict(subpix=1, subpix_model=2)}
which = sys.argv[1:] or list(MODELS)
fxp, fxt = e2e.load("e2e_pair"), e2e.load("e2e_triplet")
out = {}
for name in which:
    be = e2e.Cpu(recursion=1)
    be.params = po.census_params(recursion=1, scales=6, median=0, remove_small_cc=25, **MODELS[name])
    t = time.time()
    origin, dsm, _ = e2e.run_pair(fxp, be)
    r1 = e2e.compare_dsm(dsm, fxp["dsm"], 0.025, 1.0)
    tr = e2e.run_triplet(fxt, be)
    r2 = e2e.compare_dsm(tr["hm1"], fxt["height_map_pair_1"], 0.05, 2.0)
    r3 = e2e.compare_dsm(tr["dsm"], fxt["dsm"], 0.05, 2.0)
    out[name] = {"pair_dsm": r1, "triplet_height_map": r2, "triplet_dsm": r3, "seconds": round(time.time() - t, 1)}
    print(name, json.dumps(out[name]), flush=True)
json.dump(out, open("profiles/r04/subpix_models_e2e.json", "w"), indent=1)
