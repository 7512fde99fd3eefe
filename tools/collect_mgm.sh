#!/bin/bash
# tools/collect_mgm.sh <round-dir> -- on the GPU box: the MGM-mode evidence (kernel-trace stats of `bench.py --recursion 1`,
# FETCH/WRITE bytes of k_mgm_bands, per-mode stage times, bench lines that carry `mgm_recursion`), under
# gpurun_out/profiles/<round>/ (copy into profiles/<round>/).
set -e
cd "$(dirname "$0")/.."
R=${1:-r01}
OUT=gpurun_out/profiles/$R
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python bench.py --algo census --recursion 1 --streams 1 --steps 10 --warmup 2 --no-cpu"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_mgm -- $CMD > /dev/null 2>&1
cp "$(ls gpurun_out/prof_mgm/*/*kernel_stats.csv | head -1)" $OUT/census_mgm_1024x1024x128_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_mgm_$c -- $CMD > /dev/null 2>&1
done
python - "$OUT" <<'EOP'
import csv, glob, json, sys, collections
out = sys.argv[1]
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("gpurun_out/pmc_mgm_%s/*/*counter_collection.csv" % c)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        res[k]["%s_KiB_avg" % c] = round(sum(v) / len(v), 1)
        res[k]["launches_%s" % c] = len(v)
json.dump(dict(sorted(res.items())), open("%s/census_mgm_1024x1024x128_pmc_fetch_write.json" % out, "w"), indent=1)
EOP
python tools/mgm_time.py 2>/dev/null | grep -v amdgpu > $OUT/mgm_mode_ms.txt
for st in 1 2 4; do
  python bench.py --algo census --recursion 1 --streams $st --steps 40 --warmup 8 --no-cpu 2>/dev/null > $OUT/bench_census_mgm_${st}stream.json
done
python bench.py --algo census --steps 100 --warmup 5 > $OUT/bench_census_1gpu.json 2>/dev/null
python bench.py --algo census --streams 1 --steps 100 --warmup 5 --no-cpu > $OUT/bench_census_1gpu_1stream.json 2>/dev/null
ls -la $OUT
