#!/bin/bash
# tools/collect_profiles.sh <round-dir, e.g. r01> -- on the GPU box: rocprofv3 kernel-trace stats + separate FETCH_SIZE /
# WRITE_SIZE passes + bench lines for both matchers, written under gpurun_out/profiles/<round>/ (copy into profiles/<round>/).
set -e
cd "$(dirname "$0")/.."
R=${1:-r01}
OUT=gpurun_out/profiles/$R
mkdir -p $OUT
export TMPDIR=/tmp
for algo in census sgbm; do
  CMD="python bench.py --algo $algo --streams 1 --steps 10 --warmup 2 --no-cpu"
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$algo -- $CMD > /dev/null 2>&1
  cp "$(ls gpurun_out/prof_$algo/*/*kernel_stats.csv | head -1)" $OUT/${algo}_1024x1024x128_kernel_stats.csv
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_${algo}_$c -- $CMD > /dev/null 2>&1
  done
  python - "$algo" "$OUT" <<'EOP'
import csv, glob, json, sys, collections
algo, out = sys.argv[1], sys.argv[2]
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("gpurun_out/pmc_%s_%s/*/*counter_collection.csv" % (algo, c))[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        res[k]["%s_KiB_avg" % c] = round(sum(v) / len(v), 1)
        res[k]["launches_%s" % c] = len(v)
json.dump(dict(sorted(res.items())), open("%s/%s_1024x1024x128_pmc_fetch_write.json" % (out, algo), "w"), indent=1)
EOP
  python bench.py --algo $algo --steps 100 --warmup 5 > $OUT/bench_${algo}_1gpu.json 2>/dev/null
  python bench.py --algo $algo --streams 1 --steps 100 --warmup 5 --no-cpu > $OUT/bench_${algo}_1gpu_1stream.json 2>/dev/null
done
python tools/warp_time.py 2>/dev/null | grep -v amdgpu > $OUT/resampler_ms.txt
python tools/tile_time.py 2>/dev/null | grep -v amdgpu > $OUT/tile_pipeline_ms.txt
python tools/mgm_time.py 2>/dev/null | grep -v amdgpu > $OUT/mgm_mode_ms.txt
python tests/perf/tri_time.py 2>/dev/null | grep -v amdgpu > $OUT/triangulation_ms.txt || true
python tests/perf/fusion_time.py 2>/dev/null | grep -v amdgpu > $OUT/fusion_ms.txt
python tests/perf/raster_time.py 2>/dev/null | grep -v amdgpu > $OUT/raster_ms.txt
./tools/probes/hbm_bw > $OUT/hbm_probe.txt 2>/dev/null || true
ls -la $OUT
