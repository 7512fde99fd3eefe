#!/bin/bash
# tools/collect_profiles.sh <round-dir, e.g. r02> -- on the GPU box: for every resident-tile workload, the rocprofv3 kernel-trace
# stats, the separate FETCH_SIZE / WRITE_SIZE passes (counters in their own runs, with --kernel-trace only) and the bench line; then
# the job-level workloads, the per-band trace of the MGM launch and the side measurements.  Everything lands under
# gpurun_out/profiles/<round>/ (copy into profiles/<round>/).
set -e
cd "$(dirname "$0")/.."
R=${1:-r02}
OUT=gpurun_out/profiles/$R
mkdir -p $OUT
export TMPDIR=/tmp
# name | bench arguments of the profiled command
WORKLOADS=(
  "census_1024x1024x128|--algo census --streams 1"
  "census_mgm_1024x1024x128|--algo census --recursion 1 --streams 1"
  "sgbm_1024x1024x128|--algo sgbm --streams 1"
  "census_1000x1000x256|--workload config3 --algo census --streams 1"
  "census_mgm_1000x1000x256|--workload config3 --algo census --recursion 1 --streams 1"
  "census_2048x2048x256|--algo census --size 2048 --ndisp 256 --streams 1"
)
for wl in "${WORKLOADS[@]}"; do
  name=${wl%%|*}; args=${wl#*|}
  CMD="python bench.py $args --steps 10 --warmup 2 --no-cpu"
  rm -rf gpurun_out/prof_$name gpurun_out/pmc_${name}_*
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$name -- $CMD > /dev/null 2>&1
  cp "$(ls gpurun_out/prof_$name/*/*kernel_stats.csv | head -1)" $OUT/${name}_kernel_stats.csv
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_${name}_$c -- $CMD > /dev/null 2>&1
  done
  python - "$name" "$OUT" <<'EOP'
import csv, glob, json, sys, collections
name, out = sys.argv[1], sys.argv[2]
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("gpurun_out/pmc_%s_%s/*/*counter_collection.csv" % (name, c))[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        res[k]["%s_KiB_avg" % c] = round(sum(v) / len(v), 1)
        res[k]["launches_%s" % c] = len(v)
json.dump(dict(sorted(res.items())), open("%s/%s_pmc_fetch_write.json" % (out, name), "w"), indent=1)
EOP
  rm -rf gpurun_out/prof_$name gpurun_out/pmc_${name}_*
done
# bench lines (un-profiled): they pick the PMC files above up as `roofline.traffic` once those are committed under profiles/
cp $OUT/*_pmc_fetch_write.json profiles/$R/ 2>/dev/null || { mkdir -p profiles/$R; cp $OUT/*_pmc_fetch_write.json profiles/$R/; }
python bench.py --steps 100 --warmup 5 > $OUT/bench_census_1gpu.json 2>/dev/null
python bench.py --algo sgbm --steps 100 --warmup 5 > $OUT/bench_sgbm_1gpu.json 2>/dev/null
python bench.py --recursion 1 --steps 60 --warmup 6 --no-cpu > $OUT/bench_census_mgm_3streams.json 2>/dev/null
python bench.py --recursion 1 --streams 1 --steps 40 --warmup 4 --no-cpu > $OUT/bench_census_mgm_1stream.json 2>/dev/null
python bench.py --workload config3 --steps 60 --warmup 5 --no-cpu > $OUT/bench_config3_census.json 2>/dev/null
python bench.py --workload config3 --recursion 1 --steps 40 --warmup 6 --no-cpu > $OUT/bench_config3_census_mgm.json 2>/dev/null
python bench.py --size 2048 --ndisp 256 --steps 20 --warmup 3 --no-cpu > $OUT/bench_census_2048x2048x256.json 2>/dev/null
python bench.py --workload config4 --steps 100 > $OUT/bench_config4_1gpu.json 2>/dev/null
python bench.py --workload config4 --steps 100 --tile-algo mgm_multi > $OUT/bench_config4_mgm_multi_1gpu.json 2>/dev/null
python bench.py --workload config5 --steps 50 > $OUT/bench_config5_1gpu.json 2>/dev/null
# the per-band trace of the MGM launch (needs build/variants/cur_trace: tools/sweep_mgm.sh build)
if [ -f build/variants/cur_trace/libs2p_hip.so ]; then
  cp s2p_amd/lib/libs2p_hip.so build/libs2p_hip.orig.so
  cp build/variants/cur_trace/libs2p_hip.so s2p_amd/lib/libs2p_hip.so
  python tools/mgm_trace.py run 2> gpurun_out/trace_raw.log || true
  python tools/mgm_trace.py < gpurun_out/trace_raw.log > $OUT/mgm_band_trace.txt || true
  rm -f gpurun_out/trace_raw.log
  cp build/libs2p_hip.orig.so s2p_amd/lib/libs2p_hip.so
fi
python tools/shim_time.py 2>/dev/null | grep -v amdgpu > $OUT/shim_ms.txt || true
python tools/warp_time.py 2>/dev/null | grep -v amdgpu > $OUT/resampler_ms.txt || true
python tools/tile_time.py 2>/dev/null | grep -v amdgpu > $OUT/tile_pipeline_ms.txt || true
python tools/mgm_time.py 2>/dev/null | grep -v amdgpu > $OUT/mgm_mode_ms.txt || true
python tests/perf/tri_time.py 2>/dev/null | grep -v amdgpu > $OUT/triangulation_ms.txt || true
python tests/perf/fusion_time.py 2>/dev/null | grep -v amdgpu > $OUT/fusion_ms.txt || true
python tests/perf/raster_time.py 2>/dev/null | grep -v amdgpu > $OUT/raster_ms.txt || true
./tools/probes/hbm_bw > $OUT/hbm_probe.txt 2>/dev/null || true
ls -la $OUT
