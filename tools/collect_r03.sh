#!/bin/bash
# tools/collect_r03.sh -- on the GPU box: the round-3 evidence under gpurun_out/profiles/r03/ (copy into profiles/r03/).
#   * default bench line (MGM recursion headline, preview_8path, job, cpu_baseline with all_cores)
#   * rocprofv3 kernel-trace stats + separate FETCH_SIZE / WRITE_SIZE passes for the headline and the preview mode, one tile stream
#   * the 3-stream kernel trace of the headline -> union of k_mgm_bands busy intervals per launch (tools/inflight_union.py)
#   * job lines: config4 'mgm' / 'mgm_multi', config5; resident config3 shapes; file-level shim; pinned probe
set -e
cd "$(dirname "$0")/.."
OUT=gpurun_out/profiles/r03
mkdir -p $OUT profiles/r03
export TMPDIR=/tmp
WORKLOADS=(
  "census_mgm3_b8_1024x1024x128|--recursion 2 --streams 1 --batch-launch 8 --batch 16"
  "census_mgm3_1024x1024x128|--recursion 2 --streams 1 --batch-launch 1 --batch 6"
  "census_mgm_1024x1024x128|--recursion 1 --streams 1 --batch-launch 1 --batch 6"
  "census_1024x1024x128|--recursion 0 --streams 1 --batch 6"
  "census_mgm3_1000x1000x256|--workload config3 --recursion 2 --streams 1 --batch-launch 1 --batch 6"
)
# QUICK=<name prefix>: only the matching workloads' kernel stats / PMC passes and the default bench line (a kernel of one workload changed)
for wl in "${WORKLOADS[@]}"; do
  name=${wl%%|*}; args=${wl#*|}
  if [ -n "$QUICK" ] && [[ "$name" != $QUICK* ]]; then continue; fi
  CMD="python bench.py $args --steps 2 --warmup 1 --no-cpu --no-job"
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
cp $OUT/*_pmc_fetch_write.json profiles/r03/
if [ -n "$QUICK" ]; then python bench.py > $OUT/bench_default_1gpu.json 2>/dev/null; ls -la $OUT; exit 0; fi
# tiles in flight: the kernel trace of the 3-stream headline run
rm -rf gpurun_out/prof_inflight
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_inflight -- python bench.py --no-cpu --no-job --steps 2 --batch 48 --batch-launch 1 > /dev/null 2>&1
python tools/inflight_union.py "$(ls gpurun_out/prof_inflight/*/*kernel_trace.csv | head -1)" $OUT/mgm_inflight_1024x1024x128.json
cp $OUT/mgm_inflight_1024x1024x128.json profiles/r03/
rm -rf gpurun_out/prof_inflight
python bench.py > $OUT/bench_default_1gpu.json 2>/dev/null
python bench.py --batch-launch 1 --no-cpu --no-job > $OUT/bench_census_mgm3_1tile_per_call_3streams.json 2>/dev/null
python bench.py --recursion 1 --no-cpu --no-job > $OUT/bench_census_mgm2pred_1gpu.json 2>/dev/null
python bench.py --recursion 0 --no-cpu --no-job > $OUT/bench_census_8path_1gpu.json 2>/dev/null
python bench.py --algo sgbm --no-job > $OUT/bench_sgbm_1gpu.json 2>/dev/null
python bench.py --workload config3 --no-cpu --no-job > $OUT/bench_config3_census_mgm.json 2>/dev/null
python bench.py --workload config4 --steps 200 > $OUT/bench_config4_1gpu.json 2>/dev/null
python bench.py --workload config4 --steps 100 --tile-algo mgm_multi > $OUT/bench_config4_mgm_multi_1gpu.json 2>/dev/null
python bench.py --workload config5 --steps 50 > $OUT/bench_config5_1gpu.json 2>/dev/null
python tools/pinned_probe.py 2>/dev/null | grep -v amdgpu > $OUT/pinned_probe.txt || true
python tools/shim_time.py 2>/dev/null | grep -v amdgpu > $OUT/shim_ms.txt || true
python tools/shim_breakdown.py 2>/dev/null | grep -v amdgpu > $OUT/shim_breakdown.txt || true
NBS="1 2 4 8" STREAMS="1 2 3" STAGGERS="-1" bash tools/batch_sweep.sh > $OUT/batch_sweep.txt 2>/dev/null || true
ls -la $OUT
