#!/bin/bash
# tools/collect_r05.sh -- on the GPU box: the round-5 evidence under gpurun_out/profiles/r05/ (copy into profiles/r05/).
#   * rocprofv3 kernel-trace stats + separate FETCH_SIZE / WRITE_SIZE passes (the guide's recipe: counters in their own runs) for the
#     headline's call shape (8 tiles per launch, confidence image included, distinct pairs), the lone launch and the 8-path preview
#   * the kernel TRACE of the headline command itself (three streams, 8 tiles per call) -> union of k_mgm_bands' busy intervals per launch
#     (tools/inflight_union.py): the in-flight figure of the SAME round and call shape the bench line quotes
#   * the default bench line (run last: it reads the PMC files and the calibration of this round), job / pool lines
set -e
cd "$(dirname "$0")/.."
OUT=gpurun_out/profiles/r05
mkdir -p $OUT profiles/r05
export TMPDIR=/tmp
[ -f profiles/r05/pmc_calibration.json ] || tools/pmc_calib.sh > /dev/null 2>&1 || true
[ -f $OUT/pmc_calibration.json ] && cp $OUT/pmc_calibration.json profiles/r05/
WORKLOADS=(
  "census_mgm3_b8_1024x1024x128|--recursion 2 --streams 1 --batch-launch 8 --batch 16"
  "census_mgm3_1024x1024x128|--recursion 2 --streams 1 --batch-launch 1 --batch 6"
  "census_1024x1024x128|--recursion 0 --streams 1 --batch 6"
)
for wl in "${WORKLOADS[@]}"; do
  name=${wl%%|*}; args=${wl#*|}
  CMD="python bench.py $args --steps 2 --warmup 1 --no-cpu --no-job --no-pool"
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
  cp $OUT/${name}_pmc_fetch_write.json profiles/r05/
  rm -rf gpurun_out/prof_$name gpurun_out/pmc_${name}_*
done
# the headline command itself under the kernel trace: k_mgm_bands with calls in flight
rm -rf gpurun_out/prof_inflight
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_inflight -- python bench.py --steps 4 --warmup 2 --no-cpu --no-job --no-pool > /dev/null 2>&1
python tools/inflight_union.py "$(ls gpurun_out/prof_inflight/*/*kernel_trace.csv | head -1)" $OUT/mgm_inflight_b8_1024x1024x128.json k_mgm_bands
cp $OUT/mgm_inflight_b8_1024x1024x128.json profiles/r05/
rm -rf gpurun_out/prof_inflight
python bench.py > $OUT/bench_default_1gpu.json 2>$OUT/bench_default_1gpu.err
python bench.py --algo sgbm --no-pool --no-job --no-cpu > $OUT/bench_sgbm_1gpu.json 2>/dev/null
python bench.py --workload config4 --steps 200 > $OUT/bench_config4_1gpu.json 2>/dev/null
python bench.py --workload config4 --steps 200 --tile-algo mgm_multi > $OUT/bench_config4_mgm_multi_1gpu.json 2>/dev/null
python bench.py --workload config5 --steps 50 > $OUT/bench_config5_1gpu.json 2>/dev/null
ls -la $OUT
