#!/bin/bash
# tools/collect_round.sh [ROUND] -- on the GPU box: a round's evidence under gpurun_out/profiles/<ROUND>/ (copy into profiles/<ROUND>/; default r06).
#   * known-byte calibration of FETCH_SIZE / WRITE_SIZE incl. the band kernel's own read/write mix (tools/pmc_calib.sh)
#   * rocprofv3 kernel-trace stats + separate FETCH_SIZE / WRITE_SIZE passes (the guide's recipe: counters in their own runs) for EVERY call
#     shape the bench line reports: the headline (8 tiles per launch, confidence image, distinct pairs), the lone
#     launch, the 8-path preview, `sgbm` 1024^2 x 128 and the configs[3] job (4 tiles of 1000^2 x 256 per call, 16 candidates per lane)
#   * the kernel TRACE of the headline command itself (three streams, 8 tiles per call) -> union of k_mgm_bands' busy intervals per launch
#     (tools/inflight_union.py): the in-flight figure of the SAME round and call shape the bench line quotes
#   * SQ counters of the shipped kernels (tools/sq_probe.sh), one point at s2p's default tile geometry (tools/default_tile_time.py)
#   * the default bench line (run last: it reads the PMC files and the calibration of this round), sgbm / job lines
set -e
cd "$(dirname "$0")/.."
ROUND=${1:-r06}
OUT=gpurun_out/profiles/$ROUND
mkdir -p $OUT profiles/$ROUND
export TMPDIR=/tmp
tools/pmc_calib.sh $ROUND > /dev/null 2>&1 || true
[ -f $OUT/pmc_calibration.json ] && cp $OUT/pmc_calibration.json $OUT/pmc_calibration_timing.txt profiles/$ROUND/
WORKLOADS=(
  "census_mgm3_b8_1024x1024x128|--recursion 2 --streams 1 --batch-launch 8 --batch 16 --steps 2 --warmup 1"
  "census_mgm3_1024x1024x128|--recursion 2 --streams 1 --batch-launch 1 --batch 6 --steps 2 --warmup 1"
  "census_1024x1024x128|--recursion 0 --streams 1 --batch 6 --steps 2 --warmup 1"
  "sgbm_1024x1024x128|--algo sgbm --streams 1 --batch 6 --steps 2 --warmup 1"
  "job_mgm_b4_1000x1000x256|--workload config4 --steps 48 --in-flight 1 --warmup 1"
)
for wl in "${WORKLOADS[@]}"; do
  name=${wl%%|*}; args=${wl#*|}
  CMD="python bench.py $args --no-cpu --no-job --no-pool"
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
    per = collections.defaultdict(float)          # (kernel, dispatch): the rows of one dispatch summed (one row per dispatch on this stack)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c:
            per[(r["Kernel_Name"], r.get("Dispatch_Id", ""))] += float(r["Counter_Value"])
    acc = collections.defaultdict(list)
    for (k, _), v in per.items():
        acc[k].append(v)
    for k, v in acc.items():
        res[k]["%s_KiB_avg" % c] = round(sum(v) / len(v), 1)
        res[k]["launches_%s" % c] = len(v)
json.dump(dict(sorted(res.items())), open("%s/%s_pmc_fetch_write.json" % (out, name), "w"), indent=1)
EOP
  cp $OUT/${name}_pmc_fetch_write.json $OUT/${name}_kernel_stats.csv profiles/$ROUND/
  rm -rf gpurun_out/prof_$name gpurun_out/pmc_${name}_*
done
# the headline command itself under the kernel trace: k_mgm_bands with calls in flight
rm -rf gpurun_out/prof_inflight
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_inflight -- python bench.py --steps 4 --warmup 2 --no-cpu --no-job --no-pool > /dev/null 2>&1
python tools/inflight_union.py "$(ls gpurun_out/prof_inflight/*/*kernel_trace.csv | head -1)" $OUT/mgm_inflight_b8_1024x1024x128.json k_mgm_bands
cp $OUT/mgm_inflight_b8_1024x1024x128.json profiles/$ROUND/
rm -rf gpurun_out/prof_inflight
tools/sq_probe.sh $ROUND > /dev/null 2>&1 || true
python tools/default_tile_time.py > $OUT/default_tile_time.txt 2>/dev/null || true
python tools/config2_time.py > $OUT/config2_time.txt 2>/dev/null || true
python bench.py > $OUT/bench_default_1gpu.json 2>$OUT/bench_default_1gpu.err
python bench.py --algo sgbm --no-pool --no-job --no-cpu > $OUT/bench_sgbm_1gpu.json 2>/dev/null
python bench.py --workload config4 --steps 200 > $OUT/bench_config4_1gpu.json 2>/dev/null
python bench.py --workload config4 --steps 200 --tile-algo mgm_multi > $OUT/bench_config4_mgm_multi_1gpu.json 2>/dev/null
python bench.py --workload config5 --steps 50 > $OUT/bench_config5_1gpu.json 2>/dev/null
ls -la $OUT
