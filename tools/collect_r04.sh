#!/bin/bash
# tools/collect_r04.sh -- on the GPU box: the round-4 evidence under gpurun_out/profiles/r04/ (copy into profiles/r04/).
#   * rocprofv3 kernel-trace stats + separate FETCH_SIZE / WRITE_SIZE passes (the guide's recipe: counters in their own runs) for the
#     headline's call shape (8 tiles per launch, confidence image included, distinct pairs), the lone launch and the 8-path preview
#   * the broker under load: rocprofv3 kernel trace of the broker process while 64 Pool workers feed it (bench_pool.py
#     --use-running-broker) -> which kernels the device spends its time in when the reference's own execution model drives it
#   * default bench line, job lines, pool sweeps in both modes
set -e
cd "$(dirname "$0")/.."
OUT=gpurun_out/profiles/r04
mkdir -p $OUT
export TMPDIR=/tmp
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
  rm -rf gpurun_out/prof_$name gpurun_out/pmc_${name}_*
done
# the broker under the reference's execution model, profiled
export S2P_HIP_BROKER_DIR=/tmp/s2p_broker_prof
rm -rf gpurun_out/prof_broker $S2P_HIP_BROKER_DIR
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_broker -- python -m s2p_amd.broker --device 0 --idle 30 > $OUT/broker_profiled.log 2>&1 &
for i in $(seq 1 200); do [ -S $S2P_HIP_BROKER_DIR/gpu0.sock ] && break; sleep 0.1; done
python bench_pool.py --workers 64 --tiles 3072 --use-running-broker > $OUT/pool_broker_profiled_64.json 2>/dev/null || true
python -c "
import sys; sys.path.insert(0, '.')
from s2p_amd import broker
broker.shutdown(0)"
wait
cp "$(ls gpurun_out/prof_broker/*/*kernel_stats.csv | head -1)" $OUT/broker_64_workers_kernel_stats.csv || true
rm -rf gpurun_out/prof_broker
unset S2P_HIP_BROKER_DIR
python bench.py > $OUT/bench_default_1gpu.json 2>/dev/null
python bench.py --workload config3 --no-cpu --no-job --no-pool > $OUT/bench_config3_census_mgm.json 2>/dev/null
python bench.py --workload config4 --steps 200 > $OUT/bench_config4_1gpu.json 2>/dev/null
python bench.py --workload config4 --steps 200 --tile-algo mgm_multi > $OUT/bench_config4_mgm_multi_1gpu.json 2>/dev/null
python bench.py --workload config5 --steps 50 > $OUT/bench_config5_1gpu.json 2>/dev/null
python bench.py --workload pool > $OUT/bench_pool_1gpu.json 2>/dev/null
python bench_pool.py --workers 1,4,8,16,32,64 --tiles 512 --broker 1 > $OUT/pool_broker_sweep_final.json 2>/dev/null
python bench_pool.py --workers 1,4,8 --tiles 384 --broker 0 --task-timeout 60 > $OUT/pool_direct_sweep_final.json 2>/dev/null
python bench_pool.py --workers 64 --tiles 1536 --ragged --distinct 64 > $OUT/pool_broker_ragged64_final.json 2>/dev/null
python bench_pool.py --workers 16,64 --tiles 384 --algo mgm_multi --size 1000 --ndisp 256 --task-timeout 120 > $OUT/pool_broker_mgm_multi_1000x256.json 2>/dev/null || true
python bench_pool.py --workers 16,64 --tiles 384 --algo sgbm --task-timeout 120 > $OUT/pool_broker_sgbm.json 2>/dev/null || true
ls -la $OUT
