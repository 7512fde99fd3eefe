#!/bin/bash
# the broker under the reference's execution model, profiled: rocprofv3 kernel trace of the BROKER process while 64 Pool workers feed it
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R
OUT=gpurun_out/r04; mkdir -p $OUT
export S2P_HIP_BROKER_DIR=/tmp/s2p_broker_prof
rm -rf gpurun_out/prof_broker $S2P_HIP_BROKER_DIR; mkdir -p -m 700 $S2P_HIP_BROKER_DIR
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_broker -- python -m s2p_amd.broker --device 0 --idle 30 > $OUT/broker_profiled.log 2>&1 &
for i in $(seq 1 300); do [ -S $S2P_HIP_BROKER_DIR/gpu0.sock ] && break; sleep 0.1; done
timeout 200 python bench_pool.py --workers 64 --tiles 3072 --use-running-broker > $OUT/pool_broker_profiled_64.json 2>/dev/null || true
python -c "
import sys; sys.path.insert(0, '.')
from s2p_amd import broker
broker.shutdown(0)"
wait
cp "$(ls gpurun_out/prof_broker/*/*kernel_stats.csv | head -1)" $OUT/broker_64_workers_kernel_stats.csv || true
rm -rf gpurun_out/prof_broker
python - <<'PY'
import csv, json
rows = list(csv.DictReader(open("gpurun_out/r04/broker_64_workers_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("broker kernels: %.1f ms in total" % (tot / 1e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:6]:
    print("%9.3f ms %6d calls %8.1f us avg %5.1f%%  %s" % (float(r["TotalDurationNs"]) / 1e6, int(r["Calls"]), float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot, r["Name"][:80]))
d = json.load(open("gpurun_out/r04/pool_broker_profiled_64.json"))
p = d["pools"][-1]
print("pool under the profiler: steady", p["steady"]["tiles_per_s"], "fork->join", p["tiles_per_s_fork_to_join"], "lanes busy", p.get("broker", {}).get("lane_busy_frac_of_wall"))
PY
