#!/bin/bash
# round-4 end sequence: GPU tests, smoke, the default bench line, then what changed late in the round (mgm_multi job and Pool, 16 directions' kernel stats)
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04/gpu_tests.txt 2>&1; tail -4 gpurun_out/r04/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04/smoke.txt 2>&1; tail -2 gpurun_out/r04/smoke.txt
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r04/bench_default_1gpu.json 2> gpurun_out/r04/bench_default.err; tail -c 300 gpurun_out/r04/bench_default.err; head -c 400 gpurun_out/r04/bench_default_1gpu.json; echo
timeout 200 python bench.py --workload config4 --tile-algo mgm_multi --no-cpu --steps 200 > gpurun_out/r04/bench_config4_mgm_multi_1gpu.json 2>/dev/null; head -c 300 gpurun_out/r04/bench_config4_mgm_multi_1gpu.json; echo
timeout 200 python bench_pool.py --workers 64 --tiles 384 --algo mgm_multi --size 1000 --ndisp 256 > gpurun_out/r04/pool_broker_mgm_multi_1000x256.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/r04/pool_broker_mgm_multi_1000x256.json'))
for p in d['pools']: print('pool mgm_multi P', p['workers'], 'fork->join', p['tiles_per_s_fork_to_join'], 'steady', (p.get('steady') or {}).get('tiles_per_s'), 'tiles/call', p.get('mean_tiles_per_library_call'))"
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r04/trace_dir16
( cd $R && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04/trace_dir16 -o t -- python tools/dir16_time.py > gpurun_out/r04/trace_dir16/run.txt 2>&1 )
cd $R
find gpurun_out/r04/trace_dir16 -name "*.csv" ! -name "*kernel_stats.csv" -delete; find gpurun_out/r04/trace_dir16 -name "*.db" -delete
grep "directions" gpurun_out/r04/trace_dir16/run.txt | tail -5
ls /tmp/s2p_hip_broker_0/ 2>/dev/null; pgrep -af "s2p_amd.broker" || true
