#!/bin/bash
# round 4, after the fix of k_range_union (one pair of atomics per block): parity of the multi-scale paths, then the kernel trace and
# the configs[3] job with 'mgm_multi' tiles again
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r04
timeout 500 python -m pytest tests/test_gpu_census.py tests/test_gpu_batch.py tests/test_gpu_tile_batch.py -m gpu -q -x -k "scales or multi or fuzz" > gpurun_out/r04/gpu_ms_union.txt 2>&1; tail -2 gpurun_out/r04/gpu_ms_union.txt
bash tools/ms_trace.sh 2>&1 | tail -12
cd "$(dirname "$0")/.."
for B in 1 4; do
  echo "mgm_multi job, $B per call: $(python bench.py --workload config4 --tile-algo mgm_multi --job-batch $B --no-cpu --steps 200 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms per tile, %.1f tiles/s' % (d['ms_per_step'], d['tiles_per_s']))")"
done
python tools/ms_batch_stages.py 1000 256 mgm_multi 1,4
