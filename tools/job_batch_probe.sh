#!/bin/bash
# tools/job_batch_probe.sh -- GPU box: the configs[3] / configs[4] jobs with 1, 4, 8 tiles per library call (s2p_hip_tile_host_batch)
cd "$(dirname "$0")/.."
timeout 600 python -m pytest tests/test_gpu_tile_batch.py -x -q -m gpu 2>&1 | tail -3
for WL in config4 config5; do
  for CFG in "3 1" "2 2" "2 4" "3 4" "2 8"; do set -- $CFG
    echo "$WL in flight $1, tiles per call $2: $(python bench.py --workload $WL --in-flight $1 --job-batch $2 --steps 160 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms per tile, %.1f Mdisp/s' % (d['ms_per_step'], d['value']))")"
  done
done
