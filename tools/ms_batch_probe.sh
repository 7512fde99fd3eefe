#!/bin/bash
# tools/ms_batch_probe.sh -- GPU box: the configs[3] job with 'mgm_multi' tiles, 1 / 2 / 4 / 8 tiles per library call (round 4: multi-scale
# tiles batch level by level: one read-back per level and batch, one aggregation launch per level)
cd "$(dirname "$0")/.."
for B in 1 2 4 8; do
  echo "mgm_multi, $B per call: $(python bench.py --workload config4 --tile-algo mgm_multi --job-batch $B --no-cpu --steps 200 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms per tile, %.1f tiles/s' % (d['ms_per_step'], d['tiles_per_s']))")"
done
for B in 1 4; do
  echo "mgm, $B per call: $(python bench.py --workload config4 --tile-algo mgm --job-batch $B --no-cpu --steps 200 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms per tile, %.1f tiles/s' % (d['ms_per_step'], d['tiles_per_s']))")"
done
echo "pool, mgm_multi 1000^2 x 256 through the broker:"
python bench_pool.py --workers 16,64 --tiles 384 --algo mgm_multi --size 1000 --ndisp 256 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read())
for p in d['pools']: print('  P', p['workers'], 'fork->join', p['tiles_per_s_fork_to_join'], 'steady', (p.get('steady') or {}).get('tiles_per_s'), 'tiles/call', p.get('mean_tiles_per_library_call'))"
