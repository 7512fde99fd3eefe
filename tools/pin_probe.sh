#!/bin/bash
# tools/pin_probe.sh -- GPU box: the workers' arenas page-locked eagerly / lazily / never (S2P_HIP_BROKER_PIN), three Pools of 64 x 1 536 tiles each
# (profiles/r04/pin_probe.txt; measured before the arenas outlived their workers)
cd "$(dirname "$0")/.."
for rep in 1 2; do for MODE in eager 1 0; do
  echo "pin=$MODE: $(S2P_HIP_BROKER_PIN=$MODE python bench_pool.py --workers 64,64,64 --tiles 1536 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(' | '.join('f2j %s steady %s cold med/max %s/%s attach med %s' % (p['tiles_per_s_fork_to_join'], p['steady']['tiles_per_s'], p['cold_start_s']['median'], p['cold_start_s']['max'], (p.get('broker_connect_attach_ms') or {}).get('median')) for p in d['pools']))")"
done; done
