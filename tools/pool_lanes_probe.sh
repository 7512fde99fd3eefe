#!/bin/bash
# tools/pool_lanes_probe.sh -- GPU box: lanes x tiles-per-call of the broker with 16 / 32 / 64 Pool workers (profiles/r04/pool_lanes_probe.txt)
cd "$(dirname "$0")/.."
for cfg in "3 8" "5 8" "6 4" "4 4" "8 2"; do
  set -- $cfg
  python bench_pool.py --workers 16,32,64 --tiles 512 --lanes $1 --max-batch $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('lanes $1 batch $2:', ' | '.join('P %d: steady %s, f2j %s, b/call %s' % (p['workers'], (p.get('steady') or {}).get('tiles_per_s'), p['tiles_per_s_fork_to_join'], p.get('mean_tiles_per_library_call')) for p in d['pools']))"
done
