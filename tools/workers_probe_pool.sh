#!/bin/bash
# tools/workers_probe_pool.sh -- GPU box: persistent workers per batched MGM launch (S2P_MGM_WORKERS) through the broker and in the configs[3] job
# (profiles/r04/workers_probe.txt)
cd "$(dirname "$0")/.."
for W in 512 448 384; do
  echo "broker, workers $W: $(S2P_MGM_WORKERS=$W python bench_pool.py --workers 64 --tiles 2048 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); p=d['pools'][0]; print('steady %s tiles/s, fork->join %s' % (p['steady']['tiles_per_s'], p['tiles_per_s_fork_to_join']))")"
  echo "config4 job, workers $W: $(S2P_MGM_WORKERS=$W python bench.py --workload config4 --no-cpu --steps 200 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms per tile' % d['ms_per_step'])")"
done
