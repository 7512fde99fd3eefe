#!/bin/bash
# tools/workers_probe.sh -- GPU box: persistent workers of the 8-tile MGM launch (S2P_MGM_WORKERS) x tile streams, whole tiles
cd "$(dirname "$0")/.."
for W in ${WORKERS:-512 384 320 256}; do for S in ${STREAMS:-2 3}; do
  echo "workers $W streams $S: $(S2P_MGM_WORKERS=$W python bench.py --no-cpu --no-job --no-pool --steps 4 --streams $S 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms per tile | aggregate launch %.4f wta %.4f' % (d['ms_per_tile'], d['stage_ms']['aggregate'], d['stage_ms']['wta']))")"
done; done
