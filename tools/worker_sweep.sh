#!/bin/bash
# tools/worker_sweep.sh -- GPU box: per-tile time of the MGM-mode bench line against the number of workers of a launch
# (S2P_MGM_WORKERS) and the tiles in flight (--streams)
for W in ${WORKERS:-192 256 512}; do
  for S in ${STREAMS:-1 2 3}; do
    echo "workers $W streams $S: $(S2P_MGM_WORKERS=$W python bench.py --no-cpu --no-job --steps 3 --batch 96 --batch-launch 1 --streams $S 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_tile'], d['stage_ms']['aggregate'])")"
  done
done
