#!/bin/bash
# the tri-stereo job (BASELINE configs[4]) with 1 / 2 / 4 tiles (x 2 pairs) per library call
cd "$(dirname "$0")/.."
for B in 1 2 4; do for F in 3 2; do
  echo "config5, $B tiles per call, $F in flight: $(python bench.py --workload config5 --job-batch $B --in-flight $F --no-cpu --steps 100 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms per tile, %.1f tiles/s' % (d['ms_per_step'], d['tiles_per_s']))")"
done; done
