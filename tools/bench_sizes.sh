#!/bin/bash
# tools/bench_sizes.sh -- eager vs hipGraph replay, 1 vs 2 tile streams, for a large and a small tile
cd "$(dirname "$0")/.."
for sz in 1024 256; do for g in graphs eager; do for S in 1 2; do
  flag=""; [ "$g" = graphs ] && flag="--graphs"
  python bench.py --size $sz --ndisp 64 --streams $S $flag --steps 200 --warmup 10 --no-cpu 2>/dev/null | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('size $sz $g streams $S: %.4f ms/tile %.0f tiles/s' % (d['ms_per_step'], d['tiles_per_s']))"
done; done; done
