#!/bin/bash
# tools/small_tiles_probe.sh -- round 6: resident tiles SMALLER than the headline with more of them per library call.  A launch of 8 small
# tiles is its (W + H)-step dependency chain; more tiles under one queue amortise it.  (--batch = tiles per step is raised with the
# tiles per call so that a step stays several calls per stream.)
cd "$(dirname "$0")/.."
OUT=gpurun_out/profiles/r06
mkdir -p $OUT
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; n=d['roofline']['tiles_per_launch']; print('%.1f us per tile, %.0f tiles/s, %.1f G/s | per call: band launch %.3f ms, per tile cost %.4f wta %.4f median %.4f, total %.3f ms' % (1e3 * d['ms_per_step'] / d['config']['tiles_per_step'], d['tiles_per_s'], d['value'] / 1e3, d['roofline']['avg_launch_ms'], s['cost'], s['wta'], s['median'], s['total']))"; }
run() { python bench.py --steps 6 --warmup 2 --no-job --no-pool --no-cpu "$@" 2>/dev/null | line; }
{
for shape in "256 32" "512 64" "768 96"; do
  set -- $shape
  for nb in 8 16 32 64; do
    for st in 1 3; do
      echo "$1 x $1 x $2, $nb tiles per call, $st stream(s): $(run --size $1 --ndisp $2 --batch-launch $nb --streams $st --batch $((nb * 24)))"
    done
  done
done
} 2>&1 | tee $OUT/small_tiles_probe.txt
