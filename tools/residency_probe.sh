#!/bin/bash
# tools/residency_probe.sh -- round 6: bands per CU of the BATCHED launch.  Round 2 found "fewer bands per CU beat full residency" on ONE tile per
# launch (a latency chain); the 8-tile launch is a throughput problem: tools/issue_bound_probe.sh shows it scaling with its workers (256 -> 512:
# 6.63 -> 4.13 ms) and the SIMDs issuing ~1 instruction per 4 cycles where tools/probes/issue_rate reaches 2-3.  S2P_MGM_PER_CU caps the
# resident bands per CU through LDS padding (0 = no cap), S2P_MGM_WORKERS the workgroups of the launch.
cd "$(dirname "$0")/.."
OUT=gpurun_out/profiles/r06
mkdir -p $OUT
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('%.4f ms per tile, %.1f G/s | band launch %.3f ms (%d tiles), cost %.3f wta %.3f' % (d['ms_per_step'] / d['config']['tiles_per_step'], d['value'] / 1e3, d['roofline']['avg_launch_ms'], d['roofline']['tiles_per_launch'], s['cost'], s['wta']))"; }
run() { python bench.py --steps 6 --warmup 2 --no-job --no-pool --no-cpu "$@" 2>/dev/null | line; }
{
for rep in 1 2; do
  for cfg in "2 512" "3 768" "4 1024" "0 768" "0 1024" "3 640"; do
    set -- $cfg
    echo "per CU $1, workers $2, 1 stream:  $(S2P_MGM_PER_CU=$1 S2P_MGM_WORKERS=$2 run --streams 1)"
    echo "per CU $1, workers $2, headline:  $(S2P_MGM_PER_CU=$1 S2P_MGM_WORKERS=$2 run)"
  done
  echo "per CU 3, workers 768, 16 tiles per call, 2 streams: $(S2P_MGM_PER_CU=3 S2P_MGM_WORKERS=768 run --batch-launch 16 --streams 2 --batch 384)"
  echo "per CU 3, workers 768, 1 tile per call, 3 streams:  $(S2P_MGM_PER_CU=3 S2P_MGM_WORKERS=768 run --batch-launch 1 --batch 96)"
  echo "shipped, 1 tile per call, 3 streams:                $(run --batch-launch 1 --batch 96)"
done
} 2>&1 | tee $OUT/residency_probe.txt
