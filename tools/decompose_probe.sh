#!/bin/bash
# tools/decompose_probe.sh -- round 6: the 8-tile band launch decomposed (timing probes, results invalid): with / without its memory traffic
# (cost loads and e-stores issued out of range: same instructions, no bytes), with / without its flow control, at 1 / 2 / 3 bands per CU.
cd "$(dirname "$0")/.."
OUT=gpurun_out/profiles/r06
mkdir -p $OUT
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('band launch %.3f ms (%d tiles)' % (d['roofline']['avg_launch_ms'], d['roofline']['tiles_per_launch']))"; }
run() { python bench.py --steps 4 --warmup 2 --no-job --no-pool --no-cpu --streams 1 "$@" 2>/dev/null | line; }
{
for rep in 1 2; do
  for V in shipped nomem nopoll1 nopoll2 nomem_nopoll1 nomem_nopoll2; do
    [ $V = shipped ] && unset S2P_HIP_LIB || export S2P_HIP_LIB=$PWD/build/variants/libs2p_hip_$V.so
    for cfg in "2 256" "2 512" "3 768"; do
      set -- $cfg
      echo "$V, per CU $1, workers $2: $(S2P_MGM_PER_CU=$1 S2P_MGM_WORKERS=$2 run)"
    done
    echo "$V, one tile per launch: $(run --batch-launch 1 --batch 48)"
    unset S2P_HIP_LIB
  done
done
} 2>&1 | tee $OUT/decompose_probe.txt
