#!/bin/bash
# tools/knob_probe.sh -- round 5: knobs of the in-flight headline that earlier rounds did not sweep TOGETHER: tiles per library call beyond 8
# (the launch alone went 0.26 / 0.37 / 0.46 / 0.50 of the roofline for 1 / 2 / 4 / 8 tiles, profiles/r03/batch_sweep.txt) x tile streams, and
# the wave priority of ALL band compute waves (round 3 only raised the axis lattices') beside the row kernels of the other calls in flight.
cd "$(dirname "$0")/.."
OUT=gpurun_out/profiles/r05
mkdir -p $OUT
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms per tile, %.1f G/s, band launch %.3f ms (%d tiles), frac %.3f' % (d['ms_per_step'] / d['config']['tiles_per_step'], d['value'] / 1e3, d['roofline']['avg_launch_ms'], d['roofline']['tiles_per_launch'], d['roofline']['frac']))"; }
{
for rep in 1 2; do
  for cfg in "8 3" "12 3" "16 2" "16 3" "24 2" "32 2"; do
    set -- $cfg
    echo "tiles/call $1 streams $2: $(python bench.py --batch-launch $1 --streams $2 --batch 384 --steps 8 --warmup 3 --no-job --no-pool --no-cpu 2>/dev/null | line)"
  done
  for V in prio2 prio3 fprio2; do
    export S2P_HIP_LIB=$PWD/build/variants/libs2p_hip_$V.so
    echo "$V (8 x 3): $(python bench.py --steps 10 --warmup 3 --no-job --no-pool --no-cpu 2>/dev/null | line)"
    unset S2P_HIP_LIB
  done
done
} 2>&1 | tee $OUT/knob_probe.txt
