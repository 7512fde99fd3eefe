#!/bin/bash
# tools/il4_inflight_probe.sh -- GPU box: the in-flight headline with the shipped library and with -DS2P_MGM_IL4_PROBE (build/variants/libs2p_hip_il4.so,
# tools/build_variants.sh il4 "-DS2P_MGM_IL4_PROBE"), alternating; profiles/r05/il4_inflight_probe.txt
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms per tile, %.1f G/s, band launch %.3f ms' % (d['ms_per_step'] / d['config']['tiles_per_step'], d['value'] / 1e3, d['roofline']['avg_launch_ms']))"; }
for rep in 1 2 3; do
  for V in shipped il4; do
    if [ $V = shipped ]; then unset S2P_HIP_LIB; else export S2P_HIP_LIB=$PWD/build/variants/libs2p_hip_$V.so; fi
    echo "$V: $(python bench.py --steps 10 --warmup 3 --no-job --no-pool --no-cpu 2>/dev/null | line)"
  done
done 2>&1 | tee gpurun_out/il4_inflight_probe.txt
