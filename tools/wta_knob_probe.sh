#!/bin/bash
# tools/wta_knob_probe.sh -- GPU box: the consensus WTA's pipeline depth (pixel groups in flight per wave) and block size, re-swept under the
# round-5 pipeline (8 tiles per call, three calls in flight): probe builds from tools/build_variants.sh wpf1 "-DS2P_WTA_PF=1" wpf3 "-DS2P_WTA_PF=3"
# wnt128 "-DS2P_WTA_NT=128" wnt512 "-DS2P_WTA_NT=512", alternating with the shipped library (PF 2, 256 threads).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms per tile, %.1f G/s, wta %.3f ms' % (d['ms_per_step'] / d['config']['tiles_per_step'], d['value'] / 1e3, d['stage_ms']['wta']))"; }
for rep in 1 2; do
  for V in shipped wpf1 wpf3 wnt128 wnt512; do
    if [ $V = shipped ]; then unset S2P_HIP_LIB; else export S2P_HIP_LIB=$PWD/build/variants/libs2p_hip_$V.so; fi
    echo "$V: $(python bench.py --steps 10 --warmup 3 --no-job --no-pool --no-cpu 2>/dev/null | line)"
  done
done 2>&1 | tee gpurun_out/wta_knob_probe.txt
