#!/bin/bash
# tools/noc_valu_probe.sh -- VERDICT r04 item 3, measured before built: what could "aggregation without the C stream" (Hamming costs
# recomputed inside k_mgm_bands from the census signatures) gain NET of the instructions it adds?  S2P_MGM_PROBE_NO_C removes the cost
# loads (results invalid); the recomputation costs >= 16 more VALU instructions per step (8 xor + 8 popcount + 4 packs against today's
# 4 unpacks), which S2P_MGM_PROBE_VMOV=16 / 24 adds on top.  Headline shape (8 tiles per call, three streams), alternating, two rounds.
cd "$(dirname "$0")/.."
OUT=gpurun_out/profiles/r05
mkdir -p $OUT
for rep in 1 2; do for V in shipped noc noc_vmov16 noc_vmov24; do
  if [ $V = shipped ]; then unset S2P_HIP_LIB; else export S2P_HIP_LIB=$PWD/build/variants/libs2p_hip_$V.so; fi
  echo "$V: $(python bench.py --steps 10 --warmup 3 --no-job --no-pool --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms per tile, %.1f G/s, band launch %.3f ms (8 tiles)' % (d['ms_per_step'] / d['config']['tiles_per_step'], d['value'] / 1e3, d['roofline']['avg_launch_ms']))")"
done; done | tee $OUT/noc_valu_probe.txt
