#!/bin/bash
# tools/cpol_mid_probe.sh -- round 6: the e-store cache policy at disparity counts that are NOT multiples of 128 (a pixel's 8 x D e-bytes then
# start and end inside 128-byte lines: partial-line writes unless the L2 merges the neighbouring pixels' stores before it evicts the line).
# Shipped: nt (S2P_E_STORE_AUX = 2), chosen at D = 128 where every store of a DPP row is a whole line.  Probe builds:
#   tools/build_variants.sh est0 "-DS2P_E_STORE_AUX=0" est16 "-DS2P_E_STORE_AUX=16" cld2 "-DS2P_C_LOAD_AUX=2"
cd "$(dirname "$0")/.."
OUT=gpurun_out/profiles/r06
mkdir -p $OUT
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('%.1f G/s, %.4f ms per tile | band launch %.3f ms (%d tiles), wta %.3f' % (d['value']/1e3, d['ms_per_tile'], d['roofline']['avg_launch_ms'], d['roofline']['tiles_per_launch'], s['wta']))"; }
run() { python bench.py --steps 4 --warmup 2 --no-job --no-pool --no-cpu "$@" 2>/dev/null | line; }
{
for V in shipped est0 est16 cld2; do
  [ $V = shipped ] && unset S2P_HIP_LIB || export S2P_HIP_LIB=$PWD/build/variants/libs2p_hip_$V.so
  echo "== $V"
  for nd in 64 96 112 128 144 192 256; do
    echo "1024^2 x $nd, 8 per call x 3 in flight: $(run --size 1024 --ndisp $nd --batch 64)"
  done
  unset S2P_HIP_LIB
done
} 2>&1 | tee $OUT/cpol_mid_probe.txt
