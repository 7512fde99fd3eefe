#!/bin/bash
# tools/cpol_probe.sh -- round 6: cache-policy bits of the band kernel's cost loads (S2P_C_LOAD_AUX: shipped 0 = cached; 2 = nt, 1 = sc0, 16 = sc1)
# and e-stores (S2P_E_STORE_AUX: shipped 2 = nt; 0 = default write-back, 16 = sc1, 17 = sc0 sc1 write-through, 18 = nt sc1) for the 8-tile launch
# and the in-flight headline.  Round 1 swept them for the 8-path kernel only (one tile, C resident in the Infinity Cache).  Probe builds:
#   tools/build_variants.sh est0 "-DS2P_E_STORE_AUX=0" est16 "-DS2P_E_STORE_AUX=16" est17 "-DS2P_E_STORE_AUX=17" est18 "-DS2P_E_STORE_AUX=18" \
#                           cld2 "-DS2P_C_LOAD_AUX=2" cld16 "-DS2P_C_LOAD_AUX=16" cld1 "-DS2P_C_LOAD_AUX=1"
cd "$(dirname "$0")/.."
OUT=gpurun_out/profiles/r06
mkdir -p $OUT
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('%.4f ms per tile, %.1f G/s | band launch %.3f ms, wta %.3f' % (d['ms_per_step'] / d['config']['tiles_per_step'], d['value'] / 1e3, d['roofline']['avg_launch_ms'], s['wta']))"; }
run() { python bench.py --steps 6 --warmup 2 --no-job --no-pool --no-cpu "$@" 2>/dev/null | line; }
{
for rep in 1 2; do
  for V in shipped est0 est16 est17 est18 cld2 cld16 cld1; do
    [ $V = shipped ] && unset S2P_HIP_LIB || export S2P_HIP_LIB=$PWD/build/variants/libs2p_hip_$V.so
    echo "$V: 1 stream $(run --streams 1) | headline $(run)"
    unset S2P_HIP_LIB
  done
done
} 2>&1 | tee $OUT/cpol_probe.txt
