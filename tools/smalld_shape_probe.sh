#!/bin/bash
# tools/smalld_shape_probe.sh -- round 6: with plain e-stores at D <= 32 the band launch is no longer store-bound there (cpol_small_probe.txt):
# ring length, waves per band and bands per CU re-measured below 128 candidates.  Variants r8 / nw2 / nw8 are builds of a locally patched
# mgm_bands.hpp (S2P_MGM_RING16_UPTO 0: rings of 8 everywhere; S2P_MGM_NW_NARROW 2 / 8: waves per band for G < 16), S2P_MGM_PER_CU is the
# run-time probe of the shipped library.
cd "$(dirname "$0")/.."
OUT=gpurun_out/profiles/r06
mkdir -p $OUT
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('%.1f G/s, %.4f ms per tile | band launch %.3f ms (%d tiles), wta %.3f' % (d['value']/1e3, d['ms_per_tile'], d['roofline']['avg_launch_ms'], d['roofline']['tiles_per_launch'], s['wta']))"; }
run() { python bench.py --steps 4 --warmup 2 --no-job --no-pool --no-cpu "$@" 2>/dev/null | line; }
{
for V in ${VARIANTS:-shipped r8 nw2 nw8 percu1 percu3 percu4}; do
  unset S2P_HIP_LIB S2P_MGM_PER_CU
  case $V in
    shipped) ;;
    percu*) export S2P_MGM_PER_CU=${V#percu} ;;
    *) export S2P_HIP_LIB=$PWD/build/variants/libs2p_hip_$V.so ;;
  esac
  echo "== $V"
  for nd in ${NDISP:-16 32 64}; do
    echo "1024^2 x $nd, 8 per call x 3 in flight: $(run --size 1024 --ndisp $nd --batch 64)"
  done
done
} 2>&1 | tee $OUT/${NAME:-smalld_shape_probe}.txt
