#!/bin/bash
# tools/cpol_small_probe.sh -- round 6: the e-store cache policy BELOW 128 candidates, where a pixel's e-bytes of one direction are a
# half, a quarter or an eighth of a 128-byte line and every store of the band kernel is a partial line by construction
# (profiles/r06/smalld_probe.txt: the launch at D = 16 / 32 is bound by these stores).  Shipped: nt (S2P_E_STORE_AUX = 2), chosen at D = 128.
# Does a policy that lets the L2 merge neighbouring pixels' stores before the line leaves help?  Probe builds:
#   tools/build_variants.sh est0 "-DS2P_E_STORE_AUX=0" est1 "-DS2P_E_STORE_AUX=1" est16 "-DS2P_E_STORE_AUX=16" est17 "-DS2P_E_STORE_AUX=17" est3 "-DS2P_E_STORE_AUX=3"
cd "$(dirname "$0")/.."
OUT=gpurun_out/profiles/r06
mkdir -p $OUT
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('%.1f G/s, %.4f ms per tile | band launch %.3f ms (%d tiles), wta %.3f' % (d['value']/1e3, d['ms_per_tile'], d['roofline']['avg_launch_ms'], d['roofline']['tiles_per_launch'], s['wta']))"; }
run() { python bench.py --steps 4 --warmup 2 --no-job --no-pool --no-cpu "$@" 2>/dev/null | line; }
{
for V in ${VARIANTS:-shipped est0 est1 est3 est16 est17}; do
  [ $V = shipped ] && unset S2P_HIP_LIB || export S2P_HIP_LIB=$PWD/build/variants/libs2p_hip_$V.so
  echo "== $V"
  if [ -n "$SHAPES" ]; then      # the other call shapes (ent = -DS2P_MGM_E_PLAIN_UPTO=0: non-temporal everywhere, the rule until this change)
    for sz in 1024 512; do for nd in 16 32; do
      echo "${sz}^2 x $nd | 8 per call x 3: $(run --size $sz --ndisp $nd --batch 64) | 1 per call x 3: $(run --size $sz --ndisp $nd --batch-launch 1 --batch 48) | alone: $(run --size $sz --ndisp $nd --batch-launch 1 --batch 24 --streams 1)"
    done; done
  else
  for nd in 16 32 48 64 128; do
    echo "1024^2 x $nd, 8 per call x 3 in flight: $(run --size 1024 --ndisp $nd --batch 64)"
  done
  fi
  unset S2P_HIP_LIB
done
} 2>&1 | tee $OUT/${NAME:-cpol_small_probe}.txt
