#!/bin/bash
# tools/pitch_probe.sh -- round 6: is it the PIXEL PITCH that makes the disparity counts off the multiples of 64 slow (8 tiles of 1024^2 per launch:
# D = 112 6.4 ms against 4.1 at D = 128)?  Timing probes (results invalid): the launch of D = 128 / 192 / 256 with only the lanes of the first V candidates
# loading and storing -- the bytes of a range of V at a pitch of 128 / 192 / 256.  Probe builds:
#   tools/build_variants.sh dv96 "-DS2P_MGM_PROBE_DVALID=96" dv112 "-DS2P_MGM_PROBE_DVALID=112" dv144 "-DS2P_MGM_PROBE_DVALID=144" dv160 "-DS2P_MGM_PROBE_DVALID=160" dv48 "-DS2P_MGM_PROBE_DVALID=48" \
#     dv96s "-DS2P_MGM_PROBE_DVALID=96 -DS2P_MGM_PROBE_FULLSTORE" (112, 144, 160 alike) nost "-DS2P_MGM_PROBE_NOMEM=2" nold "-DS2P_MGM_PROBE_NOMEM=1"
cd "$(dirname "$0")/.."
OUT=gpurun_out/profiles/r06
mkdir -p $OUT
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('band launch %.3f ms (%d tiles)' % (d['roofline']['avg_launch_ms'], d['roofline']['tiles_per_launch']))"; }
run() { python bench.py --steps 4 --warmup 2 --no-job --no-pool --no-cpu --size 1024 --batch 64 "$@" 2>/dev/null | line; }
{
echo "packed (shipped): D = 48 $(run --ndisp 48) | 64 $(run --ndisp 64) | 96 $(run --ndisp 96) | 112 $(run --ndisp 112) | 128 $(run --ndisp 128) | 144 $(run --ndisp 144) | 160 $(run --ndisp 160) | 192 $(run --ndisp 192)"
for v in 48 96 112; do
  export S2P_HIP_LIB=$PWD/build/variants/libs2p_hip_dv$v.so
  [ $v = 48 ] && echo "V = $v at pitch 64:  $(run --ndisp 64)"
  echo "V = $v at pitch 128: $(run --ndisp 128)"
done
for v in 144 160; do
  export S2P_HIP_LIB=$PWD/build/variants/libs2p_hip_dv$v.so
  echo "V = $v at pitch 192: $(run --ndisp 192)"
  echo "V = $v at pitch 256: $(run --ndisp 256)"
done
echo "-- the same, but EVERY lane of the pitch stores (whole lines written; loads still only from the valid lanes):"
for v in 96 112; do
  export S2P_HIP_LIB=$PWD/build/variants/libs2p_hip_dv${v}s.so
  echo "V = $v at pitch 128, whole-line stores: $(run --ndisp 128)"
done
for v in 144 160; do
  export S2P_HIP_LIB=$PWD/build/variants/libs2p_hip_dv${v}s.so
  echo "V = $v at pitch 192, whole-line stores: $(run --ndisp 192)"
  echo "V = $v at pitch 256, whole-line stores: $(run --ndisp 256)"
done
export S2P_HIP_LIB=$PWD/build/variants/libs2p_hip_nost.so
echo "-- packed, no e-stores at all (S2P_MGM_PROBE_NOMEM=2): D = 96 $(run --ndisp 96) | 112 $(run --ndisp 112) | 128 $(run --ndisp 128) | 144 $(run --ndisp 144)"
export S2P_HIP_LIB=$PWD/build/variants/libs2p_hip_nold.so
echo "-- packed, no cost loads (S2P_MGM_PROBE_NOMEM=1):     D = 96 $(run --ndisp 96) | 112 $(run --ndisp 112) | 128 $(run --ndisp 128) | 144 $(run --ndisp 144)"
} 2>&1 | tee $OUT/pitch_probe.txt
