#!/bin/bash
# tools/smalld_probe.sh -- round 6: what bounds the 8-tile band launch at SMALL disparity counts (D = 16 ... 64: 2 ... 8 lanes per pixel, a pixel's candidates
# a quarter / half of a 128-byte line)?  Timing builds (results invalid): without the e-stores, without the cost loads, without both, without flow control.
#   tools/build_variants.sh nost "-DS2P_MGM_PROBE_NOMEM=2" nold "-DS2P_MGM_PROBE_NOMEM=1" nomem "-DS2P_MGM_PROBE_NOMEM=3" nopoll "-DS2P_MGM_PROBE_NOPOLL=1"
cd "$(dirname "$0")/.."
OUT=gpurun_out/profiles/r06
mkdir -p $OUT
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f' % d['roofline']['avg_launch_ms'])"; }
run() { python bench.py --steps 4 --warmup 2 --no-job --no-pool --no-cpu --size 1024 --batch 64 "$@" 2>/dev/null | line; }
{
echo "band launch, 8 tiles of 1024^2, ms (memory floor at the D = 128 launch's byte rate: D / 128 x 4.13)"
echo "build      D=16    D=32    D=48    D=64    D=128"
for V in shipped nost nold nomem nopoll; do
  [ $V = shipped ] && unset S2P_HIP_LIB || export S2P_HIP_LIB=$PWD/build/variants/libs2p_hip_$V.so
  echo "$V  $(run --ndisp 16)  $(run --ndisp 32)  $(run --ndisp 48)  $(run --ndisp 64)  $(run --ndisp 128)"
  unset S2P_HIP_LIB
done
echo "1 tile per launch x 3 in flight / alone, ms per tile (shipped):"
for nd in 16 32 64; do
  echo "D=$nd: $(python bench.py --steps 4 --warmup 2 --no-job --no-pool --no-cpu --size 1024 --ndisp $nd --batch 64 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('8 per call x 3: %.4f ms/tile %.1f G/s' % (d['ms_per_tile'], d['value']/1e3), {k: round(v,3) for k,v in d['stage_ms'].items() if v})")"
done
} 2>&1 | tee $OUT/smalld_probe.txt
