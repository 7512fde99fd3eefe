#!/bin/bash
# tools/cumask_sweep.sh -- round 6: take the row kernels off the band kernel's CUs.
# The library's contexts read S2P_HIP_CU_BAND (CUs of the band-pipelined MGM launches: the n lowest mask bits, on a second stream) and
# S2P_HIP_CU_ROWS (CUs of the cost / WTA / median / epilogue kernels: the m highest mask bits); unset = the whole device.  Mask bits are
# dealt round-robin over the 8 XCDs (tools/probes/cumask_map prints what a mask selects), so multiples of 8 are XCD-balanced.
# Headline command (8 tiles per call, 3 calls in flight), alternating with the unmasked build on ONE box.
cd "$(dirname "$0")/.."
OUT=gpurun_out/profiles/r06
mkdir -p $OUT
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('%.4f ms per tile, %.1f G/s | alone on one stream: band launch %.3f ms, cost %.3f wta %.3f median %.3f, 1 stream %.4f ms per tile' % (d['ms_per_step'] / d['config']['tiles_per_step'], d['value'] / 1e3, d['roofline']['avg_launch_ms'], s['cost'], s['wta'], s['median'], d.get('ms_per_tile_1_stream') or 0))"; }
run() { python bench.py --steps 10 --warmup 3 --no-job --no-pool --no-cpu 2>/dev/null | line; }
{
echo "== what a mask selects (tools/probes/cumask_map) =="
./tools/probes/cumask_map
echo
echo "== headline command, 1024^2 x 128, 8 tiles per call x 3 calls in flight =="
for rep in 1 2; do
  echo "-- repetition $rep"
  echo "unmasked (shipped):                 $(run)"
  for cfg in "224 32" "192 64" "160 96" "128 128"; do
    set -- $cfg
    echo "band $1 : rows $2 (disjoint):        $(S2P_HIP_CU_BAND=$1 S2P_HIP_CU_ROWS=$2 run)"
    echo "band $1 : rows $2, workers 2/CU:     $(S2P_HIP_CU_BAND=$1 S2P_HIP_CU_ROWS=$2 S2P_MGM_WORKERS=$((2 * $1)) run)"
  done
  for r in 32 64 96 128 192; do
    echo "band everywhere, rows on $r:         $(S2P_HIP_CU_ROWS=$r run)"
  done
  for b in 224 192 160; do
    echo "band on $b, rows everywhere:        $(S2P_HIP_CU_BAND=$b run)"
  done
  echo "band 256 on its own stream (no mask effect, the fork/join events only): $(S2P_HIP_CU_BAND=256 run)"
  echo "unmasked (shipped):                 $(run)"
done
} 2>&1 | tee $OUT/cumask_sweep.txt
