#!/bin/bash
# tools/issue_bound_probe.sh -- round 6: is the band kernel bound by instruction issue per SIMD?  Four measurements on one box:
#   1. tools/probes/issue_rate: instructions per cycle a SIMD issues, by class, with 1 / 2 / 4 waves per SIMD, and whether the scalar
#      instructions of one wave ride beside the vector instructions of another;
#   2. the 8-tile band launch with 128 ... 512 workers (one worker = 4 compute waves + fetcher: 256 workers = ONE compute wave per SIMD);
#   3. timing probes (results invalid) without the step's flow-control tests (nopoll1) and without the progress word either (nopoll2):
#      the upper bound of what a cheaper protocol can buy;
#   4. rows confined to N CUs with the bands on a second, unmasked stream (completes tools/cumask_sweep.sh).
cd "$(dirname "$0")/.."
OUT=gpurun_out/profiles/r06
mkdir -p $OUT
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('%.4f ms per tile, %.1f G/s | band launch %.3f ms (%d tiles), cost %.3f wta %.3f' % (d['ms_per_step'] / d['config']['tiles_per_step'], d['value'] / 1e3, d['roofline']['avg_launch_ms'], d['roofline']['tiles_per_launch'], s['cost'], s['wta']))"; }
run() { python bench.py --steps 6 --warmup 2 --no-job --no-pool --no-cpu "$@" 2>/dev/null | line; }
{
echo "== 1. issue rates (tools/probes/issue_rate) =="
./tools/probes/issue_rate
echo "== 2. workers of the 8-tile launch, one call at a time (--streams 1) and the headline (3 calls in flight) =="
for rep in 1 2; do
  for w in 128 192 256 384 512 768; do
    echo "workers $w, 1 stream:  $(S2P_MGM_WORKERS=$w run --streams 1)"
  done
  for w in 256 512; do
    echo "workers $w, headline:  $(S2P_MGM_WORKERS=$w run)"
  done
done
echo "== 3. the step without flow control (timing probes, results invalid) =="
for rep in 1 2; do
  for V in shipped nopoll1 nopoll2; do
    [ $V = shipped ] && unset S2P_HIP_LIB || export S2P_HIP_LIB=$PWD/build/variants/libs2p_hip_$V.so
    echo "$V, 1 stream, 8 tiles per call:  $(run --streams 1)"
    echo "$V, 1 stream, 1 tile per call:   $(run --streams 1 --batch-launch 1 --batch 48)"
    echo "$V, headline:                    $(run)"
    unset S2P_HIP_LIB
  done
done
echo "== 4. rows confined to N CUs, bands on a second unmasked stream =="
for r in 64 128 192 224; do
  echo "band everywhere, rows on $r:  $(S2P_HIP_CU_ROWS=$r run)"
done
echo "unmasked: $(run)"
} 2>&1 | tee $OUT/issue_bound_probe.txt
