#!/bin/bash
# tools/depth_probe.sh -- round 6: the volume depth rounded up to 64 (shipped) against the packed multiple of 16 of rounds 1-5, by disparity count,
# for a batch of 8 tiles per launch, for single-tile calls in flight and for a tile alone.  (The third block of profiles/r06/depth_probe.txt, "k6alone", was a build
# with 12 candidates per lane at D = 192 also for a tile launched alone: adopted from 768 px on, see mgm_lane_layout.)
# Same results in every build (tests/test_gpu_batch.py, test_gpu_census.py run against the oracle, whose depth is the multiple of 16).  Probe builds:
#   tools/build_variants.sh d16 "-DS2P_CENSUS_DEPTH16"
cd "$(dirname "$0")/.."
OUT=gpurun_out/profiles/r06
mkdir -p $OUT
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('%6.1f G/s %.4f ms/tile (band launch %.3f, cost %.3f, wta %.3f)' % (d['value']/1e3, d['ms_per_tile'], d['roofline']['avg_launch_ms'], s['cost'], s['wta']))"; }
run() { python bench.py --steps 4 --warmup 2 --no-job --no-pool --no-cpu "$@" 2>/dev/null | line; }
{
for V in d16 shipped; do
  [ $V = shipped ] && unset S2P_HIP_LIB || export S2P_HIP_LIB=$PWD/build/variants/libs2p_hip_$V.so
  echo "== $V"
  for nd in 48 80 96 112 144 160 176 192 208 240; do
    echo "1024^2 x $nd | 8 per call x 3: $(run --size 1024 --ndisp $nd --batch 64) | 1 per call x 3: $(run --size 1024 --ndisp $nd --batch-launch 1 --batch 48) | alone: $(run --size 1024 --ndisp $nd --batch-launch 1 --batch 24 --streams 1)"
  done
  for nd in 112 176 192; do
    echo "512^2 x $nd  | 8 per call x 3: $(run --size 512 --ndisp $nd --batch 128) | 1 per call x 3: $(run --size 512 --ndisp $nd --batch-launch 1 --batch 96) | alone: $(run --size 512 --ndisp $nd --batch-launch 1 --batch 48 --streams 1)"
  done
  unset S2P_HIP_LIB
done
} 2>&1 | tee $OUT/depth_probe.txt
