#!/bin/bash
# tools/k6_probe.sh -- round 6: 12 candidates per lane (K = 6, 16 lanes) for the disparity counts that are whole numbers of such lanes in the band kernel's
# batches, against 16 per lane (K = 8, padded to 256).  profiles/r06/k6_probe.txt was taken with builds that carried two switches for the A/B
# (-DS2P_MGM_K6=0: without the layout; -DS2P_MGM_NW_BATCH_K6=4: with 4-wave bands); the decision is compiled in since (mgm_lane_layout(): D = 192, in batches
# and on tiles from 768 px) and the switches are gone.  Run now, the script measures the shipped build on the same shapes.
cd "$(dirname "$0")/.."
OUT=gpurun_out/profiles/r06
mkdir -p $OUT
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('%.1f G/s, %.4f ms per tile | band launch %.3f ms (%d tiles), wta %.3f' % (d['value']/1e3, d['ms_per_tile'], d['roofline']['avg_launch_ms'], d['roofline']['tiles_per_launch'], s['wta']))"; }
run() { python bench.py --steps 4 --warmup 2 --no-job --no-pool --no-cpu "$@" 2>/dev/null | line; }
{
for V in shipped; do
  [ $V = shipped ] && unset S2P_HIP_LIB || export S2P_HIP_LIB=$PWD/build/variants/libs2p_hip_$V.so
  echo "== $V"
  for nd in 144 192; do
    echo "1024^2 x $nd, 8 per call x 3 in flight: $(run --size 1024 --ndisp $nd --batch 64)"
    echo "1024^2 x $nd, 8 per call x 1:           $(run --size 1024 --ndisp $nd --batch 32 --streams 1)"
    echo "512^2 x $nd, 8 per call x 3 in flight:  $(run --size 512 --ndisp $nd --batch 128)"
  done
  for nd in 48 96; do
    echo "1024^2 x $nd, 8 per call x 3 in flight: $(run --size 1024 --ndisp $nd --batch 64)"
    echo "512^2 x $nd, 8 per call x 3 in flight:  $(run --size 512 --ndisp $nd --batch 128)"
  done
  unset S2P_HIP_LIB
done
} 2>&1 | tee $OUT/k6_shipped.txt
