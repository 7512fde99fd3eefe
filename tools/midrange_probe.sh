#!/bin/bash
# tools/midrange_probe.sh -- round 6: disparity counts strictly between 128 and 256 (what real tiles have: per-tile ranges from the matches;
# BASELINE configs[2] is 192).  The K = 4 lane layout pads such a range to 32 lanes per pixel: two rows per wave in the band kernel, two pixels per
# wave in the WTA, up to 44 % of the lanes idle; 16 candidates per lane keep a pixel on ONE DPP row (padded).
# The A/B part of profiles/r06/midrange_probe.txt was taken with a build that carried two environment switches for the two layouts
# (S2P_MGM_K8_MID = 0 old / 1 new, S2P_WTA_K8_MID = 0 old / 1 new); the decision is compiled in since (mgm_lane_layout(), the WTA dispatch in
# census_level_enqueue) and the switches are gone.  Run now, this script measures the shipped layouts on the same shapes ("== shipped").
cd "$(dirname "$0")/.."
OUT=gpurun_out/profiles/r06
mkdir -p $OUT
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('%.1f G/s, %.4f ms per tile | band launch %.3f ms (%d tiles), wta %.3f' % (d['value']/1e3, d['ms_per_tile'], d['roofline']['avg_launch_ms'], d['roofline']['tiles_per_launch'], s['wta']))"; }
run() { python bench.py --steps 4 --warmup 2 --no-job --no-pool --no-cpu "$@" 2>/dev/null | line; }
{
for cfg in shipped; do
  echo "== shipped: 16 candidates per lane in batches and in the WTA, 8 per lane for a tile launched alone"
  for nd in 144 160 176 192 208 224 240; do
    echo "1024^2 x $nd, 8 per call x 3 in flight: $(run --size 1024 --ndisp $nd --batch 64)"
  done
  echo "1024^2 x 192, 1 per call x 3 in flight: $(run --size 1024 --ndisp 192 --batch-launch 1 --batch 48)"
  echo "1024^2 x 192, 1 per call x 1:           $(run --size 1024 --ndisp 192 --batch-launch 1 --batch 24 --streams 1)"
  echo "512^2 x 192, 8 per call x 3 in flight:  $(run --size 512 --ndisp 192 --batch 128)"
  echo "512^2 x 192, 1 per call x 3 in flight:  $(run --size 512 --ndisp 192 --batch-launch 1 --batch 96)"
  echo "512^2 x 192, 1 per call x 1:            $(run --size 512 --ndisp 192 --batch-launch 1 --batch 48 --streams 1)"
  python tools/config2_time.py 2>/dev/null | grep "'mgm' "
done
} 2>&1 | tee $OUT/midrange_shipped.txt
