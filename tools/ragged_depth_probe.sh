#!/bin/bash
# tools/ragged_depth_probe.sh -- round 6: the reference's execution model (fork Pool(64) x compute_disparity_map('mgm') on TIFFs in /dev/shm, through the
# broker) on 64 tile shapes whose disparity ranges differ in LENGTH as a real job's do (48 ... 223 candidates), with the volumes' depth rounded up to 64
# (shipped) and packed to the multiple of 16 (rounds 1-5).  Probe build: tools/build_variants.sh d16 "-DS2P_CENSUS_DEPTH16"
cd "$(dirname "$0")/.."
OUT=gpurun_out/profiles/r06
mkdir -p $OUT
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['pools'][-1]; b=d['best']
print('%d workers: %.0f tiles/s steady (%.0f fork -> join), %.1f G disparities/s over a mean of %.1f M candidates per tile; tiles per library call %.2f, lane busy %.2f, CPUs used %.1f of %.0f' % (
  p['workers'], b['steady_tiles_per_s'], b['fork_to_join_tiles_per_s'], b['Mdisp_per_s'] / 1e3, b['mean_candidates_per_tile'] / 1e6, p['mean_tiles_per_library_call'], p['broker']['lane_busy_frac_of_wall'], p['cgroup_cpu']['used_cpus'], p['cgroup_cpu']['quota_cpus']))"; }
{
for rep in 1 2; do
for V in d16 shipped; do
  [ $V = shipped ] && unset S2P_HIP_LIB || export S2P_HIP_LIB=$PWD/build/variants/libs2p_hip_$V.so
  echo "$V, 1024^2: $(python bench_pool.py --workers 64 --ragged-depth --distinct 64 2>/dev/null | line)"
  echo "$V, 768^2:  $(python bench_pool.py --workers 64 --ragged-depth --distinct 64 --size 768 2>/dev/null | line)"
  unset S2P_HIP_LIB
done
done
} 2>&1 | tee $OUT/ragged_depth_probe.txt
