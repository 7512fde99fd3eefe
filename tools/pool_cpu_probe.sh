#!/bin/bash
# tools/pool_cpu_probe.sh -- round 6: what limits the Pool model below the headline tile size on THIS box?  The box's control group grants
# 16 CPUs' worth of time (cat /sys/fs/cgroup/cpu.max: 1600000 100000) to everything a run starts.  bench_pool.py now reports the CPU the
# whole Pool used (cgroup cpu.stat) beside its rate; the lane's library call without any broker is tools/host_batch_time.py.
cd "$(dirname "$0")/.."
OUT=gpurun_out/profiles/r06
mkdir -p $OUT
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['pools'][-1]; b=p.get('broker',{}); s=p.get('steady') or {}; c=p.get('cgroup_cpu',{})
print('%s tiles/s steady (fork-join %s); CPUs used %s of %s granted, %s ms of CPU per tile, throttled %s s in %s events; per call read %s / broker %s / write %s ms' % (s.get('tiles_per_s'), p.get('tiles_per_s_fork_to_join'), c.get('used_cpus'), c.get('quota_cpus'), c.get('cpu_ms_per_tile'), c.get('throttled_s'), c.get('throttle_events'), s.get('read_ms'), s.get('gpu_ms'), s.get('write_ms')))"; }
{
echo "nproc $(nproc); cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
for shape in "1024 128 1536" "512 64 4096" "256 32 8192"; do
  set -- $shape
  for P in 8 16 32 64; do
    echo "$1 x $2, $P workers: $(python bench_pool.py --size $1 --ndisp $2 --workers $P --tiles $3 2>/dev/null | line)"
  done
done
echo "== the lane's library call without the broker (tools/host_batch_time.py) =="
python tools/host_batch_time.py 1024 128 2>/dev/null
python tools/host_batch_time.py 512 64 2>/dev/null
python tools/host_batch_time.py 256 32 2>/dev/null
} 2>&1 | tee $OUT/pool_cpu_probe.txt
