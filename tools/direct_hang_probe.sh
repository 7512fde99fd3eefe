#!/bin/bash
# tools/direct_hang_probe.sh [runs] -- GPU box: direct mode (every worker drives the GPU itself) with 16 workers and the process fence
# LIFTED (S2P_HIP_MAX_PROCS_PER_DEVICE=0), repeated: does a Pool still "lose a worker" now that a worker's HipError can cross the
# process boundary (round 5: HipError.__reduce__)?  Prints per run: ok + rates, or the exception r.get() raised -- a HipError names what
# the library saw; a TimeoutError would be a worker that really did not come back (its Python stack is dumped after 20 s).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/profiles/r05
N=${1:-4}
for i in $(seq 1 $N); do
  echo "== run $i"
  S2P_HIP_MAX_PROCS_PER_DEVICE=0 S2P_POOL_FAULTHANDLER=20 timeout 150 python bench_pool.py --workers 16 --tiles 384 --broker 0 --task-timeout 45 \
      2> gpurun_out/profiles/r05/direct16_$i.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['pools'][0]
print(p.get('error') or ('ok: steady %s tiles/s, gpu ms per call %s, cold max %s' % ((p.get('steady') or {}).get('tiles_per_s'), (p.get('steady') or {}).get('gpu_ms'), p['cold_start_s']['max'])))"
  grep -c "Timeout (0:00:20)" gpurun_out/profiles/r05/direct16_$i.err
  grep -A14 "Timeout (0:00:20)" gpurun_out/profiles/r05/direct16_$i.err | head -30
done 2>&1 | tee gpurun_out/profiles/r05/direct16_fence_lifted.txt
