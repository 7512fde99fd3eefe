#!/bin/bash
# tools/direct_hang_probe.sh -- GPU box: direct mode (every worker drives the GPU) with 16 workers, repeated; a worker that is not back after
# 20 s prints its Python stack (faulthandler), the Pool gives up after 45 s
cd "$(dirname "$0")/.."
for i in 1 2 3 4 5 6; do
  echo "== run $i"
  S2P_POOL_FAULTHANDLER=20 timeout 120 python bench_pool.py --workers 16 --tiles 384 --broker 0 --task-timeout 45 2> gpurun_out/r04/direct_hang_$i.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['pools'][0]
print(p.get('error') or ('ok: steady %s tiles/s, cold max %s' % ((p.get('steady') or {}).get('tiles_per_s'), p['cold_start_s']['max'])))"
  grep -c "Timeout" gpurun_out/r04/direct_hang_$i.err
  grep -A14 "Timeout (0:00:20)" gpurun_out/r04/direct_hang_$i.err | head -40
done
