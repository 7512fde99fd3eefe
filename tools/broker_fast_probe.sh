#!/bin/bash
# tools/broker_fast_probe.sh -- round 6: the Pool model after the broker's per-request path was shortened (one recv per request, grouping
# fields computed once per request, pre-encoded replies, parameter structs cached, two transfers per tile instead of five), with 1 and 3
# broker processes per device, at the headline tile size and at the smaller ones of profiles/r05/broker_small_tiles.txt.
cd "$(dirname "$0")/.."
OUT=gpurun_out/profiles/r06
mkdir -p $OUT
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['pools'][-1]; b=p.get('broker',{}); s=p.get('steady') or {}
print('%s tiles/s steady, fork-join %s, %s tiles per call, lanes busy %s; per call: read %s / broker %s / write %s ms, queue %s ms; errors %d' % (s.get('tiles_per_s'), p.get('tiles_per_s_fork_to_join'), p.get('mean_tiles_per_library_call'), b.get('lane_busy_frac_of_wall'), s.get('read_ms'), s.get('gpu_ms'), s.get('write_ms'), b.get('queue_ms_per_request'), d['errors']))"; }
{
for rep in 1 2; do
  for shape in "1024 128 1536" "512 64 4096" "256 32 8192"; do
    set -- $shape
    for cfg in "1 3" "3 1" "1 4"; do
      pr=${cfg% *}; ln=${cfg#* }
      echo "$1 x $2, 64 workers, procs $pr x lanes $ln: $(python bench_pool.py --size $1 --ndisp $2 --workers 64 --tiles $3 --procs $pr --lanes $ln 2>/dev/null | line)"
    done
  done
done
python bench_pool.py --size 1024 --ndisp 128 --workers 64 --tiles 1536 --ragged --distinct 64 --verify 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ragged 64 shapes, verify:', d.get('verify'), (d['pools'][-1].get('steady') or {}).get('tiles_per_s'), 'tiles/s, errors', d['errors'])"
} 2>&1 | tee $OUT/broker_fast_probe.txt
