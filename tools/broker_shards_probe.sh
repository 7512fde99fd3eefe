#!/bin/bash
# tools/broker_shards_probe.sh -- round 6: the Pool model with 1 / 2 / 3 GPU-owning broker processes per device
# (S2P_HIP_BROKER_PROCS: a worker talks to shard pid mod N; 3 lanes for the device shared out over the shards, or more with --lanes) at the
# headline tile size and at the sizes where one interpreter was the limiter (profiles/r05/broker_small_tiles.txt).
cd "$(dirname "$0")/.."
OUT=gpurun_out/profiles/r06
mkdir -p $OUT
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['pools'][-1]; b=p.get('broker',{}); s=p.get('steady') or {}
print('%s tiles/s steady, fork-join %s, %s tiles per library call, lanes busy %s, procs %s lanes %s, errors %d' % (s.get('tiles_per_s'), p.get('tiles_per_s_fork_to_join'), p.get('mean_tiles_per_library_call'), b.get('lane_busy_frac_of_wall'), b.get('procs'), b.get('lanes'), d['errors']))"; }
{
for rep in 1 2; do
  for shape in "1024 128 1536" "512 64 4096" "256 32 8192"; do
    set -- $shape
    for cfg in "1 3" "2 2" "3 1" "3 2" "4 1"; do
      pr=${cfg% *}; ln=${cfg#* }
      echo "$1 x $2, 64 workers, procs $pr x lanes $ln: $(python bench_pool.py --size $1 --ndisp $2 --workers 64 --tiles $3 --procs $pr --lanes $ln 2>/dev/null | line)"
    done
  done
done
echo "resident 512 x 64:  $(python bench.py --size 512 --ndisp 64 --steps 6 --warmup 2 --no-job --no-pool --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f tiles/s' % d['tiles_per_s'])")"
echo "resident 256 x 32:  $(python bench.py --size 256 --ndisp 32 --steps 6 --warmup 2 --no-job --no-pool --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f tiles/s' % d['tiles_per_s'])")"
} 2>&1 | tee $OUT/broker_shards_probe.txt
