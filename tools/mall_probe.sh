#!/bin/bash
# tools/mall_probe.sh -- GPU box: is the cost volume of the MGM mode served by the Infinity Cache?  The same kernels with the cost
# loads marked non-temporal (aux = 2: they stream past the cache), one tile in flight against three.  If, with three tiles in
# flight, the re-reads of the three cost volumes (402 MB) already miss the 256 MiB cache, taking them out of it costs little;
# with one tile in flight (134 MB: resident) it costs a lot.
cd "$(dirname "$0")/.."
for AUX in 0 2; do
  S2P_HIP_EXTRA_FLAGS="-DS2P_C_LOAD_AUX=$AUX" python -m s2p_amd.build --force > /dev/null 2>&1
  for REC in 2 0; do for S in 1 3; do
    echo "cost loads aux $AUX, recursion $REC, $S tile stream(s): $(python bench.py --no-cpu --no-job --recursion $REC --steps 3 --batch 96 --batch-launch 1 --streams $S 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms per tile, aggregation alone %.4f ms' % (d['ms_per_tile'], d['stage_ms']['aggregate']))")"
  done; done
done
python -m s2p_amd.build --force > /dev/null 2>&1
