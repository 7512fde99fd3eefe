#!/bin/bash
# the consensus WTA's byte keys in the regime that counts: tiles in flight on three streams (the bench headline), where the kernels share the SIMDs
cd "$(dirname "$0")/.."
for rep in 1 2 3; do for K in 1 0; do
  echo "byte keys $K: $(S2P_WTA_BYTE_KEYS=$K python bench.py --steps 10 --warmup 3 --no-job --no-pool --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms per tile, %.1f G/s, roofline frac %s' % (d['ms_per_step'] / d['config']['tiles_per_step'], d['value'] / 1e3, d['roofline']['frac']))")"
done; done
