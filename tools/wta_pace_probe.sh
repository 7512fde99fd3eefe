#!/bin/bash
# (record of one measurement: the S2P_WTA_LDS_PAD switch was removed from the library after this run)
# does the in-flight headline gain when fewer WTA row blocks fit a CU beside the band kernel's workers?  (S2P_WTA_LDS_PAD: unused dynamic LDS)
cd "$(dirname "$0")/.."
for rep in 1 2; do for PAD in 0 14000 30000 50000; do
  echo "pad $PAD: $(S2P_WTA_LDS_PAD=$PAD python bench.py --steps 10 --warmup 3 --no-job --no-pool --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms per tile, %.1f G/s' % (d['ms_per_step'] / d['config']['tiles_per_step'], d['value'] / 1e3))")"
done; done
