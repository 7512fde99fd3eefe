#!/bin/bash
# tools/flag_probe.sh "<flags 1>" "<flags 2>" ... -- GPU box: build variants of the library (S2P_HIP_EXTRA_FLAGS) and time the MGM headline
# with the chip full (8 tiles per launch, 1 and 2 streams) and for one tile alone
cd "$(dirname "$0")/.."
for FLAGS in "$@"; do
  S2P_HIP_EXTRA_FLAGS="$FLAGS" python -m s2p_amd.build --force > /dev/null 2>&1
  for CFG in "1 1" "8 1" "8 2"; do set -- $CFG
    echo "[$FLAGS] tiles/call $1 streams $2: $(python bench.py --no-cpu --no-job --steps 3 --batch 96 --streams $2 --batch-launch $1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms per tile | aggregate launch %.4f ms' % (d['ms_per_tile'], d['stage_ms']['aggregate']))")"
  done
done
python -m s2p_amd.build --force > /dev/null 2>&1
