#!/bin/bash
# tools/il4_probe.sh -- GPU box: what would the MGM kernel gain if the axis lattices' volumes had 4 image rows interleaved per pixel
# column?  A timing-only build (-DS2P_MGM_IL4_PROBE: the axis lattices ADDRESS their cost and e-volumes that way; results invalid).
cd "$(dirname "$0")/.."
for FLAG in "" "-DS2P_MGM_IL4_PROBE"; do
  S2P_HIP_EXTRA_FLAGS="$FLAG" python -m s2p_amd.build --force > /dev/null 2>&1
  for NB in 1 8; do for S in 1 2; do
    echo "[$FLAG] tiles/call $NB streams $S: $(python bench.py --no-cpu --no-job --steps 3 --batch 96 --streams $S --batch-launch $NB 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms per tile | aggregate launch %.4f ms' % (d['ms_per_tile'], d['stage_ms']['aggregate']))")"
  done; done
done
python -m s2p_amd.build --force > /dev/null 2>&1
