#!/bin/bash
# tools/ring_probe.sh -- GPU box: LDS rings of 16 entries at D = 128 (one band per CU) against rings of 8, prefetch 16, chip full / one tile
cd "$(dirname "$0")/.."
for FLAGS in "-DS2P_MGM_PF=16" "-DS2P_MGM_PF=16 -DS2P_MGM_RING16_UPTO=64" "-DS2P_MGM_PF=32 -DS2P_MGM_RING16_UPTO=64"; do
  S2P_HIP_EXTRA_FLAGS="$FLAGS" python -m s2p_amd.build --force > /dev/null 2>&1
  for NB in 1 8; do for S in 1 2; do
    echo "[$FLAGS] tiles/call $NB streams $S: $(python bench.py --no-cpu --no-job --steps 3 --batch 96 --streams $S --batch-launch $NB 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms per tile | aggregate launch %.4f ms' % (d['ms_per_tile'], d['stage_ms']['aggregate']))")"
  done; done
done
python -m s2p_amd.build --force > /dev/null 2>&1
