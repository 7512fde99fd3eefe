#!/bin/bash
# tools/pf_probe.sh -- GPU box: cost-prefetch depth of k_mgm_bands (S2P_MGM_PF steps ahead) with the chip full (8 tiles per launch) and
# for one tile alone: is the saturated kernel bound by memory latency x bytes in flight?
cd "$(dirname "$0")/.."
for PF in ${PFS:-8 16 24 32}; do
  S2P_HIP_EXTRA_FLAGS="-DS2P_MGM_PF=$PF" python -m s2p_amd.build --force > /dev/null 2>&1
  for NB in 1 8; do for S in 1 2; do
    echo "prefetch $PF steps, tiles/call $NB streams $S: $(python bench.py --no-cpu --no-job --steps 3 --batch 96 --streams $S --batch-launch $NB 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms per tile | aggregate launch %.4f ms' % (d['ms_per_tile'], d['stage_ms']['aggregate']))")"
  done; done
done
python -m s2p_amd.build --force > /dev/null 2>&1
