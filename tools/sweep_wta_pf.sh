#!/bin/bash
# tools/sweep_wta_pf.sh -- rebuild libs2p_hip.so with different WTA prefetch depths and bench the census matcher.
set -e
cd "$(dirname "$0")/.."
SRC="s2p_amd/csrc/api.hip s2p_amd/csrc/sgbm_kernels.hip s2p_amd/csrc/census_kernels.hip s2p_amd/csrc/warp_kernels.hip s2p_amd/csrc/tri_kernels.hip s2p_amd/csrc/fusion_kernels.hip s2p_amd/csrc/raster_kernels.hip"
for PF in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-value -fvisibility=hidden -Iinclude -DS2P_WTA_PF=$PF -o s2p_amd/lib/libs2p_hip.so $SRC 2>/dev/null
  for st in 1 2; do
  python bench.py --algo census --streams $st --steps 40 --warmup 3 --no-cpu | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('WTA_PF=$PF streams=$st', d['ms_per_step'], 'agg', d['stage_ms']['aggregate'], 'wta', d['stage_ms']['wta'])"
  done
done
