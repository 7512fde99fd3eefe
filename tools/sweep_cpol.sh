#!/bin/bash
# tools/sweep_cpol.sh -- rebuild with different cache policies for the e-volume stores (aggregation) and loads (WTA)
# and bench both matchers.  Arguments: "<store_aux>:<load_aux>" pairs (CPol bits: 1 sc0, 2 nt, 16 sc1).
cd "$(dirname "$0")/.."
SRC="s2p_amd/csrc/api.hip s2p_amd/csrc/sgbm_kernels.hip s2p_amd/csrc/census_kernels.hip s2p_amd/csrc/warp_kernels.hip s2p_amd/csrc/tri_kernels.hip s2p_amd/csrc/fusion_kernels.hip s2p_amd/csrc/raster_kernels.hip"
for P in "$@"; do
  ST=${P%%:*}; LD=${P##*:}
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-value -fvisibility=hidden -Iinclude -DS2P_E_STORE_AUX=$ST -DS2P_E_LOAD_AUX=$LD -o s2p_amd/lib/libs2p_hip.so $SRC 2>/dev/null || { echo "build failed for $P"; continue; }
  for algo in census sgbm; do
    python bench.py --algo $algo --streams 1 --steps 40 --warmup 3 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('store=$ST load=$LD', '$algo', d['ms_per_step'], 'cost', d['stage_ms']['cost'], 'agg', d['stage_ms']['aggregate'], 'wta', d['stage_ms']['wta'])"
  done
done
