#!/bin/bash
# tools/sweep_pf.sh -- rebuild libs2p_hip.so with different aggregation prefetch depths and bench both matchers.
set -e
cd "$(dirname "$0")/.."
SRC="s2p_amd/csrc/api.hip s2p_amd/csrc/sgbm_kernels.hip s2p_amd/csrc/census_kernels.hip s2p_amd/csrc/warp_kernels.hip s2p_amd/csrc/tri_kernels.hip s2p_amd/csrc/fusion_kernels.hip s2p_amd/csrc/raster_kernels.hip"
for PF in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-value -DS2P_AGG_PF=$PF -o s2p_amd/lib/libs2p_hip.so $SRC
  for algo in census sgbm; do
    python bench.py --algo $algo --steps 20 --warmup 3 --no-cpu | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('PF=$PF', '$algo', d['ms_per_step'], 'agg', d['stage_ms']['aggregate'], 'frac', d['roofline']['frac'])"
  done
done
