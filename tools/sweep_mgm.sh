#!/bin/bash
# tools/sweep_mgm.sh build|run  -- A/B of the band-pipelined MGM kernel's compile-time knobs.
#   build: (here, no GPU) compile one libs2p_hip.so per variant into build/variants/<name>/  (build/ travels with gpurun)
#   run:   (GPU box) bench every variant in MGM mode, one line each
# Variants: "<name> <extra hipcc flags>" below.
set -e
cd "$(dirname "$0")/.."
SRC="s2p_amd/csrc/api.hip s2p_amd/csrc/sgbm_kernels.hip s2p_amd/csrc/census_kernels.hip s2p_amd/csrc/warp_kernels.hip s2p_amd/csrc/tri_kernels.hip s2p_amd/csrc/fusion_kernels.hip s2p_amd/csrc/raster_kernels.hip"
VARIANTS=(
  "base"
  "pf4 -DS2P_MGM_PF=4"
  "ch16_fa15 -DS2P_MGM_CH=16"
  "pf16 -DS2P_MGM_PF=16"
)
case "$1" in
build)
  for v in "${VARIANTS[@]}"; do
    set -- $v; name=$1; shift
    mkdir -p build/variants/$name
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-value -fvisibility=hidden "$@" \
      -o build/variants/$name/libs2p_hip.so $SRC &
  done
  wait
  ls -la build/variants/*/libs2p_hip.so
  ;;
run)
  cp s2p_amd/lib/libs2p_hip.so build/libs2p_hip.orig.so
  for d in build/variants/*/; do
    name=$(basename $d)
    cp $d/libs2p_hip.so s2p_amd/lib/libs2p_hip.so
    for lazy in ${LAZY:-0}; do
      S2P_MGM_LAZY=$lazy timeout 120 python bench.py --algo census --recursion 1 --steps ${STEPS:-20} --warmup 4 --no-cpu 2>/dev/null | \
        python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name lazy=$lazy', d['ms_per_step'], 'agg', d['stage_ms']['aggregate'], 'wta', d['stage_ms']['wta'])"
      S2P_MGM_LAZY=$lazy timeout 120 python bench.py --algo census --recursion 1 --streams 2 --steps ${STEPS:-20} --warmup 4 --no-cpu 2>/dev/null | \
        python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name lazy=$lazy 2 streams', d['ms_per_step'])"
    done
  done
  cp build/libs2p_hip.orig.so s2p_amd/lib/libs2p_hip.so
  ;;
*) echo "usage: $0 build|run"; exit 2;;
esac
